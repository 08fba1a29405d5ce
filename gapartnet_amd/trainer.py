"""Minimal trainer for ``GAPartNet`` with Lightning's hook protocol (Lightning itself is absent from the MI355X image;
the reference runs under ``lightning.pytorch.Trainer`` configured by gapartnet.yaml ``trainer:``).

One process per GPU: launched under ``torch.distributed.run`` every rank reads RANK / LOCAL_RANK / WORLD_SIZE and joins
a ``nccl`` (= RCCL over xGMI) process group — ``gloo`` on CPU.  Scenes are independent: each rank gets a
disjoint shard of scenes (DistributedSampler); the only data-path collective is the gradient mean all-reduce of
``grad_sync.GradSync`` (31.6 MB fp32 per step for the default config, one in-place all-reduce per sparse U-Net plus one
for the small heads) between ``backward()`` and ``optimizer.step()``; BatchNorm statistics stay per rank like the reference
(no SyncBN).  Sub-networks the training schedule leaves without gradients in early epochs keep
``grad is None`` on every rank — the behaviour the reference gets from DDP's ``find_unused_parameters``.
"""
import contextlib
import os
import sys
import time
from collections import defaultdict
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


@contextlib.contextmanager
def native_stdout_to_stderr():
    """RCCL and gloo print banners ("RCCL version : ...", "[Gloo] Rank 0 is connected ...") on the C-level stdout when a
    communicator comes up; callers whose stdout is a protocol (bench.py: ONE JSON line) wrap communicator creation in this"""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def seed_everything(seed: int) -> int:
    """Lightning's ``seed_everything`` (gapartnet.yaml ``seed_everything: 23333``): python, numpy and torch generators,
    the SAME seed on every rank — the dataset shuffles its file list with the global generators at construction, and all
    ranks must hold the same order for DistributedSampler's index shards to be disjoint."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    return seed


def distributed_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(device_type: str = "cuda"):
    """-> (rank, local_rank, world_size, device).  Initialises the process group when WORLD_SIZE > 1."""
    rank, local_rank, world = distributed_env()
    use_cuda = device_type == "cuda" and torch.cuda.is_available()
    # GPN_DIST_SHARE_DEVICE=1 (test rigs with fewer GPUs than ranks): every rank on cuda:0 and gloo instead of RCCL, which
    # refuses two ranks on one device - exercises the multi-rank code path, not its performance
    share = use_cuda and os.environ.get("GPN_DIST_SHARE_DEVICE") == "1"
    device = torch.device(f"cuda:{0 if share else local_rank}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
        from .affinity import pin_to_gpu_node
        pin_to_gpu_node(device.index or 0)  # this rank's threads onto the cores of its GPU's NUMA node (GPN_NO_PIN=1: off)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        nccl = use_cuda and not share
        with native_stdout_to_stderr():
            dist.init_process_group(backend="nccl" if nccl else "gloo", rank=rank, world_size=world,
                                    **({"device_id": device} if nccl else {}))
    return rank, local_rank, world, device


class _TrainStep(nn.Module):
    """``forward`` routed to the module's ``training_step`` (the callable a DDP-style wrapper needs; tools/exchange_profile.py)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, batch, batch_idx):
        return self.module.training_step(batch, batch_idx)


def move_batch(batch, device):
    if isinstance(batch, (list, tuple)):
        return [b.to(device) if hasattr(b, "to") else b for b in batch]
    return batch.to(device) if hasattr(batch, "to") else batch


class MetricLog:
    """epoch means of everything the module ``log``s (weighted by batch_size, summed over ranks).

    Nothing is read back while an epoch runs: tensor values are accumulated as device tensors (a ``float(value)`` per key
    would be eight device->host waits per training step, the first of which holds the host until the whole forward has
    finished); ``reduce`` does ONE all-reduce and ONE host read per epoch.  The set of keys can differ between ranks
    (per-class AP keys exist only where an epoch had proposals), so ranks first agree on the union of names."""

    def __init__(self):
        self.sum: Dict[str, object] = {}
        self.weight = defaultdict(float)

    def __call__(self, name, value, batch_size=None, sync_dist=False):
        w = float(batch_size or 1)
        if isinstance(value, torch.Tensor):
            v = value.detach().to(torch.float64) * w
        else:
            v = float(value) * w
        prev = self.sum.get(name)
        self.sum[name] = v if prev is None else prev + v
        self.weight[name] += w

    def reduce(self, device) -> Dict[str, float]:
        names = sorted(self.sum)
        distributed = dist.is_initialized() and dist.get_world_size() > 1
        if distributed:
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, names)
            names = sorted(set().union(*gathered))
        if not names:
            return {}
        zero = torch.zeros((), dtype=torch.float64, device=device)
        sums = torch.stack([torch.as_tensor(self.sum.get(n, 0.0), dtype=torch.float64).to(device) + zero for n in names])
        weights = torch.tensor([self.weight.get(n, 0.0) for n in names], dtype=torch.float64, device=device)
        t = torch.stack([sums, weights], dim=1)
        if distributed:
            dist.all_reduce(t)
        host = t.cpu()
        out = {n: float(host[i, 0] / max(float(host[i, 1]), 1e-12)) for i, n in enumerate(names)}
        self.sum.clear(); self.weight.clear()
        return out


class Trainer:
    def __init__(self, max_epochs: int = 1, default_root_dir: str = "./runs", accelerator: str = "gpu",
                 limit_train_batches: Optional[int] = None, limit_val_batches: Optional[int] = None,
                 check_val_every_n_epoch: int = 1, monitor: str = "monitor_metrics/mean_mAP", save_top_k: int = 5,
                 log_every_n_steps: int = 10, find_unused_parameters: bool = True, enable_checkpointing: bool = True,
                 seed: Optional[int] = 23333, **_ignored):
        self.max_epochs, self.root = max_epochs, default_root_dir
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.check_val_every_n_epoch, self.monitor, self.save_top_k = check_val_every_n_epoch, monitor, save_top_k
        self.log_every_n_steps, self.find_unused = log_every_n_steps, find_unused_parameters
        self.enable_checkpointing = enable_checkpointing
        self.device_type = "cuda" if accelerator in ("gpu", "cuda", "auto") else "cpu"
        self.rank, self.local_rank, self.world, self.device = init_distributed(self.device_type)
        # before any datamodule.setup(): identical generators on all ranks (gapartnet.yaml seed_everything; None = leave them)
        self.seed = seed_everything(seed) if seed is not None else None
        self.current_epoch = 0
        self.global_step = 0
        self.history: List[Dict[str, float]] = []
        self._best: List = []

    # ------------------------------------------------------------------------------------------------
    def _attach(self, model, log: MetricLog):
        model.trainer = self
        if hasattr(model, "_log_sink"):
            model._log_sink = log
        model.to(self.device)

    def _set_epoch(self, model, epoch):
        self.current_epoch = epoch
        if hasattr(model, "_current_epoch"):
            model._current_epoch = epoch

    def _setup(self, datamodule, stage: str):
        """datamodule.setup under the agreed seed: the datasets shuffle their file lists with the global generators
        (dataset/gapartnet.py:49-50 of the reference), and every rank must end up with the same order"""
        if self.seed is not None:
            seed_everything(self.seed)
        datamodule.setup(stage)

    def fit(self, model, datamodule=None, train_dataloaders=None, val_dataloaders=None, ckpt_path: Optional[str] = None):
        log = MetricLog()
        self._attach(model, log)
        if datamodule is not None:
            self._setup(datamodule, "fit")
            sampler = None
            if self.world > 1:
                sampler = torch.utils.data.distributed.DistributedSampler(
                    datamodule.train_data_files, num_replicas=self.world, rank=self.rank, shuffle=True, drop_last=True)
            train_dataloaders = datamodule.train_dataloader(sampler=sampler)
            val_dataloaders = datamodule.val_dataloader()
        optimizer = model.configure_optimizers()
        start_epoch = 0
        if ckpt_path:
            state = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            model.load_state_dict(state["state_dict"])
            if "optimizer" in state:
                optimizer.load_state_dict(state["optimizer"])
            start_epoch = int(state.get("epoch", -1)) + 1
        grad_sync = None
        if self.world > 1:
            from .grad_sync import GradSync
            grad_sync = GradSync(model)
            grad_sync.broadcast_parameters()
        for epoch in range(start_epoch, self.max_epochs):
            self._set_epoch(model, epoch)
            sampler = getattr(train_dataloaders, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            model.train()
            t0 = time.time()
            feed = train_dataloaders
            if self.device.type == "cuda" and hasattr(model, "voxel_size"):
                from .dataset.prefetch import DevicePrefetcher
                aug = getattr(datamodule, "aug", None) if getattr(datamodule, "device_pipeline", False) else None
                # batch i+1 prepared while batch i trains (raw scenes are also augmented there, per batch, on the GPU)
                feed = DevicePrefetcher(train_dataloaders, model, self.device, augmentation=aug)
            from .network.net_exec import release_gradients
            from .optim import FusedAdam
            acknowledges = isinstance(optimizer, FusedAdam)
            for batch_idx, batch in enumerate(feed):
                if self.limit_train_batches is not None and batch_idx >= self.limit_train_batches:
                    break
                batch = move_batch(batch, self.device)
                optimizer.zero_grad(set_to_none=True)
                loss = model.training_step(batch, batch_idx)
                loss.backward()
                if grad_sync is not None:
                    grad_sync.sync()
                optimizer.step()
                # the step has consumed the sparse U-Nets' gradients: their persistent buffers may be overwritten by the next
                # backward pass (network/net_exec.py, gradient hand-over contract).  FusedAdam acknowledges that itself; an
                # optimizer from a configure_optimizers override does not, and would cost an allocation per step without this
                if not acknowledges:
                    release_gradients(model)
                self.global_step += 1
            metrics = log.reduce(self.device)
            metrics["epoch_time_s"] = time.time() - t0
            if val_dataloaders is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                if grad_sync is not None:
                    grad_sync.broadcast_buffers()
                metrics.update(self._eval_loop(model, val_dataloaders, "validation", log))
            metrics["epoch"] = epoch
            self.history.append(metrics)
            if self.rank == 0:
                shown = {k: round(v, 4) for k, v in metrics.items() if "loss/total" in k or "monitor" in k or k == "epoch"}
                print(f"[trainer] epoch {epoch}: {shown}", flush=True)
                if self.enable_checkpointing:
                    self._checkpoint(model, optimizer, epoch, metrics)
        return self.history

    @torch.no_grad()
    def _eval_loop(self, model, loaders, kind: str, log: MetricLog) -> Dict[str, float]:
        model.eval()
        loaders = loaders if isinstance(loaders, (list, tuple)) else [loaders]
        step = model.validation_step if kind == "validation" else model.test_step
        if hasattr(model, "defer_validation_outputs"):
            # nobody here looks at a step's return value: the model may read a step's sizes while the next step is queued
            model.defer_validation_outputs = True
        for loader_idx, loader in enumerate(loaders):
            feed = loader
            if self.device.type == "cuda" and hasattr(model, "voxel_size"):
                from .dataset.prefetch import DevicePrefetcher
                feed = DevicePrefetcher(loader, model, self.device)  # batch i + 1 voxelised / rulebooks built while batch i runs
            for batch_idx, batch in enumerate(feed):
                if self.limit_val_batches is not None and batch_idx >= self.limit_val_batches:
                    break
                step(move_batch(batch, self.device), batch_idx, loader_idx)
        (model.on_validation_epoch_end if kind == "validation" else model.on_test_epoch_end)()
        if hasattr(model, "defer_validation_outputs"):
            model.defer_validation_outputs = False
        return log.reduce(self.device)

    def validate(self, model, datamodule=None, dataloaders=None):
        log = MetricLog()
        self._attach(model, log)
        if datamodule is not None:
            self._setup(datamodule, "validate")
            dataloaders = datamodule.val_dataloader()
        return self._eval_loop(model, dataloaders, "validation", log)

    def test(self, model, datamodule=None, dataloaders=None):
        log = MetricLog()
        self._attach(model, log)
        if datamodule is not None:
            self._setup(datamodule, "test")
            dataloaders = datamodule.test_dataloader()
        return self._eval_loop(model, dataloaders, "test", log)

    def _checkpoint(self, model, optimizer, epoch, metrics):
        os.makedirs(self.root, exist_ok=True)
        score = metrics.get(self.monitor, float("-inf"))
        name = f"epoch_{epoch:03d}_mAP_{metrics.get(self.monitor, 0.0):.2f}.ckpt"
        path = os.path.join(self.root, name)
        torch.save({"state_dict": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
                    "hyper_parameters": dict(getattr(model, "hparams", {}) or {})}, path)
        self._best.append((score, path))
        self._best.sort(key=lambda sp: sp[0], reverse=True)
        for _, stale in self._best[self.save_top_k:]:
            if os.path.exists(stale):
                os.remove(stale)
        self._best = self._best[:self.save_top_k]
