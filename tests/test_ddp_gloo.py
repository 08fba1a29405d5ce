"""N > 1 path on CPU: two processes, gloo backend, the same Trainer / GradSync exchange the GPU path uses (with RCCL there).
Checks the three things data-parallel training of this model depends on: disjoint scene shards per rank, the gradient
mean all-reduce between backward and the optimizer step, and parameters staying identical across ranks after steps —
including the phase where the score / NPCS sub-networks receive no gradient on any rank (they must keep grad None, the
semantics the reference gets from DDP find_unused_parameters) and a sub-network used by one rank only."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, schedule, out_dir, root_dir="synthetic:8", packed=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from gapartnet_amd import backend
    from gapartnet_amd.dataset.gapartnet import GAPartNetInst
    from gapartnet_amd.smoke import make_model
    from gapartnet_amd.trainer import Trainer
    from oracle import torch_ops
    backend.use(torch_ops)

    model = make_model(schedule, channels=[16, 32], seed=0)
    dm = GAPartNetInst(root_dir=root_dir, max_points=1500, train_batch_size=2, val_batch_size=2, test_batch_size=2,
                       num_workers=0, packed_cache=packed)
    seen = []
    orig = model.training_step

    def spy(batch, batch_idx):
        seen.extend(pc.pc_id for pc in batch)
        return orig(batch, batch_idx)
    model.training_step = spy
    trainer = Trainer(max_epochs=1, accelerator="cpu", limit_train_batches=2, enable_checkpointing=False, check_val_every_n_epoch=100,
                      default_root_dir=out_dir)
    history = trainer.fit(model, datamodule=dm, val_dataloaders=None)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ids = [None] * world
    dist.all_gather_object(ids, seen)
    if rank == 0:
        torch.save({"params_equal": bool(torch.equal(gathered[0], gathered[1])), "ids": ids,
                    "loss": history[0]["train_loss/total_loss"], "finite": bool(torch.isfinite(flat).all())},
                   os.path.join(out_dir, "result.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("schedule", [(0, 0), (5, 10)])
def test_two_rank_ddp_training(tmp_path, schedule):
    port = _free_port()
    # validation loaders are skipped by passing val_dataloaders=None through the datamodule path
    mp.spawn(_worker, args=(2, port, schedule, str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "result.pt"), weights_only=False)
    assert res["params_equal"], "ranks diverged: gradients were not all-reduced"
    assert res["finite"] and res["loss"] > 0
    a, b = set(res["ids"][0]), set(res["ids"][1])
    assert len(a) == 4 and len(b) == 4 and not (a & b), "ranks must train on disjoint scene shards"


def _no_proposal_worker(rank, world, port, out_dir):
    """rank 1 never has a proposal (its clustering keeps nothing), rank 0 always has: ScoreNet / NPCS-Net run on rank 0 only"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from gapartnet_amd import backend
    from gapartnet_amd.dataset.gapartnet import GAPartNetInst
    from gapartnet_amd.smoke import make_model
    from gapartnet_amd.trainer import Trainer
    from oracle import torch_ops
    backend.use(torch_ops)
    model = make_model((0, 0), channels=[16, 32], seed=0)
    if rank == 1:
        model.min_num_points_per_proposal = 10 ** 9  # (a per-rank setting, not a parameter: no cluster survives on this rank)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    runs = []
    orig = model.score_unet.forward
    model.score_unet.forward = lambda *a, **k: (runs.append(1), orig(*a, **k))[1]
    dm = GAPartNetInst(root_dir="synthetic:12", max_points=1500, train_batch_size=2, val_batch_size=2, test_batch_size=2, num_workers=0)
    trainer = Trainer(max_epochs=1, accelerator="cpu", limit_train_batches=3, enable_checkpointing=False, check_val_every_n_epoch=100,
                      default_root_dir=out_dir)
    trainer.fit(model, datamodule=dm, val_dataloaders=None)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    calls = [None] * world
    dist.all_gather_object(calls, len(runs))
    moved = {n: bool((p.detach() != before[n]).any()) for n, p in model.named_parameters()}
    if rank == 0:
        torch.save({"params_equal": bool(torch.equal(gathered[0], gathered[1])), "score_unet_calls": calls, "finite": bool(torch.isfinite(flat).all()),
                    "score_moved": any(v for n, v in moved.items() if n.startswith("score_unet.")),
                    "npcs_moved": any(v for n, v in moved.items() if n.startswith("npcs_unet."))}, os.path.join(out_dir, "noprop.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_without_proposals_for_three_steps(tmp_path):
    """round-5 review: one rank has no proposal for three steps in a row while the other has some every step.  The rank without
    skips ScoreNet / NPCS-Net (their gradients stay None there, the reference's behaviour on such a batch); the exchange must
    count it in with zeros - every rank reduces the same buckets - and both ranks must end with the same parameters, the proposal
    networks' included, which have moved."""
    mp.spawn(_no_proposal_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "noprop.pt"), weights_only=False)
    assert res["score_unet_calls"][0] == 3 and res["score_unet_calls"][1] == 0, res["score_unet_calls"]
    assert res["params_equal"] and res["finite"], "ranks diverged"
    assert res["score_moved"] and res["npcs_moved"], "the proposal networks must have been updated on both ranks"


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a, self.b, self.c = torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)


def _grad_sync_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gapartnet_amd.grad_sync import GradSync
    model = _Toy()
    if rank == 1:
        with torch.no_grad():
            model.a.weight.add_(1.0)  # must be overwritten by the rank-0 broadcast
    sync = GradSync(model)
    sync.broadcast_parameters()
    x = torch.arange(8, dtype=torch.float32).view(2, 4) * (rank + 1)
    h = model.a(x)
    # rank 0 uses head b, rank 1 uses neither head (h only); head c is used by nobody
    loss = (model.b(h).sum() + h.sum()) if rank == 0 else (2.0 * h).sum()
    loss.backward()
    local = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
    sync.sync()
    synced = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
    both = [None] * world
    dist.all_gather_object(both, (local, synced, model.a.weight.detach().clone()))
    if rank == 0:
        torch.save(both, os.path.join(out_dir, "toy.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_mean_and_unused_semantics(tmp_path):
    mp.spawn(_grad_sync_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    (l0, s0, w0), (l1, s1, w1) = torch.load(os.path.join(str(tmp_path), "toy.pt"), weights_only=False)
    assert torch.equal(w0, w1), "parameters were not broadcast from rank 0"
    for name in l0:
        if name.startswith("c."):
            assert s0[name] is None and s1[name] is None, "a sub-network no rank used must keep grad None"
            continue
        g0 = l0[name] if l0[name] is not None else torch.zeros_like(s0[name])
        g1 = l1[name] if l1[name] is not None else torch.zeros_like(s0[name])
        assert l1[name] is not None or name.startswith("b."), name
        want = (g0 + g1) / 2
        assert torch.allclose(s0[name], want, rtol=0, atol=1e-6) and torch.equal(s0[name], s1[name]), name


def test_four_rank_training_keeps_ranks_identical(tmp_path):
    """the same trainer path at world_size 4 (the driver's N = 4 launch): every collective of GradSync - parameter broadcast,
    host-side used-set consensus, bucket all-reduces - with more than two participants"""
    mp.spawn(_worker, args=(4, _free_port(), (5, 10), str(tmp_path)), nprocs=4, join=True)
    res = torch.load(os.path.join(str(tmp_path), "result.pt"), weights_only=False)
    assert res["params_equal"] and res["finite"] and res["loss"] > 0
    seen = [set(i) for i in res["ids"]]
    assert [len(s) for s in seen] == [2, 2, 2, 2] and len(set().union(*seen)) == 8, "4 disjoint shards of the 8 scenes"


def test_two_ranks_train_from_one_packed_cache(tmp_path):
    """round 5: both ranks open (and, racing, build) the same memory-mapped scene cache over the same ``.pth`` files; the
    DistributedSampler's shards go through PackedSceneLoader: disjoint scenes per rank, identical parameters after the steps"""
    from tests.golden.recipe import scene_arrays
    root = str(tmp_path / "data")
    for split, n, seed0 in (("train", 8, 100), ("val", 2, 5000), ("test_intra", 2, 6000), ("test_inter", 2, 7000)):
        d = os.path.join(root, split, "pth")
        os.makedirs(d)
        for i in range(n):
            xyz, rgb, sem, ins, npcs, pix = scene_arrays(seed0 + i, 1500)
            torch.save((xyz, rgb, sem, ins, npcs, pix), os.path.join(d, f"StorageFurniture_{seed0 + i:05d}_00_{i % 32:03d}.pth"))
    mp.spawn(_worker, args=(2, _free_port(), (0, 0), str(tmp_path), root, True), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "result.pt"), weights_only=False)
    assert res["params_equal"], "ranks diverged: gradients were not all-reduced"
    assert res["finite"] and res["loss"] > 0
    a, b = set(res["ids"][0]), set(res["ids"][1])
    assert len(a) == 4 and len(b) == 4 and not (a & b), "ranks must train on disjoint scene shards"
    caches = os.listdir(os.path.join(root, ".gpn_cache"))
    assert len([c for c in caches if c.startswith("train-") and ".tmp" not in c]) == 1, caches


def test_every_rank_holds_the_same_file_order_after_setup(tmp_path):
    """the dataset shuffles its ``.pth`` list with the process-global generator; ranks whose generators have
    diverged (here: different amounts of random numbers consumed before setup) must still end up with ONE order, or
    DistributedSampler's index shards overlap.  Trainer._setup re-seeds (gapartnet.yaml seed_everything) before setup."""
    import numpy as np
    import torch
    from gapartnet_amd.dataset.gapartnet import GAPartNetInst
    from gapartnet_amd.trainer import Trainer
    for split in ("train", "val", "test_intra", "test_inter"):
        d = tmp_path / split / "pth"
        d.mkdir(parents=True)
        for i in range(12):
            torch.save((np.zeros((4, 3), np.float32),) * 2, str(d / f"Box_{split}_{i:02d}.pth"))
    orders = []
    for consumed in (0, 17):
        trainer = Trainer(accelerator="cpu", enable_checkpointing=False)
        np.random.rand(consumed)
        dm = GAPartNetInst(str(tmp_path))
        trainer._setup(dm, "fit")
        orders.append([p.split("/")[-1] for p in dm.train_data_files.pc_paths + dm.val_data_files.pc_paths])
    assert orders[0] == orders[1] and sorted(orders[0]) != orders[0]
    assert len(dm.train_data_files) == 12
    dm_all = GAPartNetInst(str(tmp_path), train_with_all=True)
    dm_all.setup("fit")
    assert len(dm_all.train_data_files) == 48, "train_with_all trains on all four splits (dataset/gapartnet.py:354-372 there)"
