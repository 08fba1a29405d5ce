"""Data-parallel gradient exchange for the perception path: the one collective step of multi-GPU training.

The reference trains under Lightning's DDP strategy (gapartnet.yaml ``trainer: strategy: ddp_find_unused_parameters_true``
semantics: mean of the ranks' gradients; parameters no rank used keep ``grad is None`` so Adam skips them).  Wrapping this
model in ``torch.nn.parallel.DistributedDataParallel`` costs 5-18 ms on a 17 ms step on MI355X (per-parameter hooks, ~320
per-parameter scale + copy kernels, the unused-parameter graph walk, and a rebuilt bucket view per step; measured with
tools/ddp_profile.py), so the exchange is done here instead, shaped by what the native executor already produces:

* each SparseUNet's backward writes ALL its parameter gradients into ONE contiguous fp32 buffer (net_exec._NetFn), so a
  UNet is one bucket and is all-reduced in place - no flatten, no copy back;
* the remaining ~100 small tensors (heads, score / NPCS MLPs) are one more bucket: one ``cat``, one all-reduce, and the
  parameters' ``.grad`` are re-pointed at views of the reduced buffer (no copy back);
* which parameters received a gradient is known on the HOST as soon as ``backward()`` returns (the kernels are still in
  flight), so the used-anywhere consensus is a ~330-byte MAX all-reduce between the hosts over gloo - no device sync.

The payload is 31.6 MB per step for the default model: ~0.2 ms on a 7-link xGMI ring, which is why no overlap with backward
is attempted.  RCCL ("nccl") averages in the collective (ReduceOp.AVG); gloo sums and the buffer is scaled afterwards.
"""
import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist


def _unet_buckets(model):
    """one list per SparseUNet the native executor can express, in the executor's flat-gradient order.  Derived from the
    module tree alone (the program is built here if the net has not run yet), so every rank gets the same buckets even
    when a sub-network ran on some ranks only (no proposals on the others)."""
    from .network import net_exec
    out = []
    for m in model.modules():
        if not getattr(m, "use_native_executor", False):
            continue
        prog = net_exec.program_for(m)
        if prog is not None:
            order = list(prog.params())
            if order and all(p.requires_grad for p in order):
                out.append((order, prog))
    return out


class GradSync:
    """``sync()`` after ``loss.backward()`` and before ``optimizer.step()``; every rank must call it every step."""

    def __init__(self, model, group=None):
        assert dist.is_initialized(), "GradSync needs an initialised process group"
        self.model, self.group = model, group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # host-side consensus channel: the default group when it is gloo already, else a gloo twin of it; if gloo cannot
        # be brought up (no usable interface), the consensus goes over the device group and costs one device sync per step
        self.host_group = group
        if self.backend != "gloo":
            try:
                from .trainer import native_stdout_to_stderr
                with native_stdout_to_stderr():  # gloo announces its connections on stdout
                    ranks = dist.get_process_group_ranks(group) if group is not None else None
                    self.host_group = dist.new_group(ranks=ranks, backend="gloo")  # the twin spans the SAME ranks
            except Exception as exc:  # noqa: BLE001 - any transport failure means "fall back", every rank fails alike
                print(f"[grad_sync] gloo side channel unavailable ({type(exc).__name__}: {exc}); using the device group",
                      flush=True)
                self.host_group = None
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._buckets: Optional[List[List[int]]] = None
        self.stats = {"in_place": 0, "flattened": 0, "skipped": 0, "steps": 0, "host_ms": 0.0}
        self._verified = {}  # id(program) -> [slice-by-slice walks of the whole-buffer shortcut done, on which buffer]
        # (gloo has no coalescing: its buckets go one call each, as before)
        self._coalesce = self.backend == "nccl"

    # ------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0):
        """what DDP does at construction: every rank starts from rank ``src``'s parameters and buffers"""
        from .trainer import native_stdout_to_stderr
        with torch.no_grad(), native_stdout_to_stderr():  # the first collective may be what brings the communicator up
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                    dist.broadcast(t.data, src, group=self.group)

    def broadcast_buffers(self, src: int = 0):
        """BatchNorm running statistics stay per rank while training (batch statistics are used there); evaluation on
        every rank uses rank ``src``'s, as DDP's per-forward buffer broadcast arranges in the reference"""
        with torch.no_grad():
            for t in self.model.buffers():
                dist.broadcast(t.data, src, group=self.group)

    # ------------------------------------------------------------------------------------------------
    def _build_buckets(self):
        taken, buckets, progs = set(), [], []
        for order, prog in _unet_buckets(self.model):
            ids = [self._index[id(p)] for p in order if id(p) in self._index]
            if len(ids) == len(order) and not (taken & set(ids)):
                buckets.append(ids)
                progs.append(prog)
                taken.update(ids)
        rest = [i for i in range(len(self.params)) if i not in taken]
        if rest:
            buckets.append(rest)
            progs.append(None)
        self._buckets, self._progs = buckets, progs

    # steps on which the whole-buffer shortcut is verified gradient by gradient (then it is trusted: the aliasing is a
    # property of how the step is written - zero_grad(set_to_none=True), no tied parameters, no gradient hooks - and does not
    # change from step to step); a larger VERIFY_STEPS keeps the full walk for longer
    VERIFY_STEPS = 3

    def _executor_flat(self, prog, grads, total) -> Optional[torch.Tensor]:
        """the executor's own gradient buffer of this step if the bucket's ``.grad`` tensors are exactly its slices, back
        to back (autograd hands the slices over without copying when ``.grad`` was None).  First and last slice are checked
        on every step (O(1)); every slice on the first VERIFY_STEPS steps THIS bucket takes the shortcut (and again whenever
        the executor hands out a new buffer), because a middle parameter whose ``.grad`` is
        NOT a slice (tied parameter, accumulation into an existing ``.grad``, a hook that made AccumulateGrad clone) would
        otherwise be left un-averaged without any error."""
        flat = getattr(prog, "last_pgrad", None)
        if flat is None or flat.numel() != total or grads[0] is None or grads[-1] is None:
            return None
        base = flat.data_ptr()
        if grads[0].data_ptr() != base or grads[-1].data_ptr() + grads[-1].numel() * 4 != base + total * 4:
            return None
        # per bucket: the score / NPCS U-Nets first run at epochs start_scorenet / start_npcs, long after the first steps
        # of the run, and the executor retires its buffer when somebody keeps a reference to it (new base address)
        seen = self._verified.setdefault(id(prog), [0, 0])  # [full walks done, base address they were done on]
        if seen[1] != base:
            seen[0], seen[1] = 0, base
        if seen[0] < self.VERIFY_STEPS:
            seen[0] += 1
            ptr = base
            for g in grads:
                if g is None or g.data_ptr() != ptr or g.dtype != torch.float32 or not g.is_contiguous():
                    return None
                ptr += g.numel() * 4
            if ptr != base + total * 4:
                return None
        return flat

    @staticmethod
    def _shared_flat(grads) -> Optional[torch.Tensor]:
        """a 1-D alias of the storage the gradients tile back to back in this order, if they do (the executor's buffer:
        autograd hands its slices to ``.grad`` without copying when ``.grad`` was None)"""
        g0 = grads[0]
        store = g0.untyped_storage()
        sptr, ptr, total = store.data_ptr(), g0.data_ptr(), 0
        for g in grads:
            if g.data_ptr() != ptr or g.dtype != g0.dtype or not g.is_contiguous():
                return None
            n = g.numel()
            ptr += n * 4
            total += n
        if g0.dtype != torch.float32 or grads[-1].untyped_storage().data_ptr() != sptr or ptr > sptr + store.nbytes():
            return None
        return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(store, g0.storage_offset(), (total,), (1,))

    def _all_reduce_mean(self, flats):
        """mean over the ranks of every buffer in ``flats``, in place, as ONE collective launch (RCCL group call): each
        separate all-reduce costs ~0.2 ms of host bookkeeping and a stream hand-off (four per step were +2 ms on a 13 ms
        step, bench.py with GPN_BENCH_FORCE_GRAD_SYNC=1)"""
        if not flats:
            return
        op = dist.ReduceOp.AVG if self.backend == "nccl" else dist.ReduceOp.SUM
        if len(flats) == 1 or not self._coalesce:
            for flat in flats:
                dist.all_reduce(flat, op=op, group=self.group)
        else:
            try:
                from torch.distributed.distributed_c10d import _coalescing_manager
                with _coalescing_manager(group=self.group, device=flats[0].device, async_ops=False):
                    for flat in flats:
                        dist.all_reduce(flat, op=op, group=self.group)
            except (ImportError, RuntimeError, TypeError) as exc:  # a backend without coalescing: one call per buffer
                print(f"[grad_sync] coalesced all-reduce unavailable ({type(exc).__name__}: {exc}); one call per bucket", flush=True)
                self._coalesce = False
                for flat in flats:
                    dist.all_reduce(flat, op=op, group=self.group)
        if self.backend != "nccl":
            for flat in flats:
                flat.div_(self.world)

    @torch.no_grad()
    def sync(self):
        t0 = time.perf_counter()
        if self._buckets is None:
            self._build_buckets()
        params = self.params
        used = torch.tensor([p.grad is not None for p in params], dtype=torch.uint8)
        if self.host_group is not None or self.backend == "gloo":
            dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.host_group)
        else:
            on_dev = used.to(params[0].device, torch.int32)
            dist.all_reduce(on_dev, op=dist.ReduceOp.MAX, group=self.group)
            used = on_dev.cpu()
        used = used.tolist()
        self.stats["steps"] += 1
        flats, repoint = [], []  # buffers to reduce in this step's one collective; (flat, live ids) whose .grad follow it
        for ids, prog in zip(self._buckets, self._progs):
            # (every rank must reduce the same number of elements: the whole-buffer path only when the consensus says every
            # parameter of the bucket has a gradient somewhere - which is always the case for a net that ran)
            if prog is not None and all(used[i] for i in ids):
                flat = self._executor_flat(prog, [params[i].grad for i in ids], prog.grad_total)
                if flat is not None:
                    flats.append(flat)
                    self.stats["in_place"] += 1
                    continue
            live = [i for i in ids if used[i]]
            if not live:
                self.stats["skipped"] += 1
                continue  # no rank touched this sub-network: gradients stay None, the optimizer skips it
            grads = [params[i].grad for i in live]
            flat = self._shared_flat(grads) if all(g is not None for g in grads) else None
            if flat is not None:
                flats.append(flat)
                self.stats["in_place"] += 1
                continue
            ref = params[live[0]]
            if all(g is None for g in grads):  # used on another rank only: contribute zeros
                flat = torch.zeros(sum(params[i].numel() for i in live), dtype=ref.dtype, device=ref.device)
            else:
                flat = torch.cat([g.reshape(-1) if g is not None else
                                  torch.zeros(params[i].numel(), dtype=ref.dtype, device=ref.device)
                                  for i, g in zip(live, grads)])
            flats.append(flat)
            repoint.append((flat, live))
            self.stats["flattened"] += 1
        self._all_reduce_mean(flats)
        for flat, live in repoint:
            for i, piece in zip(live, flat.split([params[i].numel() for i in live])):
                params[i].grad = piece.view_as(params[i])
        self.stats["host_ms"] += (time.perf_counter() - t0) * 1e3  # host time spent issuing the exchange (all steps)
