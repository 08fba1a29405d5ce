"""Adam for the whole model in one launch (include/gpn.h section O).

``FusedAdam`` IS a ``torch.optim.Adam`` (same constructor, same ``state_dict`` layout: per parameter ``step`` / ``exp_avg`` /
``exp_avg_sq``), so checkpoints written by the reference's ``configure_optimizers`` (network/model.py:1051-1055) load and
vice versa.  On CUDA fp32 parameters the step is ``gpn_adam_step``: the tensors are described by a device table that is
rebuilt only when the set of parameters with a gradient - or a gradient's address - changes (the sparse U-Nets' gradients
live in persistent buffers, network/net_exec.py; the caching allocator hands the few head gradients the same blocks step
after step).  Per step the host does one pass over the parameters' gradient addresses and one library call; torch's fused /
foreach Adam spends ~1 ms per step grouping the ~330 tensors and issues 9-15 launches.
Anything else (CPU tensors in the oracle-backed tests, amsgrad, weight decay, maximize) takes torch's own implementation."""
import ctypes

import numpy as np
import torch

from . import _C

_TABLE_DT = np.dtype([("param", np.uint64), ("grad", np.uint64), ("exp_avg", np.uint64), ("exp_avg_sq", np.uint64),
                      ("numel", np.int64)])


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **kw):
        kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, **kw)
        self._cache = {}     # group index -> (gradient signature, [(table, block_first, n_blocks, params)] per step count)
        self._nstep = {}     # parameter -> steps taken (the ``step`` tensors of the state are refreshed on state_dict())
        # gradients whose address changes from step to step (allocated by autograd for the few non-U-Net tensors) are
        # copied into buffers of their own before the update (one foreach copy), so that the device table stays valid
        self._own_grad = {}

    # ---------------------------------------------------------------------------------------------- state (de)serialisation
    def _sync_step_tensors(self):
        for p, n in self._nstep.items():
            st = self.state.get(p)
            if st:
                st["step"] = torch.tensor(float(n), dtype=torch.float32)

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache.clear()
        self._nstep = {p: int(st["step"]) for p, st in self.state.items() if "step" in st}

    # ---------------------------------------------------------------------------------------------- step
    @staticmethod
    def _native_ok(group, params) -> bool:
        return (not group["amsgrad"] and group["weight_decay"] == 0 and not group["maximize"]
                and not group.get("capturable", False) and not group.get("differentiable", False)
                and all(p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous()
                        and p.grad.is_contiguous() and not p.grad.is_sparse for p in params))

    def _grad_ptr(self, p):
        own = self._own_grad.get(p)
        return own.data_ptr() if own is not None else p.grad.data_ptr()

    def _build(self, group, params):
        """device tables for the parameters that have a gradient, one per distinct step count (parameters the training
        schedule switched on later lag behind: normally there is one table)"""
        L = _C.lib()
        dev = params[0].device
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._nstep.setdefault(p, int(st["step"]))
        by_step = {}
        for p in params:
            by_step.setdefault(self._nstep[p], []).append(p)
        subs = []
        for plist in by_step.values():
            host = np.zeros(len(plist), _TABLE_DT)
            first = np.zeros(len(plist), np.int32)
            blocks = 0
            for i, p in enumerate(plist):
                st = self.state[p]
                host[i] = (p.data_ptr(), self._grad_ptr(p), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                first[i] = blocks
                blocks += int(L.gpn_adam_blocks(ctypes.c_int64(p.numel())))
            # pinned staging + asynchronous upload: a pageable copy would wait for everything queued on the stream
            h_table, h_first = torch.from_numpy(host.view(np.uint8).copy()).pin_memory(), torch.from_numpy(first).pin_memory()
            subs.append((h_table.to(dev, non_blocking=True), h_first.to(dev, non_blocking=True), blocks, plist, (h_table, h_first)))
        return subs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            own = self._own_grad
            if own:
                moved = [p for p in own if p.grad is not None]
                if moved:
                    torch._foreach_copy_([own[p] for p in moved], [p.grad for p in moved])
            ptrs = [0 if p.grad is None else (1 if p in own else p.grad.data_ptr()) for p in group["params"]]
            cached = self._cache.get(gi)
            if cached is None or cached[0] != ptrs:
                params = [p for p in group["params"] if p.grad is not None]
                if not params:
                    continue
                if cached is not None:  # which gradients moved since the tables were built?  they get buffers of their own
                    for p, old, new in zip(group["params"], cached[0], ptrs):
                        if old > 1 and new > 1 and old != new and p.grad.is_cuda:
                            own[p] = p.grad.detach().clone()
                    ptrs = [0 if p.grad is None else (1 if p in own else p.grad.data_ptr()) for p in group["params"]]
                if not self._native_ok(group, params):
                    self._cache.pop(gi, None)
                    self._sync_step_tensors()
                    saved = self.param_groups
                    self.param_groups = [group]  # torch's implementation, for this group only
                    try:
                        super().step()
                    finally:
                        self.param_groups = saved
                    for p in params:
                        self._nstep[p] = int(self.state[p]["step"])
                    continue
                cached = (ptrs, self._build(group, params))
                self._cache[gi] = cached
            L = _C.lib()
            dev = cached[1][0][3][0].device
            stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))
            beta1, beta2 = group["betas"]
            nstep = self._nstep
            for table, first, blocks, plist, _pinned in cached[1]:
                n = nstep[plist[0]] + 1
                _C.check(L.gpn_adam_step(ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(first.data_ptr()), len(plist), blocks,
                                         ctypes.c_double(group["lr"]), ctypes.c_double(beta1), ctypes.c_double(beta2),
                                         ctypes.c_double(group["eps"]), ctypes.c_int64(n), stream), "gpn_adam_step")
                for p in plist:
                    nstep[p] = n
        return loss
