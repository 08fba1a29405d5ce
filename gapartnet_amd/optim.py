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
import weakref

import numpy as np
import torch

from . import _C

_TABLE_DT = np.dtype([("param", np.uint64), ("grad", np.uint64), ("exp_avg", np.uint64), ("exp_avg_sq", np.uint64),
                      ("numel", np.int64)])


def _copy_many(dsts, srcs):
    """dst[i].copy_(src[i]) for lists of tensors.  CUDA fp32 contiguous pairs of equal size go through ONE launch
    (gpn_copy_many: the segment table is a kernel argument); torch._foreach_copy_ took its per-tensor path for the model's ~57
    dense-head gradients - 57 hipMemcpyAsync of 3.6 us each on the training stream and 0.11 ms of host time per step."""
    fast = [(d, s_) for d, s_ in zip(dsts, srcs)
            if d.is_cuda and s_.is_cuda and d.dtype == torch.float32 and s_.dtype == torch.float32 and d.is_contiguous()
            and s_.is_contiguous() and d.numel() == s_.numel() and d.device == s_.device]
    if len(fast) != len(dsts) or not fast:
        torch._foreach_copy_(dsts, srcs)
        return
    table = np.empty((len(fast), 3), np.int64)
    table[:, 0] = [s_.data_ptr() for _, s_ in fast]
    table[:, 1] = [d.data_ptr() for d, _ in fast]
    table[:, 2] = [d.numel() for d, _ in fast]
    dev = fast[0][0].device
    stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))
    _C.check(_C.lib().gpn_copy_many(ctypes.c_void_p(table.ctypes.data), ctypes.c_int(len(fast)), stream), "gpn_copy_many")


class FusedAdam(torch.optim.Adam):
    _MAX_TABLE_SETS = 4  # device-table sets kept per parameter group (alternating proposal / no-proposal steps reuse theirs)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **kw):
        kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, **kw)
        # group index -> {signature: table sets}: a signature covers everything the device tables point at (see _signature)
        self._cache = {}
        self._nstep = {}     # parameter -> steps taken (the ``step`` tensors of the state are refreshed on state_dict())
        # gradients whose address changes from step to step (allocated by autograd for the few non-U-Net tensors) are
        # copied into buffers of their own before the update (one foreach copy), so that the device table stays valid
        self._own_grad = {}
        self._last_sig = {}  # group index -> signature of the previous step (to see which gradients moved)
        self._snap = {}      # group index -> (signature, tables, object snapshot) of the previous step, see _unchanged
        # parameters that take a step only when a DEVICE counter is non-zero (set_gate): the proposal networks of a training
        # step whose proposal count was never read on the host
        self._gated = frozenset()
        self._gate_fn = None
        # steps the gate suppressed, counted on the device PER TABLE of gated tensors (the tensors of a table step - or sit out -
        # together; their step number lags by the table's count): parameter -> int64 [1] device tensor shared by its table.
        # (Round 5 kept ONE counter for all gated tables: with training_schedule [5, 10] ScoreNet and NPCS-Net start at different
        # epochs, are two tables, and every step without proposals was counted twice - and charged to tensors that had not
        # started yet.)
        self._skip_ctr = {}
        self._gate_open = None

    def set_gate(self, params, gate_fn):
        """``params`` take part in a step only if the device counter ``gate_fn()`` points at is non-zero (gpn_adam_step_gated);
        ``gate_fn() -> (int64 device tensor, element index) | None`` is asked at every step (None = no gate this step: the
        ordinary rule - parameters without a gradient are skipped on the host - applies).  Reference behaviour this keeps:
        GAPartNet does not run ScoreNet / NPCS-Net on a batch without proposals (network/model.py:573-574), so their
        parameters, moments and step counts stay as they are; the device-counted training step cannot know on the host that
        there were none."""
        self._gated = frozenset(params)
        self._gate_fn = gate_fn
        self._cache.clear()
        self._last_sig.clear()
        self._snap.clear()

    # ---------------------------------------------------------------------------------------------- state (de)serialisation
    def _fold_skips(self, params=None):
        """the host-side step counts take over the steps the gate suppressed (a host read per counter: checkpoints and table
        rebuilds only); the counters of the given parameters (default: all) are retired - for EVERY parameter that shares them"""
        ctrs = {}
        for p, c in self._skip_ctr.items():
            ctrs.setdefault(id(c), (c, []))[1].append(p)
        wanted = None if params is None else {id(self._skip_ctr[p]) for p in params if p in self._skip_ctr}
        for key, (c, plist) in ctrs.items():
            if wanted is not None and key not in wanted:
                continue
            skipped = int(c.item())
            for p in plist:
                self._nstep[p] -= skipped
                del self._skip_ctr[p]

    def _sync_step_tensors(self):
        self._fold_skips()
        for p, n in self._nstep.items():
            st = self.state.get(p)
            if st:
                st["step"] = torch.tensor(float(n), dtype=torch.float32)

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # a checkpoint written by torch's own Adam may carry fused / foreach / capturable = True: this class decides per step
        # whether its kernel or torch's single-tensor path runs, and the latter must not see fused=True next to CPU step tensors
        for group in self.param_groups:
            group["fused"] = None
            group["foreach"] = None
            group["capturable"] = False
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]) and st["step"].is_cuda:
                st["step"] = st["step"].detach().to("cpu", torch.float32)
        self._cache.clear()
        self._last_sig.clear()
        self._snap.clear()
        self._nstep = {p: int(st["step"]) for p, st in self.state.items() if "step" in st}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_cache"):
            self._cache.clear()
            self._last_sig.clear()
            self._snap.clear()

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

    def _signature(self, group):
        """everything a table set depends on: per parameter the addresses of gradient (1 = a buffer of this class), value and
        the two moment tensors, plus the group's switches that decide between the kernel and torch's path.  One cheap host
        pass per step; without the value / state addresses a ``model.float()``, ``p.data = ...``,
        ``load_state_dict(assign=True)`` or a cleared ``optimizer.state`` would leave the kernel writing through stale
        pointers."""
        own, state = self._own_grad, self.state
        sig = []
        add = sig.append
        for p in group["params"]:  # one entry per parameter: None = no gradient, else (gradient, value, exp_avg, exp_avg_sq)
            g = p.grad
            if g is None:
                add(None)
                continue
            g_ptr = self._OWN_PTR if (own and p in own) else g.data_ptr()
            try:
                st = state[p]
                add((g_ptr, p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
            except KeyError:  # no moments yet (first step of this parameter)
                add((g_ptr, p.data_ptr(), None, None))
        return self._hyper(group), tuple(sig)

    _OWN_PTR = "own"  # signature entry of a gradient that is copied into a buffer of this class every step (not an address)

    _OWN = object()  # snapshot marker: the gradient is copied into a buffer of this class every step

    def _snapshot(self, group):
        """what _unchanged compares against: per parameter the OBJECTS the signature's addresses were read from.  Gradients
        are held WEAKLY (the executor retires a persistent gradient buffer that anybody else references, net_exec._shared);
        a dead reference no longer matches anything, so an identity match always means the same live tensor."""
        own, state = self._own_grad, self.state
        snap = []
        for p in group["params"]:
            g = p.grad
            if g is None:
                snap.append(None)
                continue
            st = state.get(p)
            snap.append((self._OWN if p in own else weakref.ref(g), g.data_ptr(), p.data_ptr(), st,
                         st.get("exp_avg") if st else None, st.get("exp_avg_sq") if st else None))
        return snap

    def _unchanged(self, group, hyper, snap) -> bool:
        """the cheap per-step test that the previous step's signature still holds: same gradient / moment tensor objects
        (the executor's gradient slices and the moments are persistent tensors: identity plus an unchanged address, which
        ``set_()`` could move), same value address, same per-parameter state dict.  A third of the cost of rebuilding the
        signature (4 address reads and a tensor-keyed dict lookup per parameter, ~330 parameters); anything that does not
        match falls through to the full signature."""
        params = group["params"]
        if len(params) != len(snap) or hyper != self._hyper(group):
            return False
        OWN, state = self._OWN, self.state
        for p, e in zip(params, snap):
            g = p.grad
            if e is None:
                if g is not None:
                    return False
                continue
            if g is None:
                return False
            g_ref, g_ptr, v_ptr, st, m, v = e
            if g_ref is OWN:
                if not g.is_cuda:
                    return False
            elif g is not g_ref() or g.data_ptr() != g_ptr:
                return False
            if p.data_ptr() != v_ptr or state.get(p) is not st or st is None or st.get("exp_avg") is not m \
                    or st.get("exp_avg_sq") is not v or m is None:
                return False
        return True

    @staticmethod
    def _hyper(group):
        return (bool(group["amsgrad"]), group["weight_decay"], bool(group["maximize"]), bool(group.get("capturable", False)),
                bool(group.get("differentiable", False)))

    @staticmethod
    def _grad_entries(group, flat):
        """parameter -> gradient entry of a signature (per-parameter entries: nothing is recovered from sentinels inside
        the data, so a zero address - an empty gradient - cannot shift the walk)"""
        return {p: e[0] for p, e in zip(group["params"], flat) if e is not None}

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
        if self._skip_ctr:  # tensors are regrouped: their true step counts first (a table's tensors share one skip counter)
            self._fold_skips(params)
        by_step = {}
        for p in params:
            by_step.setdefault((self._nstep[p], p in self._gated), []).append(p)
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

    def _torch_step(self, group, params):
        """torch's own implementation, for this group only (CPU tensors, amsgrad, weight decay, ...)"""
        self._sync_step_tensors()
        saved = self.param_groups
        self.param_groups = [group]
        try:
            super().step()
        finally:
            self.param_groups = saved
        for p in params:
            self._nstep[p] = int(self.state[p]["step"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            own = self._own_grad
            if own:
                moved = [p for p in group["params"] if p in own and p.grad is not None]  # this group's only
                if moved:
                    _copy_many([own[p] for p in moved], [p.grad for p in moved])
            sets = self._cache.setdefault(gi, {})
            prev = self._snap.get(gi)
            if prev is not None and self._unchanged(group, prev[0][0], prev[2]):
                sig, tables = prev[0], prev[1]
            else:
                sig = self._signature(group)
                tables = sets.get(sig)
            if tables is None:
                params = [p for p in group["params"] if p.grad is not None]
                if not params:
                    continue
                # which gradients moved since the previous step?  they get buffers of their own (their tables stay valid)
                last = self._last_sig.get(gi)
                if last is not None:
                    changed = False
                    old_grads = self._grad_entries(group, last[1])
                    new_grads = self._grad_entries(group, sig[1])
                    for p in group["params"]:
                        old, new = old_grads.get(p), new_grads.get(p)
                        if isinstance(old, int) and isinstance(new, int) and old != new and p.grad.is_cuda:
                            own[p] = p.grad.detach().clone()
                            changed = True
                    if changed:
                        sig = self._signature(group)
                        tables = sets.get(sig)
                if tables is None:
                    if not self._native_ok(group, params):
                        self._last_sig[gi] = sig
                        self._torch_step(group, params)
                        continue
                    tables = self._build(group, params)
                    sig = self._signature(group)  # (the moment tensors may have been created by _build)
                    while len(sets) >= self._MAX_TABLE_SETS:
                        sets.pop(next(iter(sets)))
                    sets[sig] = tables
            if prev is None or prev[0] is not sig:
                # the set of tensors that take a step changed: a table set cached under this signature was grouped by the step
                # counts of the time it was built, and some of its tensors may have sat out steps since (round 5: a batch
                # without proposals, then one with - the cached table gave ScoreNet / NPCS-Net the backbone's step number,
                # i.e. the wrong bias corrections, for the rest of the run)
                nstep, ctr_of = self._nstep, self._skip_ctr
                if any(len({(nstep[p], id(ctr_of.get(p))) for p in plist}) > 1 for _t, _f, _b, plist, _pin in tables):
                    tables = self._build(group, [p for p in group["params"] if p.grad is not None])
                    sets[sig] = tables
            self._last_sig[gi] = sig
            if prev is None or prev[0] is not sig:
                self._snap[gi] = (sig, tables, self._snapshot(group))
            L = _C.lib()
            dev = tables[0][3][0].device
            stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))
            beta1, beta2 = group["betas"]
            nstep = self._nstep
            gate = self._gate_fn() if (self._gate_fn is not None and self._gated) else None
            for table, first, blocks, plist, _pinned in tables:
                n = nstep[plist[0]] + 1
                gate_ptr = skip_ptr = None
                ctr = self._skip_ctr.get(plist[0]) if plist[0] in self._gated else None
                if plist[0] in self._gated and (gate is not None or ctr is not None):
                    # (once a step was suppressed the table's step number lags: every later launch goes through the gated entry
                    # point, with an always-open gate when this step has none)
                    if ctr is None:
                        ctr = torch.zeros((1,), dtype=torch.int64, device=dev)
                        for p in plist:
                            self._skip_ctr[p] = ctr
                    if self._gate_open is None:
                        self._gate_open = torch.ones((1,), dtype=torch.int64, device=dev)
                    g_t, g_i = gate if gate is not None else (self._gate_open, 0)
                    gate_ptr, skip_ptr = g_t.data_ptr() + 8 * int(g_i), ctr.data_ptr()
                _C.check(L.gpn_adam_step_gated(ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(first.data_ptr()), len(plist), blocks,
                                               ctypes.c_double(group["lr"]), ctypes.c_double(beta1), ctypes.c_double(beta2),
                                               ctypes.c_double(group["eps"]), ctypes.c_int64(n), ctypes.c_void_p(gate_ptr),
                                               ctypes.c_void_p(skip_ptr), stream), "gpn_adam_step")
                for p in plist:
                    nstep[p] = n
        self._release_executor_gradients()
        return loss

    def _release_executor_gradients(self):
        """tell the sparse U-Nets' executor that this step has consumed their gradients (the update kernels are queued behind
        the backward pass on the stream): the persistent gradient buffers may be overwritten by the next backward pass
        (network/net_exec.py, "gradient hand-over contract")"""
        from .network import net_exec
        progs = self.__dict__.get("_executor_programs")
        if progs is None or progs[0] != net_exec.program_count():
            found = net_exec.programs_of(p for g in self.param_groups for p in g["params"])
            progs = self._executor_programs = (net_exec.program_count(), found)
        for prog in progs[1]:
            prog.release_gradients()

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        if set_to_none:
            self._release_executor_gradients()
