"""Host side of the native layer-program executor (include/gpn.h section U, csrc/net.hip).

The reference runs its sparse U-Net as a Python walk over ~200 spconv / BatchNorm1d modules per forward
(network/backbone.py:40-49, 126-141, 150-155) and lets autograd walk back.  On MI355X the kernels of one layer take
20-40 us, i.e. about as long as the interpreter needs to dispatch them, so the walk itself bounds the step.  This
module flattens the same walk ONCE per network into a program (CONV / BN / CONCAT ops over numbered activation slots)
and runs each forward / backward with a single library call; it is used by ``SparseUNet.forward`` whenever the raw-op
backend is the HIP library.  The module tree (and so the state_dict) is untouched: the program only holds references
to the modules' parameters and buffers.

Results are those of the per-layer path (same kernels, same order); gradients of multiply-consumed activations are
summed in reverse program order.
"""
import ctypes
import os
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _C, backend
from .. import functional as GF
from ..spconv import pytorch as spconv


def rulebook_ptrs(rb):
    """(nbr, nbr_p, perm, pair_src, pair_dst, tile_off) device addresses of a rulebook, 0 = absent.  Rulebooks cut out of a
    prepared batch's arena carry them (hip_ops.ArenaRulebook.ptrs: no tensor views are made for the executor's sake)"""
    p = getattr(rb, "ptrs", None)
    if p is not None:
        return p
    nbr_p, perm = rb.nbr_p, rb.perm
    return (rb.nbr.data_ptr(), 0 if nbr_p is None else nbr_p.data_ptr(), 0 if perm is None else perm.data_ptr(),
            rb.pair_src.data_ptr(), rb.pair_dst.data_ptr(), rb.tile_off.data_ptr())

# numpy mirrors of the C structs in include/gpn.h (sizes are checked against the header in tests/test_cabi.py)
SLOT_DT = np.dtype([("data", "<u8"), ("grad", "<u8"), ("rows", "<i8"), ("channels", "<i4"), ("grad_state", "<i4"),
                    ("rows_dev", "<u8"), ("rows_plan", "<i8")])
RB_DT = np.dtype([("nbr", "<u8"), ("nbr_t", "<u8"), ("nbr_p", "<u8"), ("perm", "<u8"), ("nbr_t_p", "<u8"), ("perm_t", "<u8"),
                  ("pair_src", "<u8"), ("pair_dst", "<u8"), ("tile_off", "<u8"),
                  ("n_src", "<i8"), ("n_dst", "<i8"), ("K", "<i4"), ("reverse_taps", "<i4")])
CONV_DT = np.dtype([("W", "<u8"), ("dW", "<u8"), ("cin", "<i4"), ("cout", "<i4")])
BN_DT = np.dtype([("weight", "<u8"), ("bias", "<u8"), ("running_mean", "<u8"), ("running_var", "<u8"),
                  ("save_mean", "<u8"), ("save_invstd", "<u8"), ("dweight", "<u8"), ("dbias", "<u8"),
                  ("eps", "<f4"), ("momentum", "<f4"), ("C", "<i4"), ("reserved", "<i4")])
OP_DT = np.dtype([("kind", "<i4"), ("src0", "<i4"), ("src1", "<i4"), ("dst", "<i4"), ("rulebook", "<i4"),
                  ("param", "<i4"), ("flags", "<i4"), ("reserved", "<i4")])
OP_CONV, OP_BN, OP_CONCAT = 0, 1, 2
FLAG_RELU = 1


class Unsupported(Exception):
    """the network contains something the program cannot express; the caller falls back to the per-layer path"""


def _bn_ok(bn) -> bool:
    return (isinstance(bn, nn.BatchNorm1d) and bn.affine and bn.track_running_stats and bn.momentum is not None
            and bn.num_features % 4 == 0)


class NetProgram:
    """static description of one SparseUNet: ops, slots (level, channels), and the modules whose tensors they use."""

    def __init__(self, unet):
        from .backbone import ResBlock, UBlock  # local: backbone imports this module
        self._ResBlock, self._UBlock = ResBlock, UBlock
        self.ops: List[tuple] = []
        self.slot_level: List[int] = [0]
        self.slot_channels: List[int] = [-1]  # slot 0 = the input features (channels filled below)
        self.convs: List[nn.Module] = []
        self.bns: List[nn.Module] = []
        self.rb_keys: List[tuple] = []  # ("subm", level) | ("down", level) | ("inv", level) | ("ident", level)
        self.level_keys = {}  # level -> (subm indice_key, down indice_key)
        self.python_stem_conv = None
        cur = 0
        stem = unet.stem
        if stem is not None:
            mods = list(stem._modules.values())
            if len(mods) == 3 and isinstance(mods[0], spconv.SubMConv3d) and isinstance(mods[2], nn.ReLU):
                conv = mods[0]
                if conv.kernel_size != [3, 3, 3] or conv.bias is not None:
                    raise Unsupported("stem conv")
                self.level_keys[0] = [conv.indice_key, None]
                if conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0:
                    self.slot_channels[0] = conv.in_channels
                    cur = self._conv(conv, ("subm", 0), cur, 0)
                else:
                    self.python_stem_conv = conv  # 6-channel input: padded per-layer path, then the program
                    self.slot_channels[0] = conv.out_channels
                cur = self._bn(mods[1], cur, relu=True)
            elif len(mods) == 2 and isinstance(mods[1], nn.ReLU):
                self.slot_channels[0] = mods[0].num_features
                cur = self._bn(mods[0], cur, relu=True)
            else:
                raise Unsupported("stem layout")
        else:
            self.slot_channels[0] = unet.ublock.channels[0]
        self.out_slot = self._ublock(unet.ublock, 0, cur)
        self.n_levels = max(self.slot_level) + 1
        for lvl in range(self.n_levels):
            keys = self.level_keys.get(lvl)
            if keys is None or keys[0] is None or (lvl + 1 < self.n_levels and keys[1] is None):
                raise Unsupported("indice keys")
        # static tables
        self.ops_np = np.zeros(len(self.ops), OP_DT)
        for i, (kind, s0, s1, dst, rb, param, flags) in enumerate(self.ops):
            self.ops_np[i] = (kind, s0, s1, dst, rb, param, flags, 0)
        self.slot_level_np = np.asarray(self.slot_level, np.int64)
        self.slot_channels_np = np.asarray(self.slot_channels, np.int64)
        self.bn_C = np.asarray([bn.num_features for bn in self.bns], np.int64)
        # hyper-parameters of the modules, read once (they are not changed while a program is cached: invalidate() otherwise)
        self.bn_eps = [bn.eps for bn in self.bns]
        self.bn_momentum = [bn.momentum for bn in self.bns]
        self.conv_cin = [c.in_channels for c in self.convs]
        self.conv_cout = [c.out_channels for c in self.convs]
        self.bn_off = np.concatenate([[0], np.cumsum(self.bn_C)])  # float offsets into the flat per-BN vectors
        self.conv_numel = np.asarray([c.weight.numel() for c in self.convs], np.int64)
        self.conv_off = np.concatenate([[0], np.cumsum(self.conv_numel)])
        self.conv_ops = [(i, op) for i, op in enumerate(self.ops) if op[0] == OP_CONV]
        self.grad_sizes = [int(v) for v in self.conv_numel] + [int(v) for v in self.bn_C] * 2  # flat gradient buffer layout
        self.grad_total = int(sum(self.grad_sizes))
        self.last_pgrad = None
        self._grad_cache = {}
        self.retired_total = 0     # persistent gradient buffers replaced because their gradients were never released (grad_buffer)
        self.retired_in_a_row = 0
        self.grad_generation = 0   # backward passes that handed out the persistent gradient buffer ...
        self.grad_released = 0     # ... and the last of them whose gradients the consumer has acknowledged (release_gradients)
        self._static_tables = None
        self._anchor = torch.zeros(1, requires_grad=True)  # see _NetFn
        self.signature = self._signature(unet)
        global _PROGRAM_COUNT
        _PROGRAMS.add(self)
        _PROGRAM_COUNT += 1

    # ------------------------------------------------------------------ program construction
    def _new_slot(self, level, channels):
        self.slot_level.append(level)
        self.slot_channels.append(channels)
        return len(self.slot_level) - 1

    def _rb(self, key):
        if key not in self.rb_keys:
            self.rb_keys.append(key)
        return self.rb_keys.index(key)

    def _conv(self, conv, rb_key, src, dst_level):
        if conv.bias is not None or conv.in_channels % 16 or conv.out_channels % 16:
            raise Unsupported("conv shape / bias")
        if self.slot_channels[src] != conv.in_channels:
            raise Unsupported("channel mismatch")
        dst = self._new_slot(dst_level, conv.out_channels)
        self.convs.append(conv)
        self.ops.append((OP_CONV, src, -1, dst, self._rb(rb_key), len(self.convs) - 1, 0))
        return dst

    def _bn(self, bn, src, relu, res=-1):
        if not _bn_ok(bn) or bn.num_features != self.slot_channels[src]:
            raise Unsupported("norm layer")
        dst = self._new_slot(self.slot_level[src], bn.num_features)
        self.bns.append(bn)
        self.ops.append((OP_BN, src, res, dst, -1, len(self.bns) - 1, FLAG_RELU if relu else 0))
        return dst

    def _conv_norm(self, seq, rb_key, src, relu, res=-1):
        mods = list(seq._modules.values())
        if len(mods) != 2 or not isinstance(mods[0], spconv.SubMConv3d):
            raise Unsupported("conv-norm pair")
        want = [1, 1, 1] if rb_key[0] == "ident" else [3, 3, 3]
        if mods[0].kernel_size != want:
            raise Unsupported("kernel size")
        return self._bn(mods[1], self._conv(mods[0], rb_key, src, self.slot_level[src]), relu, res)

    def _resblock(self, blk, level, src):
        if type(blk) is not self._ResBlock:
            raise Unsupported("block type")
        keys = self.level_keys.setdefault(level, [None, None])
        key = blk.conv1[0].indice_key
        if key is None or blk.conv2[0].indice_key != key or (keys[0] not in (None, key)):
            raise Unsupported("indice key")
        keys[0] = key
        if isinstance(blk.shortcut, nn.Identity):
            skip = src
        else:
            skip = self._conv_norm(blk.shortcut, ("ident", level), src, relu=False)
        y = self._conv_norm(blk.conv1, ("subm", level), src, relu=True)
        return self._conv_norm(blk.conv2, ("subm", level), y, relu=True, res=skip)

    def _ublock(self, ub, level, src):
        if type(ub) is not self._UBlock:
            raise Unsupported("ublock type")
        for blk in ub.encoder_blocks._modules.values():
            src = self._resblock(blk, level, src)
        if len(ub.channels) == 1:
            return src
        skip = src
        down = list(ub.downsample._modules.values())
        up = list(ub.upsample._modules.values())
        if not (len(down) == 3 and isinstance(down[0], spconv.SparseConv3d) and isinstance(down[2], nn.ReLU)
                and len(up) == 3 and isinstance(up[0], spconv.SparseInverseConv3d) and isinstance(up[2], nn.ReLU)
                and down[0].indice_key is not None and down[0].indice_key == up[0].indice_key):
            raise Unsupported("down / up layout")
        self.level_keys[level][1] = down[0].indice_key
        d = self._bn(down[1], self._conv(down[0], ("down", level), src, level + 1), relu=True)
        d = self._ublock(ub.ublock, level + 1, d)
        u = self._bn(up[1], self._conv(up[0], ("inv", level), d, level), relu=True)
        cat = self._new_slot(level, self.slot_channels[u] + self.slot_channels[skip])
        self.ops.append((OP_CONCAT, u, skip, cat, -1, -1, 0))
        for blk in ub.decoder_blocks._modules.values():
            cat = self._resblock(blk, level, cat)
        return cat

    @staticmethod
    def _signature(unet):
        """changes when modules are replaced (not when their tensors are updated in place)"""
        return tuple(id(m) for m in unet.modules())

    # ------------------------------------------------------------------ per-call state
    def rulebooks_dev(self, x):
        """``rulebooks`` for a tensor whose row count is a device counter (``x.rows_dev``, hip_ops.DevCount; include/gpn.h
        section DEV): no host read anywhere - every level's buffers at the bound of level 0, the coarse levels' row counts
        device counters of their own (plans from ``x.level_plans``).  -> (rows = the bounds, table, objs, levels, per-level
        DevCounts)"""
        ops = backend.raw()
        dev_counts = [x.rows_dev]
        levels = [(x.indices, list(x.spatial_shape))]
        plans = list(getattr(x, "level_plans", None) or [])
        subm, down = [], []
        for lvl in range(self.n_levels):
            idx, shape = levels[lvl]
            skey, dkey = self.level_keys[lvl]
            rb = x.indice_dict.get(skey)
            if rb is None:
                rb = ops.rulebook_subm3(idx, shape, rows=dev_counts[lvl])
                x.indice_dict[skey] = rb
            subm.append(rb)
            if lvl + 1 < self.n_levels:
                rec = x.indice_dict.get(dkey)
                if rec is None:
                    out_idx, out_shape, rb_fwd, rb_bwd, out_rows = ops.rulebook_down_dev(
                        idx, shape, x.batch_size, dev_counts[lvl], getattr(x, "batch_dev", None),
                        out_plan=plans[lvl] if lvl < len(plans) else 0)
                    rec = spconv._DownRecord(idx, shape, out_idx, out_shape, rb_fwd, rb_bwd)
                    rec.out_rows = out_rows
                    x.indice_dict[dkey] = rec
                down.append(rec)
                levels.append((rec.out_indices, rec.out_shape))
                dev_counts.append(rec.out_rows)
        rows = np.asarray([lv[0].shape[0] for lv in levels], np.int64)
        table = np.zeros(len(self.rb_keys), RB_DT)
        objs = []
        for i, (kind, lvl) in enumerate(self.rb_keys):
            if kind == "subm":
                rb, rb_t, rev = subm[lvl], subm[lvl], 1
            elif kind == "down":
                rb, rb_t, rev = down[lvl].rb_fwd, down[lvl].rb_bwd, 0
            elif kind == "inv":
                rb, rb_t, rev = down[lvl].rb_bwd, down[lvl].rb_fwd, 0
            else:
                ikey = f"__identity_dev_{lvl}__"
                rb = x.indice_dict.get(ikey)
                if rb is None:
                    rb = ops.rulebook_identity(int(rows[lvl]), x.features.device, rows_dev=dev_counts[lvl])
                    x.indice_dict[ikey] = rb
                rb_t, rev = rb, 0
            table[i] = (rb.nbr.data_ptr(), rb_t.nbr.data_ptr(), 0, 0, 0, 0, rb.pair_src.data_ptr(), rb.pair_dst.data_ptr(),
                        rb.tile_off.data_ptr(), rb.n_src, rb.n_dst, rb.K, rev)
            objs.append((rb, rb_t))
        return rows, table, objs, levels, dev_counts

    def ident_levels(self) -> int:
        """bit l set: level l has k = 1 convs (the decoder blocks' shortcuts) and needs an identity rulebook"""
        mask = 0
        for kind, lvl in self.rb_keys:
            if kind == "ident":
                mask |= 1 << lvl
        return mask

    def adopt(self, x, prepared_levels):
        """fill ``x.indice_dict`` from a finished native batch preparation (hip_ops.PreparedBackbone.wrap()["levels"]) under the
        keys ``rulebooks`` uses, so that it finds everything built"""
        for lvl, lv in enumerate(prepared_levels):
            skey, dkey = self.level_keys[lvl]
            x.indice_dict[skey] = lv["subm"]
            if lvl + 1 < self.n_levels:
                nxt = prepared_levels[lvl + 1]
                x.indice_dict[dkey] = spconv._DownRecord(lv["indices"], lv["shape"], nxt["indices"], nxt["shape"], lv["down_fwd"],
                                                         lv["down_bwd"])
            if lv["ident"] is not None:
                x.indice_dict[f"__identity_{int(lv['indices'].shape[0])}__"] = lv["ident"]

    def rulebooks(self, x):
        """fetch / build the rulebook of every level through the tensor's indice_dict (same keys as the modules)"""
        ops = backend.raw()
        levels = [(x.indices, list(x.spatial_shape))]
        subm, down = [], []
        # row counts of the coarse levels still to be built: known to the producer of the tensor (the proposal stage reports
        # its grid's in its one read), else fetched for all levels with ONE read instead of one per level
        known = list(getattr(x, "level_counts", None) or [])
        missing = [lvl for lvl in range(self.n_levels - 1) if x.indice_dict.get(self.level_keys[lvl][1]) is None]
        if missing and len(known) < self.n_levels - 1 and missing[0] == 0 and hasattr(ops, "rulebook_level_counts") \
                and x.indices.is_cuda and x.indices.shape[0] > 0 and self.n_levels > 2:
            known = ops.rulebook_level_counts(x.indices, list(x.spatial_shape), x.batch_size, self.n_levels - 1).tolist()
        for lvl in range(self.n_levels):
            idx, shape = levels[lvl]
            skey, dkey = self.level_keys[lvl]
            rb = x.indice_dict.get(skey)
            if rb is None:
                rb = ops.rulebook_subm3(idx, shape)
                x.indice_dict[skey] = rb
            subm.append(rb)
            if lvl + 1 < self.n_levels:
                rec = x.indice_dict.get(dkey)
                if rec is None:
                    if lvl < len(known) and idx.is_cuda:
                        out_idx, out_shape, rb_fwd, rb_bwd = ops.rulebook_down(idx, shape, x.batch_size, n_out=known[lvl])
                    else:
                        out_idx, out_shape, rb_fwd, rb_bwd = ops.rulebook_down(idx, shape, x.batch_size)
                    rec = spconv._DownRecord(idx, shape, out_idx, out_shape, rb_fwd, rb_bwd)
                    x.indice_dict[dkey] = rec
                down.append(rec)
                levels.append((rec.out_indices, rec.out_shape))
        rows = np.asarray([lv[0].shape[0] for lv in levels], np.int64)
        table = np.zeros(len(self.rb_keys), RB_DT)
        objs = []
        for i, (kind, lvl) in enumerate(self.rb_keys):
            if kind == "subm":
                rb, rb_t, rev = subm[lvl], subm[lvl], 1
            elif kind == "down":
                rb, rb_t, rev = down[lvl].rb_fwd, down[lvl].rb_bwd, 0
            elif kind == "inv":
                rb, rb_t, rev = down[lvl].rb_bwd, down[lvl].rb_fwd, 0
            else:
                n = int(rows[lvl])
                ikey = f"__identity_{n}__"
                rb = x.indice_dict.get(ikey)
                if rb is None:
                    rb = spconv._identity_rulebook(n, x.features.device)
                    x.indice_dict[ikey] = rb
                rb_t, rev = rb, 0
            nbr, nbr_p, perm, pair_src, pair_dst, tile_off = rulebook_ptrs(rb)
            nbr_t, nbr_p_t, perm_t = (nbr, nbr_p, perm) if rb_t is rb else rulebook_ptrs(rb_t)[:3]
            table[i] = (nbr, nbr_t, nbr_p, perm, nbr_p_t, perm_t, pair_src, pair_dst, tile_off, rb.n_src, rb.n_dst, rb.K, rev)
            objs.append((rb, rb_t))
        return rows, table, objs, levels

    def params(self):
        # (straight from the modules' parameter dicts: nn.Module.__getattr__ costs ~0.25 us per access, ~1000 accesses a step)
        return ([c._parameters["weight"] for c in self.convs] + [b._parameters["weight"] for b in self.bns] +
                [b._parameters["bias"] for b in self.bns])

    def buffers(self, name):
        return [b._buffers[name] for b in self.bns]

    def static_tables(self, params):
        """-> fresh copies of the weight / BatchNorm tables with everything filled in that does not change from call to call
        (parameter and buffer addresses, shapes, eps / momentum).  Built once and re-used while every parameter and running
        statistic still lives at the address it was built from (one address read per tensor per call instead of a dozen
        list -> array conversions: 70 -> 30 us per pass, and the proposal networks' passes are issued while the GPU waits)."""
        run_mean, run_var = self.buffers("running_mean"), self.buffers("running_var")
        sig = [p.data_ptr() for p in params]
        sig += [t.data_ptr() for t in run_mean]
        sig += [t.data_ptr() for t in run_var]
        cached = self._static_tables
        if cached is None or cached[0] != sig:
            n_conv, n_bn = len(self.convs), len(self.bns)
            conv_table = np.zeros(n_conv, CONV_DT)
            conv_table["W"] = sig[:n_conv]
            conv_table["cin"] = self.conv_cin
            conv_table["cout"] = self.conv_cout
            bn_table = np.zeros(n_bn, BN_DT)
            bn_table["weight"] = sig[n_conv:n_conv + n_bn]
            bn_table["bias"] = sig[n_conv + n_bn:n_conv + 2 * n_bn]
            bn_table["running_mean"] = sig[n_conv + 2 * n_bn:n_conv + 3 * n_bn]
            bn_table["running_var"] = sig[n_conv + 3 * n_bn:]
            bn_table["eps"] = self.bn_eps
            bn_table["momentum"] = self.bn_momentum
            bn_table["C"] = self.bn_C
            cached = self._static_tables = (sig, conv_table, bn_table)
        return cached[1].copy(), cached[2].copy()

    def release_gradients(self):
        """the consumer of the last backward pass's gradients is done with them (see grad_buffer)"""
        self.grad_released = self.grad_generation

    def grad_buffer(self, device, params, fresh: bool):
        """-> (flat fp32 buffer of grad_total elements, per-parameter views of it in params() order).  ``fresh`` asks for a
        private buffer (gradient accumulation into existing ``.grad``, parameters as autograd inputs).  Otherwise the
        PERSISTENT pair of this device is handed out - created once, overwritten by every backward pass - under an explicit
        contract (round 5; it replaces reference-count heuristics): every hand-out is a generation, and the pair is reused
        only when the previous generation was RELEASED by whoever consumed the gradients (``release_gradients()``: called by
        FusedAdam.step / .zero_grad - bench.py's loop runs on that - and by the Trainer after ``optimizer.step()`` whatever the
        optimizer; ``net_exec.release_gradients(model)`` for any other loop).  Without a release the previous pair is retired - it stays alive, untouched, with
        whoever holds its views - and a new one is made: correct for any training loop, one allocation per pass slower.
        After a release the views are overwritten in place by the next backward pass, exactly like ``.grad`` tensors under
        ``zero_grad(set_to_none=False)``: clone what you keep."""
        def make():
            flat = torch.empty((self.grad_total,), dtype=torch.float32, device=device)
            pieces = flat.split(self.grad_sizes)
            return flat, [piece.view_as(p) for piece, p in zip(pieces, params)]
        if fresh:
            return make()
        cached = self._grad_cache.get(device)
        if cached is None or any(v.shape != p.shape for v, p in zip(cached[1], params)):
            cached = make()
            self._grad_cache[device] = cached
            self.retired_in_a_row = 0
        elif self.grad_released != self.grad_generation:
            cached = make()
            self._grad_cache[device] = cached
            # a buffer retired on EVERY pass: every gradient address then moves every step - FusedAdam re-learns its table,
            # GradSync loses the in-place exchange
            self.retired_total += 1
            self.retired_in_a_row += 1
            if self.retired_in_a_row == 8:
                print("[gapartnet_amd] the persistent gradient buffer of a U-Net was replaced on 8 backward passes in a row: "
                      "nobody acknowledges its gradients - call gapartnet_amd.network.net_exec.release_gradients(model) "
                      "after optimizer.step() (FusedAdam and the Trainer do), or call net_exec.set_autograd_parameters(True)")
        else:
            self.retired_in_a_row = 0
        self.grad_generation += 1
        return cached


# every live program, for optimizers that acknowledge consumed gradients (NetProgram.release_gradients)
import weakref
_PROGRAMS = weakref.WeakSet()
_PROGRAM_COUNT = 0


def program_count() -> int:
    """programs created so far (a cheap 'has the set changed' key for callers that cache programs_of())"""
    return _PROGRAM_COUNT


def programs_of(params):
    """the live programs that hand gradients to any of ``params``"""
    ids = {id(p) for p in params}
    return [prog for prog in list(_PROGRAMS) if any(id(p) in ids for p in prog.params())]


def release_gradients(module: nn.Module):
    """acknowledge, for every sparse U-Net inside ``module``, that the gradients of its last backward pass have been consumed
    (optimizer step done / gradients dropped): the executor may overwrite its persistent gradient buffer in the next backward
    pass.  FusedAdam.step() / .zero_grad() and gapartnet_amd.trainer call this themselves; a training loop around another
    optimizer calls it after ``optimizer.step()`` - or not at all, and pays one gradient-buffer allocation per U-Net and step."""
    for m in module.modules():
        prog = m.__dict__.get("_net_program")
        if prog:
            prog.release_gradients()


def _vp(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def _call(fn_name, prog, slots, rb_table, conv_table, bn_table, extra, device):
    from ..hip_ops import _fast_ws
    L = _C.lib()
    fn = getattr(L, fn_name)
    ws_ptr, ws_size, stream = _fast_ws(device)
    args = (_vp(prog.ops_np), len(prog.ops_np), _vp(slots), len(slots), _vp(rb_table), len(rb_table), _vp(conv_table),
            len(conv_table), _vp(bn_table), len(bn_table)) + extra
    rc = fn(*args, ctypes.c_void_p(ws_ptr), ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    if rc == 2:  # workspace too small: size it from the library's own estimate and retry
        need = L.gpn_net_ws_bytes(_vp(prog.ops_np), len(prog.ops_np), _vp(slots), len(slots), _vp(rb_table),
                                  _vp(conv_table))
        ws_ptr, ws_size, stream = _fast_ws(device, int(need))
        rc = fn(*args, ctypes.c_void_p(ws_ptr), ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    if rc:
        raise _C.GpnError(f"{fn_name} failed: {L.gpn_last_error().decode('utf-8', 'replace')}")


# Gradient hand-over contract of the executor (read this before relying on autograd features for U-Net parameters):
#   default ("direct") form - the ~110 parameters of a U-Net are NOT autograd inputs of the program.  backward() writes all
#   their gradients into ONE persistent flat buffer per program and assigns each parameter's ``.grad`` to its slice.
#   Consequences: (1) ``torch.autograd.grad(loss, unet_parameters)`` raises "not used in the graph" and
#   ``backward(inputs=[...])`` still fills the parameters' ``.grad``; (2) tensor hooks / post-accumulate-grad hooks on these
#   parameters would not fire - a parameter WITH such a hook switches the call to the autograd form automatically;
#   (3) the buffer is reused by the next backward once the consumer of the gradients has RELEASED them (an explicit
#   acknowledgement: NetProgram.release_gradients, see grad_buffer; FusedAdam / the Trainer give it) - a reference kept past
#   that point sees the next step's gradients, as a ``.grad`` kept past ``zero_grad(set_to_none=False)`` does; (4) parameters with ``requires_grad=False`` get
#   no gradient and their weight-gradient launches are skipped.  GradSync and FusedAdam build on the flat buffer.
#   autograd form - ``set_autograd_parameters(True)``: parameters are autograd inputs,
#   gradients come back through AccumulateGrad like any other op's (everything of (1)-(2) works; ~3 ms of host time per
#   training step for the three U-Nets, and no in-place gradient exchange).
_AUTOGRAD_PARAMS = False


def set_autograd_parameters(on: bool) -> bool:
    """route U-Net parameters through autograd (True) or hand their gradients over directly (False, default); returns the
    previous setting"""
    global _AUTOGRAD_PARAMS
    prev, _AUTOGRAD_PARAMS = _AUTOGRAD_PARAMS, bool(on)
    return prev


def _has_hooks(p) -> bool:
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def _call_pair(fn_name, prog, slots_a, slots_b, rb_table, conv_a, conv_b, bn_a, bn_b, extra, device):
    from ..hip_ops import _fast_ws
    L = _C.lib()
    fn = getattr(L, fn_name)
    ws_ptr, ws_size, stream = _fast_ws(device)
    args = (_vp(prog.ops_np), len(prog.ops_np), _vp(slots_a), _vp(slots_b), len(slots_a), _vp(rb_table), len(rb_table),
            _vp(conv_a), _vp(conv_b), len(conv_a), _vp(bn_a), _vp(bn_b), len(bn_a)) + extra
    rc = fn(*args, ctypes.c_void_p(ws_ptr), ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    if rc == 2:  # workspace too small: twice the library's estimate for one network
        need = 2 * L.gpn_net_ws_bytes(_vp(prog.ops_np), len(prog.ops_np), _vp(slots_a), len(slots_a), _vp(rb_table), _vp(conv_a))
        ws_ptr, ws_size, stream = _fast_ws(device, int(need))
        rc = fn(*args, ctypes.c_void_p(ws_ptr), ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    if rc:
        raise _C.GpnError(f"{fn_name} failed: {L.gpn_last_error().decode('utf-8', 'replace')}")


def _forward_tables(features, prog: NetProgram, rows, lvl_dev=None):
    """activation arena, slot / weight / BatchNorm tables of one forward pass of ``prog`` over ``features``; ``lvl_dev``: per
    level a hip_ops.DevCount when the row counts are device counters (``rows`` then holds the bounds)"""
    params = prog.params()
    dev = features.device
    n_slots = len(prog.slot_level)
    slot_rows = rows[prog.slot_level_np]
    sizes = slot_rows * prog.slot_channels_np
    sizes[0] = 0  # slot 0 is the caller's tensor
    offs = np.concatenate([[0], np.cumsum(sizes)])
    arena = torch.empty((int(offs[-1]),), dtype=torch.float32, device=dev)
    base = arena.data_ptr()
    slots = np.zeros(n_slots, SLOT_DT)
    slots["data"] = base + offs[:-1] * 4
    slots["data"][0] = features.data_ptr()
    slots["rows"] = slot_rows
    slots["channels"] = prog.slot_channels_np
    if lvl_dev is not None:
        slots["rows_dev"] = np.asarray([c.t.data_ptr() for c in lvl_dev], np.uint64)[prog.slot_level_np]
        slots["rows_plan"] = np.asarray([c.plan for c in lvl_dev], np.int64)[prog.slot_level_np]
    total_c = int(prog.bn_off[-1])
    stats = torch.empty((2, total_c), dtype=torch.float32, device=dev)
    conv_table, bn_table = prog.static_tables(params)
    bn_table["save_mean"] = stats.data_ptr() + prog.bn_off[:-1] * 4
    bn_table["save_invstd"] = stats.data_ptr() + (total_c + prog.bn_off[:-1]) * 4
    o = prog.out_slot
    out = arena[int(offs[o]):int(offs[o + 1])].view(int(slot_rows[o]), int(prog.slot_channels_np[o]))
    return out, (features, arena, stats, slots, conv_table, bn_table, sizes, params)


def _log_forward(prog, rb_objs):
    if GF.CONV_LOG is not None:
        for _, op in prog.conv_ops:
            conv, (rb, _rb_t) = prog.convs[op[5]], rb_objs[op[4]]
            GF._log(rb, conv.in_channels, conv.out_channels, "fwd")


NET_INFERENCE = 2  # include/gpn.h GPN_NET_INFERENCE


class _Mode(int):
    """the ``training`` argument of the executor ops: truthy = batch statistics (1); falsy = running statistics - 0, or the
    INFERENCE form of 0 (``inference`` set) when the op was applied with gradients disabled, i.e. no backward pass can follow:
    gpn_net_forward is then called with GPN_NET_INFERENCE and applies every conv's BatchNorm in the conv launch (no BatchNorm
    launches; the conv's own output is not kept).  Decided by the CALLER of ``apply`` (inside ``forward`` autograd has always
    switched gradients off, and ``needs_input_grad`` reports the inputs' flags whatever the mode)."""
    inference = False


def _mode(training: bool) -> "_Mode":
    m = _Mode(1 if training else 0)
    m.inference = (not training) and not torch.is_grad_enabled()
    return m


def _forward_mode(training) -> int:
    if training:
        return 1
    return NET_INFERENCE if getattr(training, "inference", False) else 0


def _forward_impl(ctx, features, prog: NetProgram, rt, training):
    rows, rb_table, rb_objs, lvl_dev = rt
    features = features.contiguous()
    out, state = _forward_tables(features, prog, rows, lvl_dev)
    _call("gpn_net_forward", prog, state[3], rb_table, state[4], state[5], (_forward_mode(training),), features.device)
    _log_forward(prog, rb_objs)
    ctx.prog, ctx.rt, ctx.training = prog, rt, training
    ctx.state = state
    return out


def _backward_tables(prog: NetProgram, state, dout, fresh: bool):
    """gradient arena and the tables of one backward pass; -> (garena, slots, conv_table, bn_table, pgrad, views)"""
    features, arena, stats, slots, conv_table, bn_table, sizes, params = state
    dev = features.device
    gsizes = sizes.copy()
    gsizes[0] = features.numel()
    gsizes[prog.out_slot] = 0  # the incoming gradient is used in place
    goffs = np.concatenate([[0], np.cumsum(gsizes)])
    garena = torch.empty((int(goffs[-1]),), dtype=torch.float32, device=dev)
    slots = slots.copy()
    slots["grad"] = garena.data_ptr() + goffs[:-1] * 4
    slots["grad_state"] = 0
    slots["grad"][prog.out_slot] = dout.data_ptr()
    slots["grad_state"][prog.out_slot] = 1
    n_conv = len(prog.convs)
    total_w, total_c = int(prog.conv_off[-1]), int(prog.bn_off[-1])
    pgrad, views = prog.grad_buffer(dev, params, fresh)
    pbase = pgrad.data_ptr()
    conv_table = conv_table.copy()
    conv_table["dW"] = pbase + prog.conv_off[:-1] * 4
    frozen = [i for i, p in enumerate(params[:n_conv]) if not p.requires_grad]
    if frozen:
        conv_table["dW"][frozen] = 0  # no weight-gradient launch for a frozen conv
    bn_table = bn_table.copy()
    bn_table["dweight"] = pbase + (total_w + prog.bn_off[:-1]) * 4
    bn_table["dbias"] = pbase + (total_w + total_c + prog.bn_off[:-1]) * 4
    return garena, slots, conv_table, bn_table, pgrad, views


def _log_backward(prog, rb_objs, need_in):
    if GF.CONV_LOG is not None:
        for _, op in prog.conv_ops:
            conv, (rb, rb_t) = prog.convs[op[5]], rb_objs[op[4]]
            if op[1] != 0 or need_in:
                GF._log(rb_t, conv.out_channels, conv.in_channels, "dgrad")
            if conv.weight.requires_grad:
                GF._log(rb, conv.in_channels, conv.out_channels, "wgrad")


def _backward_impl(ctx, dout, fresh: bool, need_in: bool):
    """runs gpn_net_backward; -> (din or None, flat parameter-gradient buffer, its per-parameter views, params)"""
    prog = ctx.prog
    rows, rb_table, rb_objs, _lvl_dev = ctx.rt
    features, params = ctx.state[0], ctx.state[-1]
    dout = dout.contiguous()
    garena, slots, conv_table, bn_table, pgrad, views = _backward_tables(prog, ctx.state, dout, fresh)
    _call("gpn_net_backward", prog, slots, rb_table, conv_table, bn_table,
          (1 if ctx.training else 0, 1 if need_in else 0), features.device)
    _log_backward(prog, rb_objs, need_in)
    din = garena[:features.numel()].view_as(features) if need_in else None
    ctx.state = None
    return din, pgrad, views, params


def _hand_over(prog, params, views, pgrad, fresh):
    """assign / add the parameter gradients of one network (see the contract above)"""
    if fresh:
        for p, g in zip(params, views):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        prog.last_pgrad = None
    else:
        for p, g in zip(params, views):
            if p.requires_grad:
                p.grad = g
        prog.last_pgrad = pgrad  # grad_sync all-reduces this buffer in place (its slices are the parameters' .grad)


class _NetFn(torch.autograd.Function):
    """the whole program as one differentiable op, direct gradient hand-over (see the contract above).  Only the features
    (and an anchor, below) are autograd inputs; the ~110 parameters of a U-Net are not: backward() hands each of them its
    slice of ONE flat gradient buffer by assigning ``.grad`` directly.  With the parameters as autograd inputs every
    backward pass cost ~3 dispatches per parameter (slice views, AccumulateGrad's detach) - ~1000 per training step for the
    three U-Nets, 3 ms of host time (tools/op_count.py) for handing over buffers that already exist.  The flat buffer and
    the per-parameter views are allocated once per program and reused while ``.grad`` is None at backward time
    (optimizer.zero_grad(set_to_none=True), the default) and nobody else holds them; a parameter that still holds a
    gradient gets the new one added, as autograd would.
    ``anchor`` is a one-element leaf that requires grad: it makes autograd call backward() even when the input features
    do not require a gradient (the backbone's voxel features)."""

    @staticmethod
    def forward(ctx, features, anchor, prog: NetProgram, rt, training):
        return _forward_impl(ctx, features, prog, rt, training)

    @staticmethod
    def backward(ctx, dout):
        prog = ctx.prog
        params = ctx.state[-1]
        fresh = any(p.grad is not None for p in params)  # accumulation into existing gradients: temporary buffer, then add
        din, pgrad, views, params = _backward_impl(ctx, dout, fresh, bool(ctx.needs_input_grad[0]))
        _hand_over(prog, params, views, pgrad, fresh)
        return din, None, None, None, None


class _NetPairFn(torch.autograd.Function):
    """two structurally identical programs over the same input and rulebooks as ONE differentiable op with two outputs
    (gpn_net_forward_pair / gpn_net_backward_pair: layer i of both networks per launch).  Gradient hand-over as in _NetFn.
    If only one output received a gradient (a loss term was absent), that network alone runs its single backward pass."""

    @staticmethod
    def forward(ctx, features, anchor, prog_a: NetProgram, prog_b: NetProgram, rt, training):
        rows, rb_table, rb_objs, lvl_dev = rt
        features = features.contiguous()
        out_a, state_a = _forward_tables(features, prog_a, rows, lvl_dev)
        out_b, state_b = _forward_tables(features, prog_b, rows, lvl_dev)
        _call_pair("gpn_net_forward_pair", prog_a, state_a[3], state_b[3], rb_table, state_a[4], state_b[4], state_a[5],
                   state_b[5], (_forward_mode(training),), features.device)
        _log_forward(prog_a, rb_objs)
        _log_forward(prog_b, rb_objs)
        ctx.progs, ctx.rt, ctx.training = (prog_a, prog_b), rt, training
        ctx.states = (state_a, state_b)
        ctx.set_materialize_grads(False)  # an output that does not reach the loss arrives as None in backward, not as zeros
        return out_a, out_b

    @staticmethod
    def backward(ctx, dout_a, dout_b):
        rows, rb_table, rb_objs, _lvl_dev = ctx.rt
        need_in = bool(ctx.needs_input_grad[0])
        features = ctx.states[0][0]
        dev = features.device
        douts = (dout_a, dout_b)
        live = [t for t in (0, 1) if douts[t] is not None]
        if not live:
            ctx.states = None
            return None, None, None, None, None, None
        tabs = {}
        for t in live:
            prog, params = ctx.progs[t], ctx.states[t][-1]
            fresh = any(p.grad is not None for p in params)
            tabs[t] = (fresh,) + _backward_tables(prog, ctx.states[t], douts[t].contiguous(), fresh)
        extra = (1 if ctx.training else 0, 1 if need_in else 0)
        if len(live) == 2:
            (_, _, slots_a, conv_a, bn_a, _, _), (_, _, slots_b, conv_b, bn_b, _, _) = tabs[0], tabs[1]
            _call_pair("gpn_net_backward_pair", ctx.progs[0], slots_a, slots_b, rb_table, conv_a, conv_b, bn_a, bn_b, extra, dev)
        else:
            for t in live:
                _, _, slots, conv_t, bn_t, _, _ = tabs[t]
                _call("gpn_net_backward", ctx.progs[t], slots, rb_table, conv_t, bn_t, extra, dev)
        din = None
        for t in live:
            fresh, garena, _slots, _c, _b, pgrad, views = tabs[t]
            _log_backward(ctx.progs[t], rb_objs, need_in)
            _hand_over(ctx.progs[t], ctx.states[t][-1], views, pgrad, fresh)
            if need_in:
                d = garena[:features.numel()].view_as(features)
                din = d if din is None else din.add_(d)
        ctx.states = None
        return din, None, None, None, None, None



class _NetFnAutograd(torch.autograd.Function):
    """the same program with its parameters as autograd inputs (``set_autograd_parameters(True)``,
    or any parameter carrying a tensor hook): gradients are returned to autograd from a private buffer, so
    ``torch.autograd.grad``, ``backward(inputs=...)``, hooks and retained gradients behave as for any other op."""

    @staticmethod
    def forward(ctx, features, prog: NetProgram, rt, training, *params):
        return _forward_impl(ctx, features, prog, rt, training)

    @staticmethod
    def backward(ctx, dout):
        prog = ctx.prog
        din, _pgrad, views, params = _backward_impl(ctx, dout, True, bool(ctx.needs_input_grad[0]))
        prog.last_pgrad = None
        grads = tuple(g if ctx.needs_input_grad[4 + i] else None for i, g in enumerate(views))
        return (din, None, None, None) + grads


def program_for(unet) -> Optional[NetProgram]:
    """the cached program of a SparseUNet (rebuilt if its module tree changed); None if it cannot be expressed"""
    cached = unet.__dict__.get("_net_program")
    if cached is not None:
        if cached is False:
            return None
        return cached  # module replacement after the first forward needs invalidate(unet)
    try:
        prog = NetProgram(unet)
    except Unsupported:
        unet.__dict__["_net_program"] = False
        return None
    unet.__dict__["_net_program"] = prog
    return prog


def runs_natively(unet) -> bool:
    """will ``run(unet, x)`` take a float32 device tensor (rather than hand it back to the per-layer path)?  A program exists
    for the module tree and every BatchNorm agrees on its training flag - what GAPartNet checks before it hands the network
    a tensor whose row count is a device counter, which only the executor can read"""
    prog = program_for(unet)
    if prog is None or prog.python_stem_conv is not None:
        return False
    training = prog.bns[0].training
    return all(bn.training == training for bn in prog.bns)


def invalidate(unet):
    """drop the cached program (call after replacing sub-modules of a SparseUNet that has already run)"""
    unet.__dict__.pop("_net_program", None)


def _count_batches(buffers, x):
    """num_batches_tracked += 1 of a training-mode pass; with a device-counted row count only if any row exists (the reference
    does not run the network at all for a step without proposals: its counters stay)"""
    rows_dev = getattr(x, "rows_dev", None)
    if rows_dev is None:
        torch._foreach_add_(buffers, 1)
    else:
        # (list-list form: the tensor-scalar overload of _foreach_add_ reads its scalar on the host - a blocking sync)
        inc = (rows_dev.t[0] > 0).to(torch.int64)
        torch._foreach_add_(buffers, [inc] * len(buffers))


def _out_tensor(out, idx, shape, x, rows_dev):
    t = spconv.SparseConvTensor(out, idx, shape, x.batch_size, x.indice_dict)
    if rows_dev is not None:
        t.rows_dev = rows_dev
        t.batch_dev = getattr(x, "batch_dev", None)
    return t


def run(unet, x):
    """SparseUNet.forward through the native executor; returns None when the per-layer path must be used."""
    prog = program_for(unet)
    if prog is None or x.features.shape[0] == 0 or not x.features.is_cuda or x.features.dtype != torch.float32:
        return None
    training = prog.bns[0].training
    for bn in prog.bns:
        if bn.training != training:
            return None
    if prog.python_stem_conv is not None:
        known = getattr(x, "level_counts", None)
        x = prog.python_stem_conv(x)
        if known is not None:
            x.level_counts = known  # same voxel set: the coarse levels' row counts that came with the voxelisation's read
    lvl_dev = None
    if getattr(x, "rows_dev", None) is not None:
        rows, rb_table, rb_objs, levels, lvl_dev = prog.rulebooks_dev(x)
    else:
        rows, rb_table, rb_objs, levels = prog.rulebooks(x)
    if int(rows.min()) < 1 or x.features.shape[1] != prog.slot_channels[0]:
        return None
    if training:
        with torch.no_grad():
            _count_batches(prog.buffers("num_batches_tracked"), x)
    params = prog.params()
    if _AUTOGRAD_PARAMS or any(_has_hooks(p) for p in params):
        out = _NetFnAutograd.apply(x.features, prog, (rows, rb_table, rb_objs, lvl_dev), _mode(training), *params)
    else:
        out = _NetFn.apply(x.features, prog._anchor, prog, (rows, rb_table, rb_objs, lvl_dev), _mode(training))
    lvl = prog.slot_level[prog.out_slot]
    idx, shape = levels[lvl]
    return _out_tensor(out, idx, shape, x, lvl_dev[lvl] if lvl_dev is not None else None)


def run_pair(unet_a, unet_b, x):
    """``(unet_a(x), unet_b(x))`` for two SparseUNets of the same structure in paired passes (one launch per layer for both
    networks); None when the pair cannot run that way (different structures, a network the executor does not express, hooks
    on parameters, the autograd-parameter form) - the caller then runs the two networks one after the other."""
    prog_a, prog_b = program_for(unet_a), program_for(unet_b)
    if prog_a is None or prog_b is None or prog_a is prog_b or _AUTOGRAD_PARAMS:
        return None
    if x.features.shape[0] == 0 or not x.features.is_cuda or x.features.dtype != torch.float32:
        return None
    if (prog_a.ops != prog_b.ops or prog_a.slot_channels != prog_b.slot_channels or prog_a.rb_keys != prog_b.rb_keys
            or prog_a.level_keys != prog_b.level_keys or prog_a.python_stem_conv is not None
            or prog_b.python_stem_conv is not None or prog_a.conv_cin != prog_b.conv_cin or prog_a.conv_cout != prog_b.conv_cout):
        return None
    training = prog_a.bns[0].training
    for bn in prog_a.bns + prog_b.bns:
        if bn.training != training:
            return None
    params = prog_a.params() + prog_b.params()
    if any(_has_hooks(p) for p in params):
        return None
    lvl_dev = None
    if getattr(x, "rows_dev", None) is not None:
        rows, rb_table, rb_objs, levels, lvl_dev = prog_a.rulebooks_dev(x)
    else:
        rows, rb_table, rb_objs, levels = prog_a.rulebooks(x)
    if int(rows.min()) < 1 or x.features.shape[1] != prog_a.slot_channels[0]:
        return None
    if training:
        with torch.no_grad():
            _count_batches(prog_a.buffers("num_batches_tracked") + prog_b.buffers("num_batches_tracked"), x)
    out_a, out_b = _NetPairFn.apply(x.features, prog_a._anchor, prog_a, prog_b, (rows, rb_table, rb_objs, lvl_dev), _mode(training))
    lvl = prog_a.slot_level[prog_a.out_slot]
    idx, shape = levels[lvl]
    d = lvl_dev[lvl] if lvl_dev is not None else None
    return _out_tensor(out_a, idx, shape, x, d), _out_tensor(out_b, idx, shape, x, d)
