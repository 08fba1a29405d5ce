"""Autograd wrappers over the raw operators (backend-neutral: they call ``backend.raw()``).

  sparse_conv        — kernel C (forward / dgrad / wgrad) as one differentiable op
  gather_rows        — kernel G: voxel rows -> points, deterministic CSR transpose in backward
  segmented_maxpool  — kernel R (differentiable max-pool over proposal segments)
"""
from typing import Optional

import torch
import torch.nn.functional as F

from . import backend

# optional log of conv launches for bench.py's roofline accounting: entries (num_pairs tensor, cin, cout, n_src, n_dst, K, kind,
# live row counters (src, dst) of a device-counted rulebook or None - n_src / n_dst are then the buffers' bounds)
CONV_LOG = None


def _log(rb, cin, cout, kind):
    if CONV_LOG is not None:
        live = getattr(rb, "live_src", None), getattr(rb, "live_dst", None)
        CONV_LOG.append((rb.num_pairs, cin, cout, rb.n_src, rb.n_dst, rb.K, kind, live if live[0] is not None else None))


def _pad16(n):
    return (n + 15) // 16 * 16


class _SparseConvFn(torch.autograd.Function):
    """out[dst] = sum_k features[src] @ W[k] over rulebook ``rb``; ``rb_t`` is the transposed rulebook used by
    dgrad (for SubM convs rb_t is rb and the taps are reversed)."""

    @staticmethod
    def forward(ctx, features, weight, rb, rb_t, reverse_taps, layout):
        ops = backend.raw()
        features = features.contiguous()
        weight = weight.contiguous()
        cin, cout = (weight.shape[2], weight.shape[0]) if layout == "oki" else (weight.shape[1], weight.shape[2])
        out = ops.conv_fwd(features, weight, rb, layout)
        _log(rb, cin, cout, "fwd")
        ctx.save_for_backward(features, weight)
        ctx.rb, ctx.rb_t, ctx.reverse_taps, ctx.layout, ctx.dims = rb, rb_t, reverse_taps, layout, (cin, cout)
        return out

    @staticmethod
    def backward(ctx, dout):
        ops = backend.raw()
        features, weight = ctx.saved_tensors
        cin, cout = ctx.dims
        dout = dout.contiguous()
        din = dW = None
        if ctx.needs_input_grad[0]:
            din = ops.conv_dgrad(dout, weight, ctx.rb, ctx.rb_t, ctx.reverse_taps, ctx.layout)
            _log(ctx.rb_t, cout, cin, "dgrad")
        if ctx.needs_input_grad[1]:
            dW = ops.conv_wgrad(features, dout, ctx.rb, ctx.layout)
            _log(ctx.rb, cin, cout, "wgrad")
        return din, dW, None, None, None, None


def sparse_conv_param(features: torch.Tensor, weight_oki: torch.Tensor, rb, rb_t, reverse_taps: bool) -> torch.Tensor:
    """same op on the conv parameter in its own layout [Cout, K, Cin] (a free view of the spconv-2.x parameter):
    packing reads it and wgrad writes its gradient directly in that layout, so no permute/copy kernels run.
    Channel counts must be multiples of 16 (otherwise use sparse_conv on the canonical view)."""
    return _SparseConvFn.apply(features, weight_oki, rb, rb_t, reverse_taps, "oki")


def sparse_conv(features: torch.Tensor, weight: torch.Tensor, rb, rb_t, reverse_taps: bool) -> torch.Tensor:
    """features [n_src, cin], weight canonical [K, cin, cout] -> [n_dst, cout].

    The kernels need channel counts that are multiples of 16; other sizes (the 6-channel stem) are zero-padded
    here, inside autograd, so gradients are sliced back automatically."""
    K, cin, cout = weight.shape
    cin_p, cout_p = _pad16(cin), _pad16(cout)
    if cin_p != cin:
        features = F.pad(features, (0, cin_p - cin))
    if cin_p != cin or cout_p != cout:
        weight = F.pad(weight, (0, cout_p - cout, 0, cin_p - cin))
    out = _SparseConvFn.apply(features, weight, rb, rb_t, reverse_taps, "kio")
    if cout_p != cout:
        out = out[:, :cout]
    return out


_IDENTITY_RULEBOOKS = {}


def _identity_rulebook(n: int, device):
    """K = 1 rulebook mapping every row to itself (cached per (rows, device); a handful of sizes per step)"""
    key = (int(n), str(device))
    rb = _IDENTITY_RULEBOOKS.get(key)
    if rb is None:
        from .spconv.pytorch import _identity_rulebook as build
        if len(_IDENTITY_RULEBOOKS) > 64:
            _IDENTITY_RULEBOOKS.clear()
        rb = _IDENTITY_RULEBOOKS[key] = build(int(n), device)
    return rb


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b of the dense heads.  On the HIP library: its own streaming kernels (csrc/linear.hip: one launch
    forward, three backward).  Over other raw-op backends (the oracle in the CPU tests), and for widths those kernels do not
    take: the K = 1 case of the conv operator family, output channels zero-padded to a multiple of 16 - one autograd node
    with three operator calls (forward, dgrad, wgrad) and no autograd-visible pad / slice ops."""

    @staticmethod
    def forward(ctx, x, weight, bias, rows=None):
        ops = backend.raw()
        x = x.contiguous()
        cout, cin = weight.shape
        ctx.native = hasattr(ops, "linear_fwd") and ops.linear_supported(cin, cout)
        ctx.rows = rows
        if rows is not None:
            assert ctx.native, "a device-counted row count needs the dense-head kernels (csrc/linear.hip)"
            w = weight.detach().contiguous()
            ctx.save_for_backward(x, w)
            ctx.has_bias = bias is not None
            return ops.linear_fwd(x, w, bias.detach() if bias is not None else None, rows=rows)
        if ctx.native:
            w = weight.detach().contiguous()
            ctx.save_for_backward(x, w)
            ctx.has_bias = bias is not None
            return ops.linear_fwd(x, w, bias.detach() if bias is not None else None)
        cout_p = _pad16(cout)
        w = weight.detach()
        if cout_p != cout:
            w = F.pad(w, (0, 0, 0, cout_p - cout))
        w = w.contiguous().view(cout_p, 1, cin)
        rb = _identity_rulebook(x.shape[0], x.device)
        out = ops.conv_fwd(x, w, rb, "oki")
        ctx.save_for_backward(x, w)
        ctx.rb, ctx.cout, ctx.has_bias = rb, cout, bias is not None
        y = out[:, :cout] if cout_p != cout else out
        return y + bias if bias is not None else y

    @staticmethod
    def backward(ctx, dy):
        ops = backend.raw()
        x, w = ctx.saved_tensors
        if ctx.rows is not None:
            return ops.linear_bwd(x, w, dy.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                  ctx.has_bias and ctx.needs_input_grad[2], rows=ctx.rows) + (None,)
        if ctx.native:
            return ops.linear_bwd(x, w, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                  ctx.has_bias and ctx.needs_input_grad[2]) + (None,)
        cout, cout_p = ctx.cout, w.shape[0]
        dy_p = dy.contiguous() if cout_p == cout else F.pad(dy, (0, cout_p - cout))
        dx = ops.conv_dgrad(dy_p, w, ctx.rb, ctx.rb, False, "oki") if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(x, dy_p, ctx.rb, "oki").view(cout_p, x.shape[1])[:cout]
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, rows=None) -> torch.Tensor:
    """``F.linear`` for the per-point / per-voxel heads (model.py:160-175: Linear(16, n) on 10^5 rows).  These are
    K = 1 cases of the fused conv kernel family (forward, dgrad, wgrad): at [160k, 16] x [16, 3..27] the library GEMMs
    the framework dispatches to run at 0.2-0.4 TFLOP/s (fwd+bwd 260-380 us per layer vs 200-250 here).
    Falls back to F.linear for shapes the kernels do not cover."""
    ops = backend.raw()
    if rows is not None:  # hip_ops.DevCount: x.shape[0] is a bound, the live row count a device counter
        return _LinearFn.apply(x, weight, bias, rows)
    if ops.name != "hip" or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32 or x.shape[0] == 0:
        return F.linear(x, weight, bias)
    if hasattr(ops, "linear_supported") and ops.linear_supported(x.shape[1], weight.shape[0]):
        # (any row count: the first torch GEMM of a process initialises the BLAS library - a 250 ms stall in the step where
        # a batch first has fewer than 16 proposals, when the score head used to fall back to F.linear)
        return _LinearFn.apply(x, weight, bias)
    if x.shape[1] % 16 != 0 or x.shape[0] < 16:
        return F.linear(x, weight, bias)
    return _LinearFn.apply(x, weight, bias)


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx, csr, rows, table_rows):
        ops = backend.raw()
        ctx.save_for_backward(idx)
        ctx.csr = csr
        ctx.n_rows = table.shape[0]
        ctx.table_rows = table_rows
        if rows is not None:
            return ops.gather_rows(table.contiguous(), idx, rows=rows)
        return ops.gather_rows(table.contiguous(), idx)

    @staticmethod
    def backward(ctx, dout):
        ops = backend.raw()
        (idx,) = ctx.saved_tensors
        if ctx.table_rows is not None:
            return ops.scatter_rows(dout.contiguous(), idx, ctx.n_rows, ctx.csr, rows=ctx.table_rows), None, None, None, None
        return ops.scatter_rows(dout.contiguous(), idx, ctx.n_rows, ctx.csr), None, None, None, None


def gather_rows(table: torch.Tensor, idx: torch.Tensor, csr=None, rows=None, table_rows=None) -> torch.Tensor:
    """table[idx] with idx<0 -> zero row (reference: ``voxel_features.features[pc_voxel_id]``, model.py:153,359,394).
    ``csr`` = (order, starts) of points grouped by row, if the caller already has it (voxelize returns it).
    ``rows`` / ``table_rows`` (hip_ops.DevCount): the live lengths of ``idx`` / of ``table`` are device counters."""
    return _GatherRowsFn.apply(table, idx, csr, rows, table_rows)


class _VoxelMeanFn(torch.autograd.Function):
    """kernel V as a differentiable op: voxel_features = mean of the member points' features, so
    d feats[i] = d voxel_features[pc_voxel_id[i]] / (points in that voxel)   (0 for points outside the grid).
    The operator the reference calls (epic_ops.voxelize, absent from the reference tree) belongs to the PointGroup
    family of voxelisation ops, whose mean mode back-propagates exactly this; without it the ScoreNet / NPCS-Net losses
    would never reach the backbone through the proposal features (network/model.py:300-346 feeds pt_features with
    their graph attached)."""

    @staticmethod
    def forward(ctx, feats, points, seg_offsets, rmin, rmax, voxel_size, grid_dims, want_stats):
        ops = backend.raw()
        out = ops.voxelize(points, feats.contiguous(), seg_offsets, rmin, rmax, voxel_size, grid_dims, want_csr=True,
                           want_stats=want_stats)
        vf, vc, vseg, pid, order, starts = out[:6]
        ctx.save_for_backward(pid, starts)
        ctx.mark_non_differentiable(vc, vseg, pid, order, starts)
        return vf, vc, vseg, pid, order, starts, (out[6] if want_stats else None)

    @staticmethod
    def backward(ctx, d_vf, *_unused):
        ops = backend.raw()
        pid, starts = ctx.saved_tensors
        counts = (starts[1:] - starts[:-1]).clamp(min=1).to(d_vf.dtype)
        d_feats = ops.gather_rows((d_vf / counts[:, None]).contiguous(), pid)
        return d_feats, None, None, None, None, None, None, None


def voxelize_mean(points, feats, seg_offsets, rmin, rmax, voxel_size, grid_dims, want_stats=False):
    """differentiable (w.r.t. ``feats``) form of the raw voxelize op -> (voxel_feats, coords, seg, pc_voxel_id, order,
    starts, stats or None); order / starts = points grouped by voxel (the CSR the voxel->point gathers use in backward),
    stats = {"max_coord", "dropped"} from the op's single host read when ``want_stats``."""
    return _VoxelMeanFn.apply(feats, points, seg_offsets, rmin, rmax, voxel_size, grid_dims, want_stats)


class _ProposalVoxelMeanFn(torch.autograd.Function):
    """voxel features of the proposal grids = ordered mean of the member points' backbone features (kernel V's mean on
    ``pt_features[point_indices]``), differentiable w.r.t. the per-point features: each point collects d voxel / count from
    the (at most two) proposals it belongs to.  The index structure comes from ``hip_ops.proposals_build``."""

    @staticmethod
    def forward(ctx, feats, point_indices, point_order, voxel_point_start, member_slot, pc_voxel_id, n_voxels, rows):
        ops = backend.raw()
        ctx.save_for_backward(member_slot, pc_voxel_id, voxel_point_start)
        ctx.n_points = feats.shape[0]
        if rows is not None:
            return ops.proposals_voxel_mean(feats.contiguous(), point_indices, point_order, voxel_point_start, n_voxels, rows=rows)
        return ops.proposals_voxel_mean(feats.contiguous(), point_indices, point_order, voxel_point_start, n_voxels)

    @staticmethod
    def backward(ctx, dout):
        ops = backend.raw()
        member_slot, pc_voxel_id, voxel_point_start = ctx.saved_tensors
        # (one thread per ORIGINAL point walks its at most two memberships: no data-dependent extent)
        return (ops.proposals_voxel_mean_bwd(dout.contiguous(), member_slot, pc_voxel_id, voxel_point_start, ctx.n_points),
                None, None, None, None, None, None, None)


def proposal_voxel_mean(feats, built, rows=None) -> torch.Tensor:
    return _ProposalVoxelMeanFn.apply(feats, built["point_indices"], built["point_order"], built["voxel_point_start"],
                                      built["member_slot"], built["pc_voxel_id"], built["V"], rows)


class _ScoreLossFn(torch.autograd.Function):
    """proposal score loss (model.py:348-396 of the reference with the class selection of :560-566) as one launch; backward is
    one multiplication.  -> (loss 0-dim, score_preds [P] without a gradient)"""

    @staticmethod
    def forward(ctx, logits, cls_source, proposal_offsets, ious, fg_thresh, bg_thresh, rows):
        if rows is not None:
            loss, preds, d_logits = backend.raw().score_loss(logits.contiguous(), cls_source, proposal_offsets, ious, fg_thresh,
                                                             bg_thresh, rows=rows)
        else:
            loss, preds, d_logits = backend.raw().score_loss(logits.contiguous(), cls_source, proposal_offsets, ious, fg_thresh,
                                                             bg_thresh)
        ctx.save_for_backward(d_logits)
        ctx.mark_non_differentiable(preds)
        return loss.reshape(()), preds

    @staticmethod
    def backward(ctx, grad_loss, _grad_preds):
        (d_logits,) = ctx.saved_tensors
        return d_logits * grad_loss, None, None, None, None, None, None


def score_loss_available(logits: torch.Tensor) -> bool:
    return (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[0] > 0
            and hasattr(backend.raw(), "score_loss"))


def score_loss(logits, cls_source, proposal_offsets, ious, fg_thresh: float = 0.75, bg_thresh: float = 0.25, rows=None):
    return _ScoreLossFn.apply(logits, cls_source, proposal_offsets, ious, fg_thresh, bg_thresh, rows)


class _NpcsLossFn(torch.autograd.Function):
    """symmetry-aware NPCS loss of all proposals (model.py:398-462): two launches forward, one backward"""

    @staticmethod
    def forward(ctx, logits, gt_npcs, sem_preds, sem_labels, proposal_offsets, proposal_indices, sym, p_rows, m_rows):
        ops = backend.raw()
        logits, gt_npcs = logits.contiguous(), gt_npcs.contiguous()
        sem_preds, sem_labels = sem_preds.to(torch.int32).contiguous(), sem_labels.to(torch.int64).contiguous()
        if p_rows is not None:
            loss, scratch = ops.npcs_loss_fwd(logits, gt_npcs, sem_preds, sem_labels, proposal_offsets, sym, rows=p_rows)
        else:
            loss, scratch = ops.npcs_loss_fwd(logits, gt_npcs, sem_preds, sem_labels, proposal_offsets, sym)
        ctx.save_for_backward(logits, gt_npcs, sem_preds, sem_labels, proposal_indices, scratch)
        ctx.sym, ctx.P, ctx.m_rows = sym, proposal_offsets.shape[0] - 1, m_rows
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        ops = backend.raw()
        logits, gt_npcs, sem_preds, sem_labels, proposal_indices, scratch = ctx.saved_tensors
        if ctx.m_rows is not None:
            d_logits = ops.npcs_loss_bwd(logits, gt_npcs, sem_preds, sem_labels, proposal_indices, ctx.P, ctx.sym, scratch, grad_loss,
                                         m_rows=ctx.m_rows)
        else:
            d_logits = ops.npcs_loss_bwd(logits, gt_npcs, sem_preds, sem_labels, proposal_indices, ctx.P, ctx.sym, scratch, grad_loss)
        return d_logits, None, None, None, None, None, None, None, None


def npcs_loss(logits, gt_npcs, sem_preds, sem_labels, proposal_offsets, proposal_indices, sym, p_rows=None, m_rows=None) -> torch.Tensor:
    return _NpcsLossFn.apply(logits, gt_npcs, sem_preds, sem_labels, proposal_offsets, proposal_indices, sym, p_rows, m_rows)


class _PointLossesFn(torch.autograd.Function):
    """kernel family P: focal + dice + offset-distance + offset-direction losses of all points in one pass"""

    @staticmethod
    def forward(ctx, logits, offsets, labels, gt_offsets, instance_labels, ignore_index, metrics=False):
        ops = backend.raw()
        logits, offsets = logits.contiguous(), offsets.contiguous()
        labels, gt_offsets = labels.contiguous(), gt_offsets.contiguous()
        instance_labels = instance_labels.to(torch.int32).contiguous()
        ctx.ignore_index = ignore_index
        if metrics:
            losses, stats, preds, accu = ops.point_losses_fwd(logits, labels, offsets, gt_offsets, instance_labels, ignore_index,
                                                              metrics=True)
            ctx.save_for_backward(logits, offsets, labels, gt_offsets, instance_labels, stats)
            ctx.mark_non_differentiable(preds, accu)
            return losses, preds, accu
        losses, stats = ops.point_losses_fwd(logits, labels, offsets, gt_offsets, instance_labels, ignore_index)
        ctx.save_for_backward(logits, offsets, labels, gt_offsets, instance_labels, stats)
        return losses

    @staticmethod
    def backward(ctx, grad_losses, *_unused):
        ops = backend.raw()
        logits, offsets, labels, gt_offsets, instance_labels, stats = ctx.saved_tensors
        d_logits, d_offsets = ops.point_losses_bwd(logits, labels, offsets, gt_offsets, instance_labels, ctx.ignore_index,
                                                   stats, grad_losses.contiguous())
        return d_logits, d_offsets, None, None, None, None, None


def point_losses_available(logits: torch.Tensor) -> bool:
    return backend.raw().name == "hip" and logits.is_cuda and logits.dim() == 2 and logits.shape[1] <= 32 \
        and logits.shape[0] > 0 and logits.dtype == torch.float32


def point_losses(logits, offsets, labels, gt_offsets, instance_labels, ignore_index: int = -100) -> torch.Tensor:
    """-> [4] = (focal loss, dice loss, offset L1 loss, offset direction loss) exactly as network/model.py:177-226 computes
    them from ~70 torch ops (focal_loss + dice_loss + loss_offset); one launch forward, one backward"""
    return _PointLossesFn.apply(logits, offsets, labels, gt_offsets, instance_labels, int(ignore_index))


def point_losses_with_metrics(logits, offsets, labels, gt_offsets, instance_labels, ignore_index: int = -100):
    """``point_losses`` plus the by-products of the logits the step needs anyway, from the same pass: -> (losses [4],
    sem_preds [M] i64 = torch.argmax(logits, -1), accu [2] = ((sem_preds == labels).sum().float() / M,
    ((sem_preds == labels) & (labels > 0)).sum() / (labels > 0).sum()) - network/model.py:531-541)"""
    return _PointLossesFn.apply(logits, offsets, labels, gt_offsets, instance_labels, int(ignore_index), True)


class _SegmentedMaxpoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, begin, end, rows, m_rows):
        ops = backend.raw()
        if rows is not None:
            pooled, argmax = ops.segmented_maxpool_fwd(values.contiguous(), begin, end, rows=rows)
        else:
            pooled, argmax = ops.segmented_maxpool_fwd(values.contiguous(), begin, end)
        ctx.save_for_backward(argmax)
        ctx.M, ctx.rows, ctx.m_rows = values.shape[0], rows, m_rows
        ctx.mark_non_differentiable(argmax)
        return pooled, argmax

    @staticmethod
    def backward(ctx, dpooled, _dargmax):
        ops = backend.raw()
        (argmax,) = ctx.saved_tensors
        if ctx.rows is not None:
            return ops.segmented_maxpool_bwd(dpooled.contiguous(), argmax, ctx.M, rows=ctx.rows, m_rows=ctx.m_rows), None, None, None, None
        return ops.segmented_maxpool_bwd(dpooled.contiguous(), argmax, ctx.M), None, None, None, None


def segmented_maxpool(values, begin, end, rows=None, m_rows=None):
    """``rows`` / ``m_rows`` (hip_ops.DevCount): the number of segments / of value rows are device counters"""
    return _SegmentedMaxpoolFn.apply(values, begin, end, rows, m_rows)


class _BnActFn(torch.autograd.Function):
    """y = act(batch_norm(x) [+ residual]) as one op (kernel family BN)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu):
        ops = backend.raw()
        y, mean, invstd = ops.bn_fwd(x.contiguous(), None if residual is None else residual.contiguous(), weight, bias,
                                     running_mean, running_var, training, momentum, eps, relu)
        ctx.save_for_backward(x, y, weight, mean, invstd)
        ctx.relu, ctx.training, ctx.has_res = relu, training, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = backend.raw()
        x, y, weight, mean, invstd = ctx.saved_tensors
        dx, dres, dw, db = ops.bn_bwd(x, y, dy.contiguous(), weight, mean, invstd, ctx.relu, ctx.training, ctx.has_res)
        return dx, dres, dw, db, None, None, None, None, None, None


def bn_act(x: torch.Tensor, bn: torch.nn.BatchNorm1d, relu: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``relu(bn(x) + residual)`` (any of the two optional) with the parameters / running statistics of the given
    ``nn.BatchNorm1d`` module, in its current train/eval mode.  Falls back to the module itself for shapes the fused
    kernels do not cover (C % 4 != 0, empty input, no affine / running stats)."""
    C = x.shape[1]
    if x.shape[0] == 0 or C % 4 != 0 or not bn.affine or not bn.track_running_stats or bn.momentum is None:
        y = bn(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    training = bn.training
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BnActFn.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, float(bn.momentum),
                          float(bn.eps), bool(relu))
