"""CPU-tensor front-end of the oracle with the same raw-op interface as ``gapartnet_amd.hip_ops``.

TEST INFRASTRUCTURE: lets ``tests/`` run the host-side glue (autograd wrappers, spconv/epic_ops mirrors,
model) on CPU against the restated algorithms, and lets ``bench.py`` time its ``cpu_baseline`` leg.  Never
imported by the product package.
"""
import numpy as np
import torch

import oracle as O
from gapartnet_amd.hip_ops import Rulebook, rows_csr  # noqa: F401  (dataclass + pure-torch helper)

name = "oracle"
TILE_ROWS = O.TILE_ROWS


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


def voxelize(points, feats, seg_offsets, seg_range_min, seg_range_max, voxel_size, grid_dims, want_csr=False,
             want_stats=False):
    vf, vc, vs, pid = O.voxelize(_np(points), _np(feats), _np(seg_offsets), _np(seg_range_min), _np(seg_range_max),
                                 voxel_size, grid_dims)
    out = (_t(vf), _t(vc), _t(vs), _t(pid))
    if want_csr:
        order, starts = rows_csr(out[3], vf.shape[0])
        out = out + (order, starts)
    if want_stats:
        out = out + ({"max_coord": [int(v) for v in vc.max(0)] if vc.shape[0] else [0, 0, 0],
                      "dropped": int((pid < 0).sum())},)
    return out


def _rb(lists, K, n_src, n_dst):
    src, dst, toff = lists
    return Rulebook(_t(src), _t(dst), _t(toff), K, n_src, n_dst, torch.tensor(src.shape[0], dtype=torch.int64))


def rulebook_subm3(indices, spatial_shape):
    N = indices.shape[0]
    return _rb(O.rulebook_subm3(_np(indices), spatial_shape), 27, N, N)


def rulebook_down(indices, spatial_shape, batch_size):
    N = indices.shape[0]
    d = O.rulebook_down(_np(indices), spatial_shape)
    No = d["out_indices"].shape[0]
    return _t(d["out_indices"]), d["out_shape"], _rb(d["fwd"], 8, N, No), _rb(d["bwd"], 8, No, N)


def _lists(rb):
    return _np(rb.pair_src), _np(rb.pair_dst), _np(rb.tile_off)


def _canon(W, layout):
    """canonical [K, Cin, Cout] view of a weight given in parameter layout [Cout, K, Cin] ("oki")"""
    return W.permute(1, 2, 0).contiguous() if layout == "oki" else W


def conv_fwd(features, W, rb, layout="kio"):
    return _t(O.spconv_fwd(_np(features), _np(_canon(W, layout)), _lists(rb), rb.n_dst))


def conv_dgrad(dout, W, rb, rb_t, reverse_taps, layout="kio"):
    return _t(O.spconv_dgrad(_np(dout), _np(_canon(W, layout)), _lists(rb), rb.n_dst, rb.n_src))


def conv_wgrad(features, dout, rb, layout="kio"):
    dW = _t(O.spconv_wgrad(_np(features), _np(dout), _lists(rb), rb.n_dst, rb.K))
    return dW.permute(2, 0, 1).contiguous() if layout == "oki" else dW


def bn_fwd(x, res, weight, bias, running_mean, running_var, training, momentum, eps, relu):
    """plain-torch restatement of BatchNorm1d (+ residual) (+ ReLU), network/backbone.py:40-49 / model.py:86"""
    if training:
        mean = x.double().mean(0)
        var = x.double().var(0, unbiased=False)
        invstd = (1.0 / torch.sqrt(var + eps)).float()
        if running_mean is not None:
            n = x.shape[0]
            unbiased = var * (n / (n - 1)) if n > 1 else var
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * unbiased.float())
        mean = mean.float()
    else:
        mean, invstd = running_mean, torch.rsqrt(running_var + eps)
    y = (x - mean) * invstd * weight + bias
    if res is not None:
        y = y + res
    if relu:
        y = torch.relu(y)
    return y, mean, invstd


def bn_bwd(x, y, dy, weight, mean, invstd, relu, training, has_res):
    g = torch.where(y > 0, dy, torch.zeros_like(dy)) if relu else dy
    xhat = (x - mean) * invstd
    db = g.double().sum(0).float()
    dw = (g.double() * xhat.double()).sum(0).float()
    if training:
        n = x.shape[0]
        dx = (g - db / n - xhat * (dw / n)) * invstd * weight
    else:
        dx = g * invstd * weight
    return dx, (g if has_res else None), dw, db


def gather_rows(table, idx):
    return _t(O.gather_rows(_np(table), _np(idx)))


def scatter_rows(dout, idx, n_rows, csr=None):
    return _t(O.scatter_rows(_np(dout), _np(idx), n_rows))


def ball_query(points, query, batch_indices, batch_offsets, radius, num_samples, point_labels=None,
               query_labels=None):
    idx, cnt = O.ball_query(_np(points), _np(query), _np(batch_indices), _np(batch_offsets), radius,
                            int(num_samples), _np(point_labels), _np(query_labels))
    return _t(idx), _t(cnt)


def ccl(begin_end, edges, compacted=False):
    return _t(O.ccl(_np(begin_end), _np(edges), compacted))


def segmented_reduce(values, begin, end, mode):
    return _t(O.segmented_reduce(_np(values), _np(begin), _np(end), mode))


def segmented_maxpool_fwd(values, begin, end):
    p, a = O.segmented_maxpool(_np(values), _np(begin), _np(end))
    return _t(p), _t(a)


def segmented_maxpool_bwd(dpooled, argmax, M):
    return _t(O.segmented_maxpool_bwd(_np(dpooled), _np(argmax), M))


def instance_iou(proposal_offsets, instance_labels, batch_indices, num_points_per_instance):
    return _t(O.instance_iou(_np(proposal_offsets), _np(instance_labels), _np(batch_indices),
                             _np(num_points_per_instance)))


def nms(ious, scores, threshold):
    return _t(O.nms(_np(ious), _np(scores), threshold))


def pn2_ball_query(radius, nsample, xyz, new_xyz):
    return _t(O.pn2_ball_query(radius, nsample, _np(xyz), _np(new_xyz)))


def pn2_group_points(points, idx):
    return _t(O.pn2_group_points(_np(points), _np(idx)))


def pn2_group_points_grad(grad_out, idx, n):
    return _t(O.pn2_group_points_grad(_np(grad_out), _np(idx), n))


def pn2_gather_points(points, idx):
    return _t(O.pn2_gather_points(_np(points), _np(idx)))


def pn2_gather_points_grad(grad_out, idx, n):
    return _t(O.pn2_gather_points_grad(_np(grad_out), _np(idx), n))


def pn2_furthest_point_sampling(xyz, npoint):
    return _t(O.pn2_furthest_point_sampling(_np(xyz), npoint))


def pn2_three_nn(unknown, known):
    d2, idx = O.pn2_three_nn(_np(unknown), _np(known))
    return _t(d2), _t(idx)


def pn2_knn(unknown, known, k):
    d2, idx = O.pn2_knn(_np(unknown), _np(known), k)
    return _t(d2), _t(idx)


def pn2_three_interpolate(points, idx, weight):
    return _t(O.pn2_three_interpolate(_np(points), _np(idx), _np(weight)))


def pn2_three_interpolate_grad(grad_out, idx, weight, m):
    return _t(O.pn2_three_interpolate_grad(_np(grad_out), _np(idx), _np(weight), m))
