"""CPU oracle for the GAPartNet hot path — TEST INFRASTRUCTURE ONLY.

numpy front-end over ``oracle/gpn_oracle.c`` (see that file's header for the parity status and the
reference citations).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product package ``gapartnet_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgpn_oracle.so")
TILE_ROWS = 32


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gpn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for name in ("orc_voxelize", "orc_rulebook_subm3", "orc_rulebook_down", "orc_rulebook_down_lists"):
            getattr(_lib, name).restype = ctypes.c_int64
        _lib.orc_nms.restype = ctypes.c_int32
    return _lib


def set_threads(n: int) -> int:
    """OpenMP threads of the conv / ball-query loops (results do not depend on it); -> the count now in effect"""
    return int(lib().orc_set_threads(ctypes.c_int(int(n))))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def n_tiles(n):
    return (int(n) + TILE_ROWS - 1) // TILE_ROWS


# ---------------------------------------------------------------------------------------------- V
def voxelize(points, feats, seg_offsets, seg_range_min, seg_range_max, voxel_size, grid_dims):
    points, feats = _f32(points), _f32(feats)
    seg_offsets = _i64(seg_offsets)
    M, C, S = points.shape[0], feats.shape[1], seg_offsets.shape[0] - 1
    rmin = _f32(np.broadcast_to(_f32(seg_range_min).reshape(-1, 3), (S, 3)))
    rmax = _f32(np.broadcast_to(_f32(seg_range_max).reshape(-1, 3), (S, 3)))
    vs, gd = _f32(voxel_size), _i32(grid_dims)
    vf = np.zeros((M, C), np.float32)
    vc = np.zeros((M, 3), np.int32)
    vseg = np.zeros((M,), np.int32)
    pid = np.zeros((M,), np.int32)
    V = lib().orc_voxelize(_p(points), _p(feats), _p(seg_offsets), _p(rmin), _p(rmax),
                           ctypes.c_int64(M), ctypes.c_int(C), ctypes.c_int64(S), _p(vs), _p(gd),
                           _p(vf), _p(vc), _p(vseg), _p(pid))
    return vf[:V].copy(), vc[:V].copy(), vseg[:V].copy(), pid


# ---------------------------------------------------------------------------------------------- K
def rulebook_subm3(indices, spatial_shape):
    indices = _i32(indices)
    N = indices.shape[0]
    shape = _i32(spatial_shape)
    src = np.zeros((27 * max(N, 1),), np.int32)
    dst = np.zeros((27 * max(N, 1),), np.int32)
    toff = np.zeros((27, n_tiles(N) + 1), np.int32)
    P = lib().orc_rulebook_subm3(_p(indices), ctypes.c_int64(N), _p(shape), _p(src), _p(dst), _p(toff))
    return src[:P].copy(), dst[:P].copy(), toff


def rulebook_down(indices, spatial_shape):
    indices = _i32(indices)
    N = indices.shape[0]
    shape = _i32(spatial_shape)
    out_idx = np.zeros((max(N, 1), 4), np.int32)
    f2c = np.zeros((max(N, 1),), np.int32)
    tap = np.zeros((max(N, 1),), np.int32)
    No = lib().orc_rulebook_down(_p(indices), ctypes.c_int64(N), _p(shape), _p(out_idx), _p(f2c), _p(tap))
    f2c, tap = f2c[:N], tap[:N]
    fs, fd = np.zeros((max(N, 1),), np.int32), np.zeros((max(N, 1),), np.int32)
    bs, bd = np.zeros((max(N, 1),), np.int32), np.zeros((max(N, 1),), np.int32)
    ft = np.zeros((8, n_tiles(No) + 1), np.int32)
    bt = np.zeros((8, n_tiles(N) + 1), np.int32)
    P = lib().orc_rulebook_down_lists(_p(_i32(f2c)), _p(_i32(tap)), ctypes.c_int64(N), ctypes.c_int64(No),
                                      _p(fs), _p(fd), _p(ft), _p(bs), _p(bd), _p(bt))
    out_shape = [int(s) // 2 for s in spatial_shape]
    return dict(out_indices=out_idx[:No].copy(), out_shape=out_shape, fine_to_coarse=f2c.copy(),
                tap=tap.copy(), fwd=(fs[:P].copy(), fd[:P].copy(), ft), bwd=(bs[:P].copy(), bd[:P].copy(), bt))


# ---------------------------------------------------------------------------------------------- C
def spconv_fwd(inp, W, rb, n_dst):
    src, dst, toff = rb
    inp, W = _f32(inp), _f32(W)
    K, cin, cout = W.shape
    out = np.zeros((n_dst, cout), np.float32)
    lib().orc_spconv_fwd(_p(inp), _p(W), _p(_i32(src)), _p(_i32(dst)), _p(_i32(toff)), ctypes.c_int(K),
                         ctypes.c_int64(n_dst), ctypes.c_int(cin), ctypes.c_int(cout), _p(out))
    return out


def spconv_dgrad(dout, W, rb, n_dst, n_src):
    src, dst, toff = rb
    dout, W = _f32(dout), _f32(W)
    K, cin, cout = W.shape
    din = np.zeros((n_src, cin), np.float32)
    lib().orc_spconv_dgrad(_p(dout), _p(W), _p(_i32(src)), _p(_i32(dst)), _p(_i32(toff)), ctypes.c_int(K),
                           ctypes.c_int64(n_dst), ctypes.c_int64(n_src), ctypes.c_int(cin),
                           ctypes.c_int(cout), _p(din))
    return din


def spconv_wgrad(inp, dout, rb, n_dst, K):
    src, dst, toff = rb
    inp, dout = _f32(inp), _f32(dout)
    cin, cout = inp.shape[1], dout.shape[1]
    dW = np.zeros((K, cin, cout), np.float32)
    lib().orc_spconv_wgrad(_p(inp), _p(dout), _p(_i32(src)), _p(_i32(dst)), _p(_i32(toff)), ctypes.c_int(K),
                           ctypes.c_int64(n_dst), ctypes.c_int(cin), ctypes.c_int(cout), _p(dW))
    return dW


def gather_rows(table, idx):
    table, idx = _f32(table), _i32(idx)
    out = np.zeros((idx.shape[0], table.shape[1]), np.float32)
    lib().orc_gather_rows(_p(table), _p(idx), ctypes.c_int64(idx.shape[0]), ctypes.c_int(table.shape[1]), _p(out))
    return out


def scatter_rows(dout, idx, n_rows):
    dout, idx = _f32(dout), _i32(idx)
    dt = np.zeros((n_rows, dout.shape[1]), np.float32)
    lib().orc_scatter_rows(_p(dout), _p(idx), ctypes.c_int64(idx.shape[0]), ctypes.c_int64(n_rows),
                           ctypes.c_int(dout.shape[1]), _p(dt))
    return dt


# ---------------------------------------------------------------------------------------------- B/L
def ball_query(points, query, batch_indices, batch_offsets, radius, K, point_labels=None, query_labels=None):
    points, query = _f32(points), _f32(query)
    bi, bo = _i32(batch_indices), _i32(batch_offsets)
    pl = None if point_labels is None else _i32(point_labels)
    ql = None if query_labels is None else _i32(query_labels)
    Q = query.shape[0]
    idx = np.zeros((Q, K), np.int32)
    cnt = np.zeros((Q,), np.int32)
    lib().orc_ball_query(_p(points), _p(query), _p(bi), _p(bo), _p(pl), _p(ql), ctypes.c_int64(points.shape[0]),
                         ctypes.c_int64(Q), ctypes.c_int64(bo.shape[0] - 1), ctypes.c_float(radius),
                         ctypes.c_int(K), _p(idx), _p(cnt))
    return idx, cnt


def ccl(begin_end, edges, compacted=False):
    be, edges = _i32(begin_end), _i32(edges)
    Q = be.shape[0] // 2
    labels = np.zeros((Q,), np.int32)
    lib().orc_ccl(_p(be), _p(edges), ctypes.c_int64(Q), ctypes.c_int64(edges.shape[0]),
                  ctypes.c_int(1 if compacted else 0), _p(labels))
    return labels


# ---------------------------------------------------------------------------------------------- R/I/N
def segmented_reduce(values, begin, end, mode):
    values, begin, end = _f32(values), _i32(begin), _i32(end)
    P, C = begin.shape[0], values.shape[1]
    out = np.zeros((P, C), np.float32)
    lib().orc_segmented_reduce(_p(values), _p(begin), _p(end), ctypes.c_int64(P), ctypes.c_int(C),
                               ctypes.c_int({"sum": 0, "min": 1, "max": 2}[mode]), _p(out))
    return out


def segmented_maxpool(values, begin, end):
    values, begin, end = _f32(values), _i32(begin), _i32(end)
    P, C = begin.shape[0], values.shape[1]
    pooled = np.zeros((P, C), np.float32)
    arg = np.zeros((P, C), np.int32)
    lib().orc_segmented_maxpool_fwd(_p(values), _p(begin), _p(end), ctypes.c_int64(P), ctypes.c_int(C),
                                    _p(pooled), _p(arg))
    return pooled, arg


def segmented_maxpool_bwd(dpooled, argmax, M):
    dpooled, argmax = _f32(dpooled), _i32(argmax)
    P, C = dpooled.shape
    dv = np.zeros((M, C), np.float32)
    lib().orc_segmented_maxpool_bwd(_p(dpooled), _p(argmax), ctypes.c_int64(P), ctypes.c_int(C),
                                    ctypes.c_int64(M), _p(dv))
    return dv


def instance_iou(proposal_offsets, instance_labels, batch_indices, num_points_per_instance):
    po, il, bi = _i32(proposal_offsets), _i32(instance_labels), _i32(batch_indices)
    npi = _i32(num_points_per_instance)
    P, (B, I) = po.shape[0] - 1, npi.shape
    out = np.zeros((P, I), np.float32)
    lib().orc_instance_iou(_p(po), _p(il), _p(bi), _p(npi), ctypes.c_int64(P), ctypes.c_int64(B),
                           ctypes.c_int(I), _p(out))
    return out


def nms(ious, scores, threshold):
    ious, scores = _f32(ious), _f32(scores)
    P = scores.shape[0]
    order = _i32(np.argsort(-scores, kind="stable"))
    keep = np.zeros((max(P, 1),), np.int32)
    n = lib().orc_nms(_p(ious), _p(order), ctypes.c_int64(P), ctypes.c_float(threshold), _p(keep))
    return keep[:n].astype(np.int64)


# ---------------------------------------------------------------------------------------------- F
def pn2_ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_pn2_ball_query(b, n, m, ctypes.c_float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


def pn2_group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, npts, ns = idx.shape
    out = np.zeros((b, c, npts, ns), np.float32)
    lib().orc_pn2_group_points(b, c, n, npts, ns, _p(points), _p(idx), _p(out))
    return out


def pn2_group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, npts, ns = grad_out.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().orc_pn2_group_points_grad(b, c, n, npts, ns, _p(grad_out), _p(idx), _p(gp))
    return gp


def pn2_gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().orc_pn2_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def pn2_gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().orc_pn2_gather_points_grad(b, c, n, m, _p(grad_out), _p(idx), _p(gp))
    return gp


def pn2_furthest_point_sampling(xyz, npoint):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)
    idx = np.zeros((b, npoint), np.int32)
    lib().orc_pn2_furthest_point_sampling(b, n, npoint, _p(xyz), _p(temp), _p(idx))
    return idx


def pn2_three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().orc_pn2_three_nn(b, n, m, _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def pn2_knn(unknown, known, k):
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((b, n, k), np.float32)
    idx = np.zeros((b, n, k), np.int32)
    lib().orc_pn2_knn(b, n, m, k, _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def pn2_three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().orc_pn2_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def pn2_three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    gp = np.zeros((b, c, m), np.float32)
    lib().orc_pn2_three_interpolate_grad(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(gp))
    return gp
