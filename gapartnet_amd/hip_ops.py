"""Raw (no autograd) HIP operators: torch CUDA tensors in/out, every computation inside libgpn_hip.so.

PyTorch is used here only for device memory (caching allocator) and the current HIP stream.  Every function
raises if given non-CUDA tensors — there is no CPU path in the product (the CPU oracle lives in ``oracle/``
and is test infrastructure).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _C
from ._C import check, f32, host_f32x3, host_i32x3, i32, i64, ptr, szt

TILE_ROWS = 32
PACK_TRANSPOSE = 1
PACK_REVERSE = 2
name = "hip"


def _stream():
    # raw current-stream handle (the C call behind torch.cuda.current_stream(), ~20x cheaper than building the wrapper)
    return _C.ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _dev(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _C.GpnError("gapartnet_amd HIP operators need CUDA(HIP) tensors; got a CPU tensor "
                              "(there is no CPU fallback in the product path)")
        dev = t.device
    return dev


def _c(t, dtype):
    if t is None:
        return None
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


_WS_CACHE = {}


def _ws(nbytes, device):
    """scratch workspace for one kernel call.  One grow-only buffer per (device, stream) is reused by every call:
    kernels on a stream run in order, so a later call can only overwrite scratch an earlier one is done with."""
    nbytes = max(int(nbytes), 256)
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device()))
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(nbytes * 2, 64 << 20),), dtype=torch.uint8, device=device)
        _WS_CACHE[key] = buf
    return buf


def n_tiles(n):
    return (int(n) + TILE_ROWS - 1) // TILE_ROWS


@dataclass
class Rulebook:
    """K pair lists ordered by (tap, dst) + per-32-row tile offsets (see include/gpn.h)."""
    pair_src: torch.Tensor
    pair_dst: torch.Tensor
    tile_off: torch.Tensor
    K: int
    n_src: int
    n_dst: int
    num_pairs: torch.Tensor  # 0-dim int64 on device (no host sync)
    nbr: Optional[torch.Tensor] = None  # tap-major neighbour table [K*n_dst + 1] i32 (-1 = none) the fused conv gathers from
    # optional tile order of the large levels (gpn_rulebook_tile_order): the table with its columns sorted by neighbour
    # mask inside 16384-row blocks, and the destination row of every tile position
    nbr_p: Optional[torch.Tensor] = None
    perm: Optional[torch.Tensor] = None
    # device-counted rulebooks (section DEV): n_src / n_dst above are the buffers' bounds, these the live counts' counters
    live_src: Optional["DevCount"] = None
    live_dst: Optional["DevCount"] = None

    def pairs_host(self) -> int:
        return int(self.num_pairs.item())


class ArenaRulebook:
    """A Rulebook whose tables are slices of ONE arena (gpn_backbone_prepare).  The network executor only needs addresses
    (``ptrs`` = nbr, nbr_p, perm, pair_src, pair_dst, tile_off; 0 = absent) - 28 rulebooks x 7 tables per batch were ~150 tensor
    views made and ~250 addresses read back per step for that; the views are made when something asks for them (the module-by-
    module path, the tests), through the same attribute names."""
    __slots__ = ("_owner", "_spec", "_made", "K", "n_src", "n_dst", "live_src", "live_dst", "ptrs")
    _FIELDS = ("nbr", "nbr_p", "perm", "pair_src", "pair_dst", "tile_off", "num_pairs")

    def __init__(self, owner, spec, K, n_src, n_dst):
        self._owner, self._spec, self._made = owner, spec, {}
        self.K, self.n_src, self.n_dst = K, n_src, n_dst
        self.live_src = self.live_dst = None
        base = owner.arena_ptr
        self.ptrs = tuple(base + spec[f][0] if spec.get(f) is not None else 0 for f in self._FIELDS[:6])

    def __getattr__(self, name):  # (only reached for names that are not slots: the table fields)
        if name not in ArenaRulebook._FIELDS:
            raise AttributeError(name)
        made = self._made
        if name not in made:
            ent = self._spec.get(name)
            t = None
            if ent is not None:
                off, count, dtype, shape = ent
                t = self._owner._view(off, count, dtype)
                t = t[0] if shape == () else (t.view(*shape) if shape is not None else t)
            made[name] = t
        return made[name]

    def pairs_host(self) -> int:
        return int(self.num_pairs.item())


class DevCount:
    """A row count that is still on the device (include/gpn.h section DEV): ``t`` = int64 [1] device tensor (usually a view
    into the counts array of the launch that produced it), ``plan`` = the host's estimate of its value (0 = none; it sizes
    grids and picks kernel variants, results never depend on it).  Tensors that go with it are allocated for a BOUND."""
    __slots__ = ("t", "plan")

    def __init__(self, t: torch.Tensor, plan: int = 0):
        assert t.dtype == torch.int64 and t.numel() == 1 and t.is_cuda
        self.t, self.plan = t, max(int(plan), 0)

    @property
    def ptr(self):
        return _C.ctypes.c_void_p(self.t.data_ptr())

    def args(self):
        """(device pointer, plan) as ctypes arguments"""
        return _C.ctypes.c_void_p(self.t.data_ptr()), _C.ctypes.c_int64(self.plan)


# ---------------------------------------------------------------------------------------------------- V
def voxelize(points, feats, seg_offsets, seg_range_min, seg_range_max, voxel_size, grid_dims, want_csr=False,
             want_stats=False):
    """Kernel V. Returns (voxel_feats [V,C], voxel_coords [V,3] i32, voxel_seg [V] i32, pc_voxel_id [M] i32
    [, point_order [M] i32, voxel_point_start [V+1] i32] [, stats]).  One host sync (number of voxels); with
    ``want_stats`` the same read also brings stats = {"max_coord": [3 ints], "dropped": points outside the grid}."""
    dev = _dev(points, feats, seg_offsets)
    points, feats = _c(points, torch.float32), _c(feats, torch.float32)
    seg_offsets = _c(seg_offsets, torch.int64)
    M, C, S = points.shape[0], feats.shape[1], seg_offsets.shape[0] - 1
    rmin = _c(seg_range_min.reshape(-1, 3).expand(S, 3), torch.float32)
    rmax = _c(seg_range_max.reshape(-1, 3).expand(S, 3), torch.float32)
    vf = torch.empty((M, C), dtype=torch.float32, device=dev)
    vc = torch.empty((M, 3), dtype=torch.int32, device=dev)
    vseg = torch.empty((M,), dtype=torch.int32, device=dev)
    pid = torch.empty((M,), dtype=torch.int32, device=dev)
    nv = torch.empty((1,), dtype=torch.int64, device=dev)  # (always written by the library)
    order = torch.empty((M,), dtype=torch.int32, device=dev) if want_csr else None
    vstart = torch.empty((M + 1,), dtype=torch.int32, device=dev) if want_csr else None
    L = _C.lib()
    ws = _ws(L.gpn_voxelize_ws_bytes(i64(M), i32(C)), dev)
    check(L.gpn_voxelize_ex(ptr(points), ptr(feats), ptr(seg_offsets), ptr(rmin), ptr(rmax), i64(M), i32(C),
                            i64(S), host_f32x3(voxel_size), host_i32x3(grid_dims), ptr(vf), ptr(vc), ptr(vseg),
                            ptr(pid), ptr(nv), ptr(order), ptr(vstart), ptr(ws), szt(ws.numel()), _stream()),
          "gpn_voxelize")
    stats = None
    if want_stats:
        live = torch.arange(M, device=dev)[:, None] < nv
        head = torch.cat([nv, torch.where(live, vc, torch.zeros_like(vc)).amax(0).to(torch.int64) if M > 0 else
                          torch.zeros((3,), dtype=torch.int64, device=dev), (pid < 0).sum()[None]]).tolist()
        V, stats = head[0], {"max_coord": head[1:4], "dropped": head[4]}
    else:
        V = int(nv.item())
    out = (vf[:V], vc[:V], vseg[:V], pid)
    if want_csr:
        out = out + (order, vstart[:V + 1])
    if want_stats:
        out = out + (stats,)
    return out


_PINNED_STATS = []  # pinned int64 buffers of finished voxelize_scenes_begin / _finish pairs, for reuse


def voxelize_scenes_begin(points, feats, seg_offsets, voxel_size, n_levels=0, sorted_form=False):
    """first half of ``voxelize_scenes``: every launch of gpn_voxelize_scenes plus an asynchronous copy of its statistics to
    pinned memory - NO host read.  ``voxelize_scenes_finish(handle)`` is the second half; issued a step later (the device
    prefetcher does: dataset/prefetch.py) it finds the statistics there and does not wait."""
    dev = _dev(points, feats, seg_offsets)
    points, feats = _c(points, torch.float32), _c(feats, torch.float32)
    seg_offsets = _c(seg_offsets, torch.int64)
    M, C, S = points.shape[0], feats.shape[1], seg_offsets.shape[0] - 1
    vf = torch.empty((M, C), dtype=torch.float32, device=dev)
    idx4 = torch.empty((max(M, 1), 4), dtype=torch.int32, device=dev)
    pid = torch.empty((M,), dtype=torch.int32, device=dev)
    order = torch.empty((M,), dtype=torch.int32, device=dev)
    vstart = torch.empty((M + 1,), dtype=torch.int32, device=dev)
    stats = torch.empty((8 + n_levels,), dtype=torch.int64, device=dev)
    L = _C.lib()
    ws = _ws(L.gpn_voxelize_scenes_ws_bytes(i64(M), i32(C), i64(S), i32(n_levels)), dev)
    # (sorted_form: the stable-sort implementation the sort-free one replaced in round 5 - kept as the fallback for grids that do
    # not fit the occupancy bitmap, and what the tests compare the sort-free form with)
    fn = L.gpn_voxelize_scenes_sorted if sorted_form else L.gpn_voxelize_scenes
    check(fn(ptr(points), ptr(feats), ptr(seg_offsets), i64(M), i32(C), i64(S), host_f32x3(voxel_size),
             i32(n_levels), ptr(vf), ptr(idx4), ptr(pid), ptr(order), ptr(vstart), ptr(stats), ptr(ws),
             szt(ws.numel()), _stream()), "gpn_voxelize_scenes")
    host = None
    for i, buf in enumerate(_PINNED_STATS):
        if buf.numel() == stats.numel():
            host = _PINNED_STATS.pop(i)
            break
    if host is None:
        host = torch.empty((stats.numel(),), dtype=torch.int64).pin_memory()
    host.copy_(stats, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return dict(vf=vf, idx4=idx4, pid=pid, order=order, vstart=vstart, stats=stats, host=host, done=done, n_levels=n_levels)


def voxelize_scenes_finish(handle):
    """second half: the one host read of batch preparation (of pinned memory, after the copy's event) and the slicing"""
    handle["done"].synchronize()
    st = handle["host"].tolist()
    if len(_PINNED_STATS) < 8:
        _PINNED_STATS.append(handle["host"])
    if st[5] != 0:
        return None
    V, n_levels = st[0], handle["n_levels"]
    return (handle["vf"][:V], handle["idx4"][:V], handle["pid"], handle["order"], handle["vstart"][:V + 1], st[1:4], st[4],
            st[8:8 + n_levels])


def voxelize_scenes(points, feats, seg_offsets, voxel_size, n_levels=0, sorted_form=False):
    """Scene-batch voxelisation with the reference's per-scene ranges and NO host read before or between the launches
    (gpn_voxelize_scenes).  -> (voxel_feats [V,C], indices [V,4] i32 = (scene,x,y,z), pc_voxel_id [M] i32, point_order [M],
    voxel_point_start [V+1], max_coord [3 ints], dropped, level_counts [n_levels ints]) after ONE host read, or None when
    a cell index did not fit the packed keys (>= 1024 cells along an axis: the caller takes voxelize() instead)."""
    got = voxelize_scenes_finish(voxelize_scenes_begin(points, feats, seg_offsets, voxel_size, n_levels, sorted_form))
    if got is None and not sorted_form:  # (stats[5] = 2: the batch's grid does not fit the sort-free form's bitmap)
        got = voxelize_scenes_finish(voxelize_scenes_begin(points, feats, seg_offsets, voxel_size, n_levels, True))
    return got


# ---------------------------------------------------------------------------------------------------- BP
class PreparedBackbone:
    """what gpn_backbone_prepare left in its arena: ``run()`` is the blocking native call (meant for a worker thread: ctypes
    releases the interpreter lock for its whole duration), ``wrap()`` turns the descriptor into tensors / Rulebooks (views of the
    arena)"""
    KHEAD, KRB = 16, 10
    KLEVEL = 8 + 4 * KRB

    def __init__(self, xyz, feats, seg_offsets, voxel_size, n_levels, ident_levels, stream):
        dev = _dev(xyz, feats, seg_offsets)
        self.xyz, self.feats, self.seg_offsets = _c(xyz, torch.float32), _c(feats, torch.float32), _c(seg_offsets, torch.int64)
        self.voxel_size, self.n_levels, self.ident_levels = [float(v) for v in voxel_size], int(n_levels), int(ident_levels)
        self.M, self.C, self.S = int(self.xyz.shape[0]), int(self.feats.shape[1]), int(self.seg_offsets.shape[0]) - 1
        L = _C.lib()
        L.gpn_backbone_prepare_arena_bytes.restype = _C.ctypes.c_size_t
        need = int(L.gpn_backbone_prepare_arena_bytes(i64(self.M), i32(self.C), i64(self.S), i32(self.n_levels)))
        self.arena = torch.empty((need,), dtype=torch.uint8, device=dev)  # (on the caller's current stream)
        self.desc = np.full((int(L.gpn_backbone_prepare_desc_words(i32(self.n_levels))),), -1, np.int64)
        self.pinned = torch.empty((8 + self.n_levels,), dtype=torch.int64).pin_memory()
        self.stream_handle = int(stream.cuda_stream)
        self.device_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.rc, self.error = None, None
        self._typed = {}
        self.arena_ptr = self.arena.data_ptr()

    def run(self):
        L = _C.lib()
        rc = L.gpn_backbone_prepare(ptr(self.xyz), ptr(self.feats), ptr(self.seg_offsets), i64(self.M), i32(self.C), i64(self.S),
                                    host_f32x3(self.voxel_size), i32(self.n_levels), _C.ctypes.c_uint32(self.ident_levels),
                                    i64(TILE_ORDER_MIN_ROWS), i32(TILE_ORDER_BLOCK), i32(self.device_index), ptr(self.arena),
                                    szt(self.arena.numel()), _C.ctypes.c_void_p(self.desc.ctypes.data),
                                    _C.ctypes.c_void_p(self.pinned.data_ptr()), _C.ctypes.c_void_p(self.stream_handle))
        self.rc = int(rc)
        if rc:
            self.error = L.gpn_last_error().decode("utf-8", "replace")
        return self

    def fallback(self) -> bool:
        return self.rc != 0 or int(self.desc[5]) != 0

    _ITEM = {torch.int32: 4, torch.float32: 4, torch.int64: 8}

    def _view(self, off, count, dtype):
        """``count`` elements of ``dtype`` at byte offset ``off`` of the arena (one slicing op on a typed view of the whole arena:
        ~150 of these per batch)"""
        if off < 0:
            return None
        base = self._typed.get(dtype)
        if base is None:
            size = self._ITEM[dtype]
            base = self._typed[dtype] = self.arena[:self.arena.numel() // size * size].view(dtype)
        first = off // self._ITEM[dtype]  # (the library aligns every table to 256 bytes)
        return base[first:first + count]

    def _rulebook(self, words):
        o_nbr, o_src, o_dst, o_toff, o_np, o_nbrp, o_perm, n_src, n_dst, K = (int(v) for v in words)
        if o_nbr < 0:
            return None
        # pair-list capacities as gpn_backbone_prepare carved them: 27 n (SubM), the fine row count (stride-2 maps), n (identity)
        cap = 27 * n_dst if K == 27 else (n_dst if K == 1 else max(n_src, n_dst))
        i32 = torch.int32
        spec = dict(pair_src=(o_src, cap, i32, None), pair_dst=(o_dst, cap, i32, None),
                    tile_off=(o_toff, K * (n_tiles(n_dst) + 1), i32, (K, -1)), num_pairs=(o_np, 1, torch.int64, ()),
                    nbr=(o_nbr, K * n_dst + 1, i32, None))
        if o_perm >= 0:
            spec["perm"] = (o_perm, (n_dst + 15) // 16 * 16 + 16, i32, None)
            spec["nbr_p"] = (o_nbrp, K * n_dst + 1, i32, None)
        return ArenaRulebook(self, spec, K, n_src, n_dst)

    def wrap(self):
        """-> dict(features, indices, spatial_shape, pc_voxel_id, csr, level_counts, levels = [dict(indices, shape, subm,
        down_fwd, down_bwd, ident)])"""
        d = self.desc
        V = int(d[0])
        M, C = self.M, self.C
        out = dict(features=self._view(int(d[8]), M * C, torch.float32).view(M, C)[:V],
                   indices=self._view(int(d[9]), M * 4, torch.int32).view(M, 4)[:V], spatial_shape=[int(v) for v in d[1:4]],
                   pc_voxel_id=self._view(int(d[10]), M, torch.int32),
                   csr=(self._view(int(d[11]), M, torch.int32), self._view(int(d[12]), M + 1, torch.int32)[:V + 1]),
                   dropped=int(d[4]), levels=[])
        for l in range(self.n_levels):
            w = d[self.KHEAD + l * self.KLEVEL: self.KHEAD + (l + 1) * self.KLEVEL]
            rows = int(w[0])
            idx = out["indices"] if l == 0 else self._view(int(w[4]), rows * 4, torch.int32).view(rows, 4)
            rbs = [self._rulebook(w[8 + k * self.KRB: 8 + (k + 1) * self.KRB]) for k in range(4)]
            out["levels"].append(dict(indices=idx, shape=[int(v) for v in w[1:4]], subm=rbs[0], down_fwd=rbs[1], down_bwd=rbs[2],
                                      ident=rbs[3]))
        out["level_counts"] = [int(lv["indices"].shape[0]) for lv in out["levels"][1:]]
        return out


# ---------------------------------------------------------------------------------------------------- K
def rulebook_subm3(indices, spatial_shape, rows: Optional[DevCount] = None) -> Rulebook:
    """``rows``: the live row count of ``indices`` is a device counter (indices.shape[0] = its bound): every buffer is sized for
    the bound, the tables are laid out for the live count, no tile order"""
    dev = _dev(indices)
    indices = _c(indices, torch.int32)
    N = indices.shape[0]
    cap = max(27 * N, 1)
    src = torch.empty((cap,), dtype=torch.int32, device=dev)
    dst = torch.empty((cap,), dtype=torch.int32, device=dev)
    toff = torch.empty((27, n_tiles(N) + 1), dtype=torch.int32, device=dev)
    npairs = torch.empty((1,), dtype=torch.int64, device=dev)  # (always written by the library)
    L = _C.lib()
    ws = _ws(L.gpn_rulebook_subm3_ws_bytes(i64(N)), dev)
    nbr = torch.empty((27 * N + 1,), dtype=torch.int32, device=dev)
    if rows is not None:
        check(L.gpn_rulebook_subm3_dev(ptr(indices), i64(N), *rows.args(), host_i32x3(spatial_shape), ptr(nbr), ptr(src), ptr(dst),
                                       ptr(toff), ptr(npairs), ptr(ws), szt(ws.numel()), _stream()), "gpn_rulebook_subm3_dev")
        return Rulebook(src, dst, toff, 27, N, N, npairs[0], nbr, live_src=rows, live_dst=rows)
    check(L.gpn_rulebook_subm3(ptr(indices), i64(N), host_i32x3(spatial_shape), ptr(nbr), ptr(src), ptr(dst), ptr(toff),
                               ptr(npairs), ptr(ws), szt(ws.numel()), _stream()), "gpn_rulebook_subm3")
    return _with_tile_order(Rulebook(src, dst, toff, 27, N, N, npairs[0], nbr))


import os as _os
# Rows re-ordered by neighbour mask (gpn_rulebook_tile_order) for the masked-tile conv kernel (csrc/spconv_tiles.hip): a
# tile executes a tap as soon as ANY of its rows has that neighbour.  In voxel order a level-0 tile of the bench scenes runs
# 4.3x the MFMA row-slots of its useful pairs, sorted by mask inside 16384-row blocks 2.1x (levels 1 / 2: 2.2x / 2.0x ->
# 1.4x; stride-2 convs 4.4x -> 1.3x, inverse convs 4.3x -> 1.0x; tools/conv_tiles_bench.py).  The order costs a block-local sort (one workgroup per 16384 rows)
# and a permuted table per rulebook; levels below 4096 tiles run on the direct kernel and keep voxel order.
# (round 4: also level 2 of the bench - 25k rows, direct / split kernel through `perm`: 8.17 / 8.23 / 8.14 -> 8.10 / 8.10 / 8.09 ms
# per step in three interleaved same-box runs, profiles/r04_findings.md; 4096 gave nothing more)
TILE_ORDER_MIN_ROWS = 16384  # (module attributes: tests and tools set them to order small inputs)
TILE_ORDER_BLOCK = 16384


def _with_tile_order(rb: "Rulebook") -> "Rulebook":
    if rb.nbr is not None and rb.n_dst >= TILE_ORDER_MIN_ROWS:
        rb.perm, rb.nbr_p = tile_order(rb.nbr, rb.K, rb.n_dst)
    return rb


def tile_order(nbr, K, n):
    """-> (perm [ceil(n/16)*16 + 16] i32, nbr_p [K*n + 1] i32): rows of every 16384-row block sorted by neighbour mask"""
    dev = nbr.device
    L = _C.lib()
    perm = torch.empty(((n + 15) // 16 * 16 + 16,), dtype=torch.int32, device=dev)
    nbr_p = torch.empty((K * n + 1,), dtype=torch.int32, device=dev)
    ws = _ws(L.gpn_rulebook_tile_order_ws_bytes(i64(n)), dev)
    check(L.gpn_rulebook_tile_order(ptr(nbr), i32(K), i64(n), i32(TILE_ORDER_BLOCK), ptr(perm), ptr(nbr_p), ptr(ws),
                                    szt(ws.numel()), _stream()), "gpn_rulebook_tile_order")
    return perm, nbr_p




def rulebook_identity(n, device, rows_dev: Optional[DevCount] = None) -> Rulebook:
    """the K = 1 rulebook over ``n`` rows (SubMConv3d(k=1), linear layers on the conv kernels) in one launch; ``rows_dev``: n is
    a bound, the live count a device counter"""
    rows = torch.empty((max(n, 1),), dtype=torch.int32, device=device)[:n]
    tile_off = torch.empty((1, n_tiles(n) + 1), dtype=torch.int32, device=device)
    nbr = torch.empty((n + 1,), dtype=torch.int32, device=device)
    npairs = torch.empty((1,), dtype=torch.int64, device=device)
    if rows_dev is not None:
        check(_C.lib().gpn_rulebook_identity_dev(i64(n), *rows_dev.args(), ptr(rows), ptr(tile_off), ptr(nbr), ptr(npairs), _stream()),
              "gpn_rulebook_identity_dev")
        return Rulebook(rows, rows, tile_off, 1, n, n, npairs[0], nbr, live_src=rows_dev, live_dst=rows_dev)
    check(_C.lib().gpn_rulebook_identity(i64(n), ptr(rows if n > 0 else nbr), ptr(tile_off), ptr(nbr), ptr(npairs), _stream()),
          "gpn_rulebook_identity")
    return Rulebook(rows, rows, tile_off, 1, n, n, npairs[0], nbr)


def rulebook_level_counts(indices, spatial_shape, batch_size, n_levels):
    """-> device tensor [n_levels] i64: rows of each of the next ``n_levels`` stride-2 levels below ``indices`` (what the
    successive rulebook_down calls would report), so that a caller can fetch them all with ONE host read."""
    dev = _dev(indices)
    indices = _c(indices, torch.int32)
    N = indices.shape[0]
    counts = torch.empty((n_levels,), dtype=torch.int64, device=dev)
    L = _C.lib()
    ws = _ws(L.gpn_rulebook_level_counts_ws_bytes(i64(N), i32(n_levels)), dev)
    check(L.gpn_rulebook_level_counts(ptr(indices), i64(N), ptr(None), i64(batch_size), host_i32x3(spatial_shape), i32(n_levels),
                                      ptr(counts), ptr(ws), szt(ws.numel()), _stream()), "gpn_rulebook_level_counts")
    return counts


def rulebook_down(indices, spatial_shape, batch_size, n_out=None):
    """Returns (out_indices [No,4] i32, out_shape, rb_fwd (dst=coarse), rb_bwd (dst=fine)).  One host sync, none when the
    caller already knows the coarse row count ``n_out`` (rulebook_level_counts / the proposal stage's counts)."""
    dev = _dev(indices)
    indices = _c(indices, torch.int32)
    N = indices.shape[0]
    out_idx = torch.empty((max(N, 1), 4), dtype=torch.int32, device=dev)
    f2c = torch.empty((max(N, 1),), dtype=torch.int32, device=dev)
    tap = torch.empty((max(N, 1),), dtype=torch.int32, device=dev)
    nout = torch.empty((1,), dtype=torch.int64, device=dev)  # (always written by the library)
    L = _C.lib()
    ws = _ws(L.gpn_rulebook_down_ws_bytes(i64(N)), dev)
    check(L.gpn_rulebook_down(ptr(indices), i64(N), i64(batch_size), host_i32x3(spatial_shape), ptr(out_idx),
                              ptr(f2c), ptr(tap), ptr(nout), ptr(ws), szt(ws.numel()), _stream()),
          "gpn_rulebook_down")
    No = int(nout.item()) if n_out is None else int(n_out)
    cap = max(N, 1)
    fs = torch.empty((cap,), dtype=torch.int32, device=dev)
    fd = torch.empty((cap,), dtype=torch.int32, device=dev)
    bs = torch.empty((cap,), dtype=torch.int32, device=dev)
    bd = torch.empty((cap,), dtype=torch.int32, device=dev)
    ft = torch.empty((8, n_tiles(No) + 1), dtype=torch.int32, device=dev)
    bt = torch.empty((8, n_tiles(N) + 1), dtype=torch.int32, device=dev)
    npairs = torch.empty((1,), dtype=torch.int64, device=dev)  # (always written by the library)
    ws = _ws(L.gpn_rulebook_down_lists_ws_bytes(i64(N), i64(No)), dev)
    fn = torch.empty((8 * No + 1,), dtype=torch.int32, device=dev)
    bn = torch.empty((8 * N + 1,), dtype=torch.int32, device=dev)
    check(L.gpn_rulebook_down_lists(ptr(f2c), ptr(tap), i64(N), i64(No), ptr(fn), ptr(fs), ptr(fd), ptr(ft), ptr(bn),
                                    ptr(bs), ptr(bd), ptr(bt), ptr(npairs), ptr(ws), szt(ws.numel()), _stream()),
          "gpn_rulebook_down_lists")
    out_shape = [int(s) // 2 for s in spatial_shape]
    rb_fwd = Rulebook(fs, fd, ft, 8, N, No, npairs[0], fn)  # (dst = coarse: 1-8 children per row, the order buys nothing: 10.7 -> 10 us)
    rb_bwd = _with_tile_order(Rulebook(bs, bd, bt, 8, No, N, npairs[0], bn))  # dst = fine: one tap per row, 4.3x -> 1.0x slots
    return out_idx[:No], out_shape, rb_fwd, rb_bwd


def rulebook_down_dev(indices, spatial_shape, batch_size, rows: DevCount, batch: Optional[DevCount], out_plan: int = 0):
    """rulebook_down with the fine row count (and optionally the number of batch entries - the proposals of a step) on the
    device: no host read.  ``indices.shape[0]`` / ``batch_size`` are bounds.  -> (out_indices [bound,4], out_shape, rb_fwd,
    rb_bwd, coarse rows as a DevCount with plan ``out_plan``)"""
    dev = _dev(indices)
    indices = _c(indices, torch.int32)
    N = indices.shape[0]
    out_idx = torch.empty((N, 4), dtype=torch.int32, device=dev)
    f2c = torch.empty((N,), dtype=torch.int32, device=dev)
    tap = torch.empty((N,), dtype=torch.int32, device=dev)
    nout = torch.empty((1,), dtype=torch.int64, device=dev)
    L = _C.lib()
    shape = host_i32x3(spatial_shape)
    ws = _ws(L.gpn_rulebook_down_dev_ws_bytes(i64(N), i64(batch_size), shape), dev)
    b_ptr, b_plan = batch.args() if batch is not None else (_C.ctypes.c_void_p(0), _C.ctypes.c_int64(0))
    check(L.gpn_rulebook_down_dev(ptr(indices), i64(N), *rows.args(), i64(batch_size), b_ptr, b_plan, shape, ptr(out_idx), ptr(f2c),
                                  ptr(tap), ptr(nout), ptr(ws), szt(ws.numel()), _stream()), "gpn_rulebook_down_dev")
    out_rows = DevCount(nout, out_plan)
    No = N  # bound of the coarse level
    fs = torch.empty((N,), dtype=torch.int32, device=dev)
    fd = torch.empty((N,), dtype=torch.int32, device=dev)
    bs = torch.empty((N,), dtype=torch.int32, device=dev)
    bd = torch.empty((N,), dtype=torch.int32, device=dev)
    ft = torch.empty((8, n_tiles(No) + 1), dtype=torch.int32, device=dev)
    bt = torch.empty((8, n_tiles(N) + 1), dtype=torch.int32, device=dev)
    npairs = torch.empty((1,), dtype=torch.int64, device=dev)
    ws = _ws(L.gpn_rulebook_down_lists_ws_bytes(i64(N), i64(No)), dev)
    fn = torch.empty((8 * No + 1,), dtype=torch.int32, device=dev)
    bn = torch.empty((8 * N + 1,), dtype=torch.int32, device=dev)
    check(L.gpn_rulebook_down_lists_dev(ptr(f2c), ptr(tap), i64(N), *rows.args(), i64(No), *out_rows.args(), ptr(fn), ptr(fs), ptr(fd),
                                        ptr(ft), ptr(bn), ptr(bs), ptr(bd), ptr(bt), ptr(npairs), ptr(ws), szt(ws.numel()), _stream()),
          "gpn_rulebook_down_lists_dev")
    out_shape = [int(s) // 2 for s in spatial_shape]
    rb_fwd = Rulebook(fs, fd, ft, 8, N, No, npairs[0], fn, live_src=rows, live_dst=out_rows)
    rb_bwd = Rulebook(bs, bd, bt, 8, No, N, npairs[0], bn, live_src=out_rows, live_dst=rows)
    return out_idx, out_shape, rb_fwd, rb_bwd, out_rows


# ---------------------------------------------------------------------------------------------------- C
LAYOUT_OKI = 4


def _wdims(W, layout):
    """(K, cin, cout) of a weight given as canonical [K, Cin, Cout] ("kio") or parameter layout [Cout, K, Cin] ("oki")"""
    if layout == "oki":
        return W.shape[1], W.shape[2], W.shape[0]
    return W.shape[0], W.shape[1], W.shape[2]


def pack_weights(W, flags, layout="kio"):
    dev = _dev(W)
    W = _c(W, torch.float32)
    K, cin, cout = _wdims(W, layout)
    packed = torch.empty((K * cin * cout,), dtype=torch.float32, device=dev)
    if layout == "oki":
        flags |= LAYOUT_OKI
    check(_C.lib().gpn_spconv_pack_weights(ptr(W), i32(K), i32(cin), i32(cout), i32(flags), ptr(packed), _stream()),
          "gpn_spconv_pack_weights")
    return packed


def _conv_packed(features, packed, rb: Rulebook, cin, cout):
    dev = _dev(features)
    features = _c(features, torch.float32)
    assert features.shape[0] == rb.n_src and features.shape[1] == cin, (features.shape, rb.n_src, cin)
    out = torch.empty((rb.n_dst, cout), dtype=torch.float32, device=dev)
    L = _C.lib()
    ws_bytes = L.gpn_spconv_fwd_ws_bytes(i32(rb.K), i64(rb.n_dst), i32(cin), i32(cout))
    ws = _ws(ws_bytes, dev) if ws_bytes else None
    check(L.gpn_spconv_fwd(ptr(features), ptr(packed), ptr(rb.nbr), i32(rb.K), i64(rb.n_dst), i32(cin), i32(cout),
                           ptr(out), ptr(ws), szt(ws.numel() if ws is not None else 0), _stream()), "gpn_spconv_fwd")
    return out


_FAST_WS = {}


def _fast_ws(device, need=0):
    """(pointer, bytes) of the shared per-(device, stream) scratch buffer, grown on demand (see _ws)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch._C._cuda_getCurrentRawStream(idx)
    key = (idx, stream)
    ent = _FAST_WS.get(key)
    if ent is None or ent[1] < need:
        size = max(int(need) * 2, 256 << 20)
        buf = torch.empty((size,), dtype=torch.uint8, device=device)
        ent = (buf.data_ptr(), size, buf)
        _FAST_WS[key] = ent
    return ent[0], ent[1], stream


def _raise(what):
    raise _C.GpnError(f"{what} failed: {_C.lib().gpn_last_error().decode('utf-8', 'replace')}")


def _conv_w(features, W, layout, flags, rb: Rulebook, cin_op, cout_op):
    """pack + fused conv in one library call; features must be contiguous fp32 on the GPU (autograd wrappers ensure it)"""
    if not features.is_cuda:
        _dev(features)
    K, cin_w, cout_w = _wdims(W, layout)
    assert features.shape[0] == rb.n_src and features.shape[1] == cin_op, (features.shape, rb.n_src, cin_op)
    L = _C.lib()
    out = torch.empty((rb.n_dst, cout_op), dtype=torch.float32, device=features.device)
    if layout == "oki":
        flags |= LAYOUT_OKI
    ws_ptr, ws_size, stream = _fast_ws(features.device)
    rc = L.gpn_spconv_fwd_w(features.data_ptr(), W.data_ptr(), K, cin_w, cout_w, flags, rb.nbr.data_ptr(), rb.n_dst,
                            out.data_ptr(), ws_ptr, ws_size, stream)
    if rc == 2:  # workspace too small: grow once and retry
        ws_ptr, ws_size, stream = _fast_ws(features.device, L.gpn_spconv_fwd_w_ws_bytes(K, rb.n_dst, cin_op, cout_op))
        rc = L.gpn_spconv_fwd_w(features.data_ptr(), W.data_ptr(), K, cin_w, cout_w, flags, rb.nbr.data_ptr(), rb.n_dst,
                                out.data_ptr(), ws_ptr, ws_size, stream)
    if rc:
        _raise("gpn_spconv_fwd_w")
    return out


def conv_fwd_ordered(features, W, rb: Rulebook, flags=0):
    """conv over ``rb`` through gpn_spconv_fwd_ordered: uses the rulebook's tile order (nbr_p / perm) when it has one.
    ``W`` canonical [K, Cin, Cout]; ``flags`` = PACK_TRANSPOSE | PACK_REVERSE turns the call into the dgrad of a SubM conv
    (features = dout).  What the network executor does per layer; the per-layer autograd path (conv_fwd) passes the plain
    table only."""
    dev = _dev(features)
    features = _c(features, torch.float32)
    K, cin_w, cout_w = W.shape
    cin, cout = (cout_w, cin_w) if flags & PACK_TRANSPOSE else (cin_w, cout_w)
    assert features.shape == (rb.n_src, cin), (features.shape, rb.n_src, cin)
    packed = pack_weights(W, flags)
    out = torch.empty((rb.n_dst, cout), dtype=torch.float32, device=dev)
    L = _C.lib()
    ws_bytes = L.gpn_spconv_fwd_ws_bytes(i32(K), i64(rb.n_dst), i32(cin), i32(cout))
    ws = _ws(ws_bytes, dev) if ws_bytes else None
    check(L.gpn_spconv_fwd_ordered(ptr(features), ptr(packed), ptr(rb.nbr), ptr(rb.nbr_p), ptr(rb.perm), i32(K), i64(rb.n_dst),
                                   i32(cin), i32(cout), ptr(out), ptr(ws), szt(ws.numel() if ws is not None else 0), _stream()),
          "gpn_spconv_fwd_ordered")
    return out


def conv_fwd(features, W, rb: Rulebook, layout="kio"):
    """out[dst] = sum_k in[src] @ W[k];  channel counts multiples of 16."""
    K, cin, cout = _wdims(W, layout)
    return _conv_w(features, W, layout, 0, rb, cin, cout)


def conv_dgrad(dout, W, rb: Rulebook, rb_t: Rulebook, reverse_taps: bool, layout="kio"):
    """din[src] = sum_k dout[dst] @ W[k]^T, computed as a forward conv over the transposed rulebook rb_t."""
    K, cin, cout = _wdims(W, layout)
    return _conv_w(dout, W, layout, PACK_TRANSPOSE | (PACK_REVERSE if reverse_taps else 0), rb_t, cout, cin)


def conv_wgrad(features, dout, rb: Rulebook, layout="kio"):
    """weight gradient in the same layout as the weight was given ("kio": [K,Cin,Cout]; "oki": [Cout,K,Cin])"""
    if not (features.is_cuda and dout.is_cuda):
        _dev(features, dout)
    cin, cout = features.shape[1], dout.shape[1]
    shape = (cout, rb.K, cin) if layout == "oki" else (rb.K, cin, cout)
    dW = torch.empty(shape, dtype=torch.float32, device=features.device)
    L = _C.lib()
    ws_ptr, ws_size, stream = _fast_ws(features.device, 0)
    args = (features.data_ptr(), dout.data_ptr(), rb.pair_src.data_ptr(), rb.pair_dst.data_ptr(), rb.tile_off.data_ptr(),
            rb.K, rb.n_dst, cin, cout, LAYOUT_OKI if layout == "oki" else 0, dW.data_ptr())
    rc = L.gpn_spconv_wgrad(*args, ws_ptr, ws_size, stream)
    if rc == 2:
        ws_ptr, ws_size, stream = _fast_ws(features.device, L.gpn_spconv_wgrad_ws_bytes(i32(rb.K), i32(cin), i32(cout), i64(rb.n_dst)))
        rc = L.gpn_spconv_wgrad(*args, ws_ptr, ws_size, stream)
    if rc:
        _raise("gpn_spconv_wgrad")
    return dW


# ---------------------------------------------------------------------------------------------------- P
def point_losses_fwd(logits, labels, offsets, gt_offsets, instance_labels, ignore_index, metrics=False):
    """-> (losses [4] f32 = focal, dice, offset distance, offset direction; stats = opaque device block for the backward);
    ``metrics``: also (preds [M] i64 = argmax of the logits, accu [2] f32 = point accuracy over all points / over the points
    on a part) from the same pass (gpn_point_losses_fwd_metrics)"""
    dev = _dev(logits, offsets)
    logits, offsets, gt_offsets = _c(logits, torch.float32), _c(offsets, torch.float32), _c(gt_offsets, torch.float32)
    labels, instance_labels = _c(labels, torch.int64), _c(instance_labels, torch.int32)
    M, C = logits.shape
    losses = torch.empty((4,), dtype=torch.float32, device=dev)
    stats = torch.empty((4,), dtype=torch.float64, device=dev)
    L = _C.lib()
    ws = _ws(L.gpn_point_losses_ws_bytes(i64(M)), dev)
    if metrics:
        preds = torch.empty((M,), dtype=torch.int64, device=dev)
        accu = torch.empty((2,), dtype=torch.float32, device=dev)
        check(L.gpn_point_losses_fwd_metrics(ptr(logits), ptr(labels), ptr(offsets), ptr(gt_offsets), ptr(instance_labels), i64(M),
                                             i32(C), i64(ignore_index), ptr(losses), ptr(preds), ptr(accu), ptr(stats), ptr(ws),
                                             szt(ws.numel()), _stream()), "gpn_point_losses_fwd_metrics")
        return losses, stats, preds, accu
    check(L.gpn_point_losses_fwd(ptr(logits), ptr(labels), ptr(offsets), ptr(gt_offsets), ptr(instance_labels), i64(M),
                                 i32(C), i64(ignore_index), ptr(losses), ptr(stats), ptr(ws), szt(ws.numel()), _stream()),
          "gpn_point_losses_fwd")
    return losses, stats


def point_losses_bwd(logits, labels, offsets, gt_offsets, instance_labels, ignore_index, stats, grad_losses):
    dev = _dev(logits, offsets)
    M, C = logits.shape
    d_logits = torch.empty((M, C), dtype=torch.float32, device=dev)
    d_offsets = torch.empty((M, 3), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_point_losses_bwd(ptr(logits), ptr(labels), ptr(offsets), ptr(gt_offsets), ptr(instance_labels),
                                        i64(M), i32(C), i64(ignore_index), ptr(stats), ptr(_c(grad_losses, torch.float32)),
                                        ptr(d_logits), ptr(d_offsets), _stream()), "gpn_point_losses_bwd")
    return d_logits, d_offsets


# ---------------------------------------------------------------------------------------------------- BN
def _p(t):
    return None if t is None else t.data_ptr()


def bn_fwd(x, res, weight, bias, running_mean, running_var, training, momentum, eps, relu):
    """fused BatchNorm1d(+residual)(+ReLU) forward -> (y, mean, invstd); running stats updated in place when training.
    x / res must be contiguous fp32 (the autograd wrapper ensures it)."""
    if not x.is_cuda:
        _dev(x)
    N, C = x.shape
    y = torch.empty_like(x)
    L = _C.lib()
    if training:
        stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
        mean, invstd = stats[0], stats[1]
        ws_ptr, ws_size, stream = _fast_ws(x.device, 0)
        rc = L.gpn_bn_fwd_train(x.data_ptr(), _p(res), weight.data_ptr(), bias.data_ptr(), N, C, eps, momentum,
                                1 if relu else 0, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _p(running_mean),
                                _p(running_var), ws_ptr, ws_size, stream)
        if rc:
            _raise("gpn_bn_fwd_train")
    else:
        mean = running_mean
        invstd = 1.0 / torch.sqrt(running_var + eps)
        rc = L.gpn_bn_fwd_eval(x.data_ptr(), _p(res), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(),
                               invstd.data_ptr(), N, C, 1 if relu else 0, y.data_ptr(),
                               torch._C._cuda_getCurrentRawStream(x.device.index))
        if rc:
            _raise("gpn_bn_fwd_eval")
    return y, mean, invstd


def bn_bwd(x, y, dy, weight, mean, invstd, relu, training, has_res):
    """-> (dx, dres or None, dweight, dbias)"""
    if not x.is_cuda:
        _dev(x)
    N, C = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if has_res else None
    dwb = torch.empty((2, C), dtype=torch.float32, device=x.device)
    dw, db = dwb[0], dwb[1]
    ws_ptr, ws_size, stream = _fast_ws(x.device, 0)
    rc = _C.lib().gpn_bn_bwd(x.data_ptr(), y.data_ptr(), dy.data_ptr(), weight.data_ptr(), mean.data_ptr(),
                             invstd.data_ptr(), N, C, 1 if relu else 0, 1 if training else 0, dx.data_ptr(), _p(dres),
                             dw.data_ptr(), db.data_ptr(), ws_ptr, ws_size, stream)
    if rc:
        _raise("gpn_bn_bwd")
    return dx, dres, dw, db


# ---------------------------------------------------------------------------------------------------- G
def gather_rows(table, idx, rows: Optional[DevCount] = None):
    dev = _dev(table, idx)
    table, idx = _c(table, torch.float32), _c(idx, torch.int32)
    out = torch.empty((idx.shape[0], table.shape[1]), dtype=torch.float32, device=dev)
    if rows is not None:  # the live length of idx is a device counter (rows of `out` past it stay unwritten)
        check(_C.lib().gpn_gather_rows_dev(ptr(table), ptr(idx), i64(idx.shape[0]), *rows.args(), i32(table.shape[1]), ptr(out),
                                           _stream()), "gpn_gather_rows_dev")
        return out
    check(_C.lib().gpn_gather_rows(ptr(table), ptr(idx), i64(idx.shape[0]), i32(table.shape[1]), ptr(out), _stream()),
          "gpn_gather_rows")
    return out


def rows_csr(idx, n_rows):
    """CSR of positions grouped by row id (stable): (order [n] i32 with idx<0 entries last, starts [n_rows+1] i32)."""
    idx = idx.to(torch.int64)
    key = torch.where(idx >= 0, idx, torch.full_like(idx, n_rows))
    _, order = torch.sort(key, stable=True)
    counts = torch.bincount(key, minlength=n_rows + 1)[:n_rows]
    starts = torch.zeros((n_rows + 1,), dtype=torch.int32, device=idx.device)
    starts[1:] = counts.cumsum(0).to(torch.int32)
    return order.to(torch.int32), starts


def scatter_rows(dout, idx, n_rows, csr=None, rows: Optional[DevCount] = None):
    """transpose of gather_rows: dtable[r] = ordered sum of dout[i] over idx[i] == r.  ``rows``: the live number of table rows
    is a device counter (n_rows = its bound; needs ``csr``)."""
    dev = _dev(dout, idx)
    dout = _c(dout, torch.float32)
    order, starts = csr if csr is not None else rows_csr(idx, n_rows)
    out = torch.empty((n_rows, dout.shape[1]), dtype=torch.float32, device=dev)
    if rows is not None:
        assert csr is not None, "a device-counted scatter needs the caller's CSR"
        check(_C.lib().gpn_scatter_rows_csr_dev(ptr(dout), ptr(_c(order, torch.int32)), ptr(_c(starts, torch.int32)), i64(n_rows),
                                                *rows.args(), i32(dout.shape[1]), ptr(out), _stream()), "gpn_scatter_rows_csr_dev")
        return out
    check(_C.lib().gpn_scatter_rows_csr(ptr(dout), ptr(_c(order, torch.int32)), ptr(_c(starts, torch.int32)),
                                        i64(n_rows), i32(dout.shape[1]), ptr(out), _stream()),
          "gpn_scatter_rows_csr")
    return out


# ---------------------------------------------------------------------------------------------------- B/L
def ball_query(points, query, batch_indices, batch_offsets, radius, num_samples, point_labels=None,
               query_labels=None):
    dev = _dev(points, query)
    points, query = _c(points, torch.float32), _c(query, torch.float32)
    bi, bo = _c(batch_indices, torch.int32), _c(batch_offsets, torch.int32)
    pl, ql = _c(point_labels, torch.int32), _c(query_labels, torch.int32)
    Q, K = query.shape[0], int(num_samples)
    idx = torch.empty((Q, K), dtype=torch.int32, device=dev)
    cnt = torch.zeros((Q,), dtype=torch.int32, device=dev)
    L = _C.lib()
    Np = points.shape[0]
    if Np >= 2048 and radius > 0:  # grid-accelerated form (identical results); tiny inputs: the plain scan is one launch
        ws = _ws(L.gpn_ball_query_grid_ws_bytes(i64(Np)), dev)
        check(L.gpn_ball_query_grid(ptr(points), ptr(query), ptr(bi), ptr(bo), ptr(pl), ptr(ql), i64(Np), i64(Q),
                                    i64(bo.shape[0] - 1), f32(radius), i32(K), ptr(idx), ptr(cnt), ptr(ws),
                                    szt(ws.numel()), _stream()), "gpn_ball_query_grid")
        return idx, cnt
    check(L.gpn_ball_query(ptr(points), ptr(query), ptr(bi), ptr(bo), ptr(pl), ptr(ql), i64(Np),
                           i64(Q), i64(bo.shape[0] - 1), f32(radius), i32(K), ptr(idx), ptr(cnt), _stream()),
          "gpn_ball_query")
    return idx, cnt


def ccl(begin_end, edges, compacted=False):
    dev = _dev(begin_end, edges)
    be, edges = _c(begin_end, torch.int32), _c(edges, torch.int32)
    Q = be.shape[0] // 2
    labels = torch.empty((Q,), dtype=torch.int32, device=dev)
    L = _C.lib()
    ws = _ws(L.gpn_ccl_ws_bytes(i64(Q)), dev)
    check(L.gpn_ccl(ptr(be), ptr(edges), i64(Q), i64(edges.shape[0]), i32(1 if compacted else 0), ptr(labels),
                    ptr(ws), szt(ws.numel()), _stream()), "gpn_ccl")
    return labels


# ---------------------------------------------------------------------------------------------------- R/I/N
_MODES = {"sum": 0, "min": 1, "max": 2}


def segmented_reduce(values, begin, end, mode):
    dev = _dev(values, begin, end)
    values, begin, end = _c(values, torch.float32), _c(begin, torch.int32), _c(end, torch.int32)
    P, C = begin.shape[0], values.shape[1]
    out = torch.empty((P, C), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_segmented_reduce(ptr(values), ptr(begin), ptr(end), i64(P), i32(C), i32(_MODES[mode]), ptr(out),
                                        _stream()), "gpn_segmented_reduce")
    return out


def segmented_maxpool_fwd(values, begin, end, rows: Optional[DevCount] = None):
    dev = _dev(values, begin, end)
    values, begin, end = _c(values, torch.float32), _c(begin, torch.int32), _c(end, torch.int32)
    P, C = begin.shape[0], values.shape[1]
    pooled = torch.empty((P, C), dtype=torch.float32, device=dev)
    arg = torch.empty((P, C), dtype=torch.int32, device=dev)
    if rows is not None:  # the number of segments is a device counter (P = its bound)
        check(_C.lib().gpn_segmented_maxpool_fwd_dev(ptr(values), ptr(begin), ptr(end), i64(P), *rows.args(), i32(C), ptr(pooled),
                                                     ptr(arg), _stream()), "gpn_segmented_maxpool_fwd_dev")
        return pooled, arg
    check(_C.lib().gpn_segmented_maxpool_fwd(ptr(values), ptr(begin), ptr(end), i64(P), i32(C), ptr(pooled), ptr(arg),
                                             _stream()), "gpn_segmented_maxpool_fwd")
    return pooled, arg


def segmented_maxpool_bwd(dpooled, argmax, M, rows: Optional[DevCount] = None, m_rows: Optional[DevCount] = None):
    dev = _dev(dpooled, argmax)
    dpooled, argmax = _c(dpooled, torch.float32), _c(argmax, torch.int32)
    P, C = dpooled.shape
    dv = torch.empty((M, C), dtype=torch.float32, device=dev)
    if rows is not None:
        check(_C.lib().gpn_segmented_maxpool_bwd_dev(ptr(dpooled), ptr(argmax), i64(P), *rows.args(), i32(C), i64(M), *m_rows.args(),
                                                     ptr(dv), _stream()), "gpn_segmented_maxpool_bwd_dev")
        return dv
    check(_C.lib().gpn_segmented_maxpool_bwd(ptr(dpooled), ptr(argmax), i64(P), i32(C), i64(M), ptr(dv), _stream()),
          "gpn_segmented_maxpool_bwd")
    return dv


def instance_iou(proposal_offsets, instance_labels, batch_indices, num_points_per_instance, rows: Optional[DevCount] = None):
    dev = _dev(proposal_offsets, instance_labels, batch_indices, num_points_per_instance)
    po, il = _c(proposal_offsets, torch.int32), _c(instance_labels, torch.int32)
    bi, npi = _c(batch_indices, torch.int32), _c(num_points_per_instance, torch.int32)
    P, (B, I) = po.shape[0] - 1, npi.shape
    if rows is not None:  # the proposal count is a device counter (P = its bound; rows past it stay unwritten)
        out = torch.empty((P, I), dtype=torch.float32, device=dev)
        check(_C.lib().gpn_instance_iou_dev(ptr(po), ptr(il), ptr(bi), ptr(npi), i64(P), *rows.args(), i64(B), i32(I), ptr(out),
                                            _stream()), "gpn_instance_iou_dev")
        return out
    out = torch.zeros((P, I), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_instance_iou(ptr(po), ptr(il), ptr(bi), ptr(npi), i64(P), i64(B), i32(I), ptr(out), _stream()),
          "gpn_instance_iou")
    return out


def nms(ious, scores, threshold):
    dev = _dev(ious, scores)
    ious, scores = _c(ious, torch.float32), _c(scores, torch.float32)
    P = scores.shape[0]
    order = torch.sort(scores, descending=True, stable=True)[1].to(torch.int32)
    keep = torch.empty((max(P, 1),), dtype=torch.int32, device=dev)
    nk = torch.zeros((1,), dtype=torch.int32, device=dev)
    L = _C.lib()
    ws = _ws(L.gpn_nms_ws_bytes(i64(P)), dev)
    check(L.gpn_nms(ptr(ious), ptr(order), i64(P), f32(threshold), ptr(keep), ptr(nk), ptr(ws), szt(ws.numel()),
                    _stream()), "gpn_nms")
    return keep[: int(nk.item())].to(torch.int64)


# ---------------------------------------------------------------------------------------------------- F
def pn2_ball_query(radius, nsample, xyz, new_xyz):
    dev = _dev(xyz, new_xyz)
    xyz, new_xyz = _c(xyz, torch.float32), _c(new_xyz, torch.float32)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=dev)
    check(_C.lib().gpn_pn2_ball_query(i32(b), i32(n), i32(m), f32(radius), i32(nsample), ptr(new_xyz), ptr(xyz),
                                      ptr(idx), _stream()), "gpn_pn2_ball_query")
    return idx


def pn2_group_points(points, idx):
    dev = _dev(points, idx)
    points, idx = _c(points, torch.float32), _c(idx, torch.int32)
    b, c, n = points.shape
    _, npts, ns = idx.shape
    out = torch.empty((b, c, npts, ns), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_pn2_group_points(i32(b), i32(c), i32(n), i32(npts), i32(ns), ptr(points), ptr(idx), ptr(out),
                                        _stream()), "gpn_pn2_group_points")
    return out


def pn2_group_points_grad(grad_out, idx, n):
    dev = _dev(grad_out, idx)
    grad_out, idx = _c(grad_out, torch.float32), _c(idx, torch.int32)
    b, c, npts, ns = grad_out.shape
    gp = torch.zeros((b, c, n), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_pn2_group_points_grad(i32(b), i32(c), i32(n), i32(npts), i32(ns), ptr(grad_out), ptr(idx),
                                             ptr(gp), _stream()), "gpn_pn2_group_points_grad")
    return gp


def pn2_gather_points(points, idx):
    dev = _dev(points, idx)
    points, idx = _c(points, torch.float32), _c(idx, torch.int32)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_pn2_gather_points(i32(b), i32(c), i32(n), i32(m), ptr(points), ptr(idx), ptr(out), _stream()),
          "gpn_pn2_gather_points")
    return out


def pn2_gather_points_grad(grad_out, idx, n):
    dev = _dev(grad_out, idx)
    grad_out, idx = _c(grad_out, torch.float32), _c(idx, torch.int32)
    b, c, m = grad_out.shape
    gp = torch.zeros((b, c, n), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_pn2_gather_points_grad(i32(b), i32(c), i32(n), i32(m), ptr(grad_out), ptr(idx), ptr(gp),
                                              _stream()), "gpn_pn2_gather_points_grad")
    return gp


def pn2_furthest_point_sampling(xyz, npoint):
    dev = _dev(xyz)
    xyz = _c(xyz, torch.float32)
    b, n, _ = xyz.shape
    temp = torch.full((b, n), 1e10, dtype=torch.float32, device=dev)
    idx = torch.zeros((b, npoint), dtype=torch.int32, device=dev)
    L = _C.lib()
    ws = _ws(L.gpn_pn2_furthest_point_sampling_ws_bytes(i32(b), i32(n)), dev)  # non-empty only for clouds of >= 65536 points
    check(L.gpn_pn2_furthest_point_sampling_ws(i32(b), i32(n), i32(npoint), ptr(xyz), ptr(temp), ptr(idx), ptr(ws),
                                               szt(ws.numel()), _stream()), "gpn_pn2_furthest_point_sampling_ws")
    return idx


def pn2_three_nn(unknown, known):
    dev = _dev(unknown, known)
    unknown, known = _c(unknown, torch.float32), _c(known, torch.float32)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
    check(_C.lib().gpn_pn2_three_nn(i32(b), i32(n), i32(m), ptr(unknown), ptr(known), ptr(d2), ptr(idx), _stream()),
          "gpn_pn2_three_nn")
    return d2, idx


def pn2_knn(unknown, known, k):
    dev = _dev(unknown, known)
    unknown, known = _c(unknown, torch.float32), _c(known, torch.float32)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((b, n, k), dtype=torch.float32, device=dev)
    idx = torch.empty((b, n, k), dtype=torch.int32, device=dev)
    check(_C.lib().gpn_pn2_knn(i32(b), i32(n), i32(m), i32(k), ptr(unknown), ptr(known), ptr(d2), ptr(idx), _stream()),
          "gpn_pn2_knn")
    return d2, idx


def pn2_three_interpolate(points, idx, weight):
    dev = _dev(points, idx, weight)
    points, idx, weight = _c(points, torch.float32), _c(idx, torch.int32), _c(weight, torch.float32)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_pn2_three_interpolate(i32(b), i32(c), i32(m), i32(n), ptr(points), ptr(idx), ptr(weight),
                                             ptr(out), _stream()), "gpn_pn2_three_interpolate")
    return out


def pn2_three_interpolate_grad(grad_out, idx, weight, m):
    dev = _dev(grad_out, idx, weight)
    grad_out, idx, weight = _c(grad_out, torch.float32), _c(idx, torch.int32), _c(weight, torch.float32)
    b, c, n = grad_out.shape
    gp = torch.zeros((b, c, m), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_pn2_three_interpolate_grad(i32(b), i32(c), i32(n), i32(m), ptr(grad_out), ptr(idx),
                                                  ptr(weight), ptr(gp), _stream()), "gpn_pn2_three_interpolate_grad")
    return gp


# ---------------------------------------------------------------------------------------------------- PR
def proposals_build(points_xyz, offset_preds, sem_preds, instance_labels, batch_indices, batch_size, radius, K1, K2,
                    min_points, fullscale, max_scale, jitter, read_counts: bool = True):
    """Section PR of include/gpn.h: the whole proposal stage (model.py:228-346 of the reference) in one library call and ONE
    host read.  ``points_xyz`` = [N,3] view of the [N,6] point matrix (row stride passed through).  -> dict of tensors
    sliced to their actual sizes, or None when no proposal survives."""
    dev = _dev(points_xyz, offset_preds, sem_preds, batch_indices)
    N = int(points_xyz.shape[0])
    assert points_xyz.dtype == torch.float32 and points_xyz.stride(1) == 1 and sem_preds.dtype == torch.int64
    stride = int(points_xyz.stride(0))
    offset_preds = _c(offset_preds.detach(), torch.float32)
    sem_preds = sem_preds.contiguous()
    batch_indices = _c(batch_indices, torch.int32)
    inst = _c(instance_labels, torch.int32) if instance_labels is not None else None
    jit = torch.cat([jitter[0].reshape(3), jitter[1].reshape(3)]).to(device=dev, dtype=torch.float32).contiguous()
    L = _C.lib()
    L.gpn_proposals_max_proposals.restype = _C.ctypes.c_int64
    P_ub = int(L.gpn_proposals_max_proposals(i64(N), i32(min_points)))
    T2 = 2 * N

    def new(shape, dtype):
        return torch.empty(shape, dtype=dtype, device=dev)
    counts = new((8,), torch.int64)
    valid_mask = new((N,), torch.bool)
    valid_indices = new((N,), torch.int64)
    sorted_indices, point_indices, proposal_indices = new((T2,), torch.int64), new((T2,), torch.int64), new((T2,), torch.int64)
    batch_p, sem_p, inst_p = new((T2,), torch.int32), new((T2,), torch.int32), new((T2,), torch.int32)
    xyz_p = new((T2, 3), torch.float32)
    sizes, offsets = new((P_ub + 1,), torch.int64), new((P_ub + 1,), torch.int32)
    member_slot = new((T2,), torch.int32)
    coords4 = new((T2, 4), torch.int32)
    pid, order, vstart = new((T2,), torch.int32), new((T2,), torch.int32), new((T2 + 1,), torch.int32)
    ws = _ws(L.gpn_proposals_build_ws_bytes(i64(N), i64(batch_size), i32(K1), i32(K2), i32(min_points)), dev)
    check(L.gpn_proposals_build(ptr(points_xyz), i32(stride), ptr(offset_preds), ptr(sem_preds), ptr(inst), ptr(batch_indices),
                                i64(N), i64(batch_size), f32(radius), i32(K1), i32(K2), i32(min_points), f32(fullscale),
                                f32(max_scale), ptr(jit), ptr(counts), ptr(valid_mask), ptr(valid_indices), ptr(sorted_indices),
                                ptr(point_indices), ptr(proposal_indices), ptr(batch_p), ptr(xyz_p), ptr(sem_p), ptr(inst_p),
                                ptr(sizes), ptr(offsets), ptr(member_slot), ptr(coords4), ptr(pid), ptr(order), ptr(vstart),
                                ptr(ws), szt(ws.numel()), _stream()), "gpn_proposals_build")
    if not read_counts:
        # no host read: every output at its bound, the counts stay on the device (DevCount views of `counts`; section DEV of
        # include/gpn.h).  P_ub + 1 offsets, T2 = 2 N per-point rows, T2 voxel rows.
        return dict(counts=counts, Q_dev=counts[0:1], M_dev=counts[1:2], P_dev=counts[2:3], V_dev=counts[3:4],
                    coarse_dev=counts[6:7], P=P_ub, M=T2, V=T2, valid_mask=valid_mask, valid_indices=valid_indices,
                    sorted_indices=sorted_indices, point_indices=point_indices, proposal_indices=proposal_indices,
                    batch_indices=batch_p, pt_xyz=xyz_p, sem_preds=sem_p, instance_labels=inst_p if instance_labels is not None else None,
                    sizes=sizes[:P_ub], proposal_offsets=offsets, member_slot=member_slot, voxel_coords=coords4, pc_voxel_id=pid,
                    point_order=order, voxel_point_start=vstart)
    Q, M, P, V, dropped, _, coarse = counts.tolist()[:7]  # the stage's single device -> host read
    if M == 0:
        return None
    return dict(counts_host=(Q, M, P, V, dropped, coarse), Q=Q, M=M, P=P, V=V, dropped=dropped, coarse=coarse, valid_mask=valid_mask, valid_indices=valid_indices[:Q],
                sorted_indices=sorted_indices[:M], point_indices=point_indices[:M], proposal_indices=proposal_indices[:M],
                batch_indices=batch_p[:M], pt_xyz=xyz_p[:M], sem_preds=sem_p[:M],
                instance_labels=inst_p[:M] if instance_labels is not None else None, sizes=sizes[:P],
                proposal_offsets=offsets[:P + 1], member_slot=member_slot, voxel_coords=coords4[:V], pc_voxel_id=pid[:M],
                point_order=order[:M], voxel_point_start=vstart[:V + 1])


def proposals_postprocess(score_preds, sizes, proposal_offsets, point_indices, proposal_indices, member_slot, score_threshold,
                          min_points, iou_threshold, rows: Optional[DevCount] = None, defer: bool = False):
    """Section PP of include/gpn.h: score filter + NMS + compaction tables of a validation step's proposals in one call and ONE
    host read.  -> (kept_ids [P''] i32 ascending, new_offsets [P''+1] i32, src_row [M''] i64), or None when a kernel table
    overflowed (the caller falls back to the torch formulation)."""
    dev = _dev(score_preds, sizes)
    P = int(score_preds.shape[0])
    M = int(point_indices.shape[0])
    N = int(member_slot.shape[0]) // 2
    L = _C.lib()
    kept_ids = torch.empty((max(P, 1),), dtype=torch.int32, device=dev)
    new_offsets = torch.empty((P + 1,), dtype=torch.int32, device=dev)
    src_row = torch.empty((max(M, 1),), dtype=torch.int64, device=dev)
    counts = torch.empty((3,), dtype=torch.int64, device=dev)
    ws = _ws(L.gpn_proposals_postprocess_ws_bytes(i64(P)), dev)
    p_dev, p_plan = (rows.ptr, i64(rows.plan)) if rows is not None else (ptr(None), i64(0))
    check(L.gpn_proposals_postprocess(ptr(_c(score_preds, torch.float32)), ptr(_c(sizes, torch.int64)), ptr(_c(proposal_offsets, torch.int32)),
                                      ptr(_c(point_indices, torch.int64)), ptr(_c(proposal_indices, torch.int64)),
                                      ptr(_c(member_slot, torch.int32)), i64(N), i64(P), p_dev, p_plan, f32(score_threshold),
                                      i64(min_points), f32(iou_threshold), ptr(kept_ids), ptr(new_offsets), ptr(src_row), ptr(counts),
                                      ptr(ws), szt(ws.numel()), _stream()), "gpn_proposals_postprocess")
    if defer:  # the read is the caller's, later (PostprocessHandle.result): nothing waits for this step's kernels now
        host = _PINNED_COUNTS.pop() if _PINNED_COUNTS else torch.empty((3,), dtype=torch.int64).pin_memory()
        host.copy_(counts, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return PostprocessHandle(kept_ids, new_offsets, src_row, host, ev)
    n_kept, n_points, overflow = counts.tolist()  # the step's one read of its post-processing
    if overflow:
        return None
    return kept_ids[:n_kept], new_offsets[:n_kept + 1], src_row[:n_points]


def scene_prepare(points, sem_labels, instance_labels, seg_offsets, mats=None, shifts=None):
    """Section SP of include/gpn.h: label compaction, augmentation and the per-instance statistics of a batch of RAW scenes in three
    launches and ONE host read (the instances per scene).  points [N, 3 + C] f32, sem_labels [N] int16 / int32 / int64,
    instance_labels [N] i32, seg_offsets [B + 1] i64 (device), mats [B, 3, 3] / shifts [B, C] f64 on the device or None.
    -> dict(points, batch_indices, instance_labels, instance_regions, num_points_per_instance, instance_sem_labels [B, max K],
    num_instances = [K per scene]) or None when a scene has more distinct ids than the kernel tables hold."""
    dev = _dev(points, instance_labels)
    points, instance_labels = _c(points, torch.float32), _c(instance_labels, torch.int32)
    sem_labels = sem_labels.contiguous()
    sem_bytes = sem_labels.element_size()
    assert sem_labels.dtype in (torch.int16, torch.int32, torch.int64), sem_labels.dtype
    N, W = points.shape
    B = int(seg_offsets.shape[0]) - 1
    L = _C.lib()
    cap = L.gpn_scene_prepare_max_instances()
    out_points = torch.empty_like(points)
    batch_indices = torch.empty((N,), dtype=torch.int32, device=dev)
    ins_out = torch.empty((N,), dtype=torch.int32, device=dev)
    regions = torch.empty((N, 9), dtype=torch.float32, device=dev)
    npi = torch.empty((B, cap), dtype=torch.int32, device=dev)
    isl = torch.empty((B, cap), dtype=torch.int32, device=dev)
    overflow = torch.empty((1,), dtype=torch.int32, device=dev)
    k_host = _PINNED_K.pop() if _PINNED_K and _PINNED_K[-1].numel() >= B else torch.empty((max(B, 64),), dtype=torch.int64).pin_memory()
    ws = _ws(L.gpn_scene_prepare_ws_bytes(i32(B)), dev)
    check(L.gpn_scene_prepare(ptr(points), ptr(sem_labels), i32(sem_bytes), ptr(instance_labels), ptr(_c(seg_offsets, torch.int64)),
                              i64(N), i32(W - 3), i32(B), ptr(_c(mats, torch.float64)), ptr(_c(shifts, torch.float64)),
                              ptr(out_points), ptr(batch_indices), ptr(ins_out), ptr(regions), ptr(npi), ptr(isl), ptr(k_host),
                              ptr(overflow), ptr(ws), szt(ws.numel()), _stream()), "gpn_scene_prepare")
    ev = torch.cuda.Event(blocking=True)
    ev.record()
    ev.synchronize()  # the preparation's one host read: K per scene, written into pinned memory by the first kernel
    k = k_host[:B].tolist()
    _PINNED_K.append(k_host)
    if min(k) < 0:
        return None
    width = max(k)
    return dict(points=out_points, batch_indices=batch_indices, instance_labels=ins_out, instance_regions=regions,
                num_points_per_instance=npi[:, :width].contiguous(), instance_sem_labels=isl[:, :width].contiguous(),
                num_instances=k)


_PINNED_K = []
_PINNED_COUNTS = []


class PostprocessHandle:
    """outputs of a gpn_proposals_postprocess call whose counts have not been read yet; ``result()`` waits for them (no wait if
    the call's kernels have run) -> (kept_ids, new_offsets, src_row) sliced to size, or None after a table overflow"""

    def __init__(self, kept_ids, new_offsets, src_row, host, event):
        self.kept_ids, self.new_offsets, self.src_row, self.host, self.event = kept_ids, new_offsets, src_row, host, event

    def result(self):
        self.event.synchronize()
        n_kept, n_points, overflow = self.host.tolist()
        _PINNED_COUNTS.append(self.host)
        if overflow:
            return None
        return self.kept_ids[:n_kept], self.new_offsets[:n_kept + 1], self.src_row[:n_points]


def proposals_targets(sem_labels, gt_npcs, point_indices, rows: DevCount):
    """sem_labels [N] i64 / gt_npcs [N,3] f32 (either may be None) at the proposal points -> ([M] i64 or None, [M,3] f32 or
    None), M = point_indices.shape[0] = the bound; rows past the device count stay unwritten"""
    dev = _dev(point_indices)
    M = point_indices.shape[0]
    sem_out = torch.empty((M,), dtype=torch.int64, device=dev) if sem_labels is not None else None
    npcs_out = torch.empty((M, 3), dtype=torch.float32, device=dev) if gt_npcs is not None else None
    check(_C.lib().gpn_proposals_targets_dev(ptr(_c(sem_labels, torch.int64)), ptr(_c(gt_npcs, torch.float32)), ptr(point_indices),
                                             i64(M), *rows.args(), ptr(sem_out), ptr(npcs_out), _stream()), "gpn_proposals_targets_dev")
    return sem_out, npcs_out


def proposals_voxel_mean(feats, point_indices, point_order, voxel_point_start, V, rows: Optional[DevCount] = None):
    dev = _dev(feats)
    feats = _c(feats, torch.float32)
    out = torch.empty((V, feats.shape[1]), dtype=torch.float32, device=dev)
    if rows is not None:
        check(_C.lib().gpn_proposals_voxel_mean_dev(ptr(feats), ptr(point_indices), ptr(point_order), ptr(voxel_point_start), i64(V),
                                                    *rows.args(), i32(feats.shape[1]), ptr(out), _stream()), "gpn_proposals_voxel_mean_dev")
        return out
    check(_C.lib().gpn_proposals_voxel_mean(ptr(feats), ptr(point_indices), ptr(point_order), ptr(voxel_point_start), i64(V),
                                            i32(feats.shape[1]), ptr(out), _stream()), "gpn_proposals_voxel_mean")
    return out


def proposals_voxel_mean_bwd(dout, member_slot, pc_voxel_id, voxel_point_start, N):
    dev = _dev(dout)
    dout = _c(dout, torch.float32)
    dfeats = torch.empty((N, dout.shape[1]), dtype=torch.float32, device=dev)
    check(_C.lib().gpn_proposals_voxel_mean_bwd(ptr(dout), ptr(member_slot), ptr(pc_voxel_id), ptr(voxel_point_start), i64(N),
                                                i32(dout.shape[1]), ptr(dfeats), _stream()), "gpn_proposals_voxel_mean_bwd")
    return dfeats


# ---------------------------------------------------------------------------------------------------- NPCS loss
def _npcs_args(sym):
    i32a = lambda v: (_C.ctypes.c_int32 * len(v))(*v)  # noqa: E731
    return i32a(sym["first"]), i32a(sym["count"]), i32a(sym["group"]), i32(len(sym["first"]))


def npcs_loss_fwd(logits, gt_npcs, sem_preds, sem_labels, proposal_offsets, sym, rows: Optional[DevCount] = None):
    """-> (loss [1] f32, scratch for the backward).  ``sym`` = dict(sym_of_class i64 [classes], mats f32 [n,3,3] (device),
    first / count / group: python lists per symmetry type)."""
    dev = _dev(logits, gt_npcs)
    logits, gt_npcs = _c(logits, torch.float32), _c(gt_npcs, torch.float32)
    sem_preds, sem_labels = _c(sem_preds, torch.int32), _c(sem_labels, torch.int64)
    proposal_offsets = _c(proposal_offsets, torch.int32)
    P = proposal_offsets.shape[0] - 1
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    scratch = torch.empty((9 * P + 4,), dtype=torch.float32, device=dev)
    first, count, group, n = _npcs_args(sym)
    if rows is not None:  # the proposal count is a device counter (P = its bound)
        check(_C.lib().gpn_npcs_loss_fwd_dev(ptr(logits), i32(logits.shape[1]), ptr(gt_npcs), ptr(sem_preds), ptr(sem_labels),
                                             ptr(proposal_offsets), i64(P), *rows.args(), ptr(sym["sym_of_class"]), ptr(sym["mats"]),
                                             first, count, group, n, ptr(loss), ptr(scratch), _stream()), "gpn_npcs_loss_fwd_dev")
        return loss, scratch
    check(_C.lib().gpn_npcs_loss_fwd(ptr(logits), i32(logits.shape[1]), ptr(gt_npcs), ptr(sem_preds), ptr(sem_labels),
                                     ptr(proposal_offsets), i64(P), ptr(sym["sym_of_class"]), ptr(sym["mats"]), first, count,
                                     group, n, ptr(loss), ptr(scratch), _stream()), "gpn_npcs_loss_fwd")
    return loss, scratch


def npcs_loss_bwd(logits, gt_npcs, sem_preds, sem_labels, proposal_indices, P, sym, scratch, grad_loss,
                  m_rows: Optional[DevCount] = None):
    dev = _dev(logits)
    d_logits = torch.empty_like(logits)
    first, count, group, n = _npcs_args(sym)
    if m_rows is not None:  # the point count is a device counter (rows of d_logits past it stay unwritten)
        check(_C.lib().gpn_npcs_loss_bwd_dev(ptr(logits), i32(logits.shape[1]), ptr(gt_npcs), ptr(sem_preds), ptr(sem_labels),
                                             ptr(_c(proposal_indices, torch.int64)), i64(logits.shape[0]), *m_rows.args(), i64(P),
                                             ptr(sym["sym_of_class"]), ptr(sym["mats"]), first, count, group, n, ptr(scratch),
                                             ptr(_c(grad_loss.reshape(1), torch.float32)), ptr(d_logits), _stream()), "gpn_npcs_loss_bwd_dev")
        return d_logits
    check(_C.lib().gpn_npcs_loss_bwd(ptr(logits), i32(logits.shape[1]), ptr(gt_npcs), ptr(sem_preds), ptr(sem_labels),
                                     ptr(_c(proposal_indices, torch.int64)), i64(logits.shape[0]), i64(P), ptr(sym["sym_of_class"]),
                                     ptr(sym["mats"]), first, count, group, n, ptr(scratch),
                                     ptr(_c(grad_loss.reshape(1), torch.float32)), ptr(d_logits), _stream()), "gpn_npcs_loss_bwd")
    return d_logits


# ---------------------------------------------------------------------------------------------------- H (dense heads)
def linear_supported(cin: int, cout: int) -> bool:
    return bool(_C.lib().gpn_linear_supported(i32(cin), i32(cout)))


def linear_fwd(x, weight, bias, rows: Optional[DevCount] = None):
    """y = x @ weight.T + bias in one launch (csrc/linear.hip); weight [cout, cin] as torch.nn.Linear holds it"""
    dev = _dev(x, weight)
    x, weight = _c(x, torch.float32), _c(weight, torch.float32)
    N, cin = x.shape
    cout = weight.shape[0]
    y = torch.empty((N, cout), dtype=torch.float32, device=dev)
    if rows is not None:  # the row count is a device counter (N = its bound)
        check(_C.lib().gpn_linear_fwd_dev(ptr(x), ptr(weight), ptr(_c(bias, torch.float32) if bias is not None else None), i64(N),
                                          *rows.args(), i32(cin), i32(cout), ptr(y), _stream()), "gpn_linear_fwd_dev")
        return y
    check(_C.lib().gpn_linear_fwd(ptr(x), ptr(weight), ptr(_c(bias, torch.float32) if bias is not None else None), i64(N),
                                  i32(cin), i32(cout), ptr(y), _stream()), "gpn_linear_fwd")
    return y


def linear_bwd(x, weight, dy, need_dx: bool, need_dw: bool, need_db: bool, rows: Optional[DevCount] = None):
    """-> (dx or None, dW or None, db or None): three launches at most (dx; partial dW / db per 512 rows; their ordered sum)"""
    dev = _dev(x, dy)
    x, weight, dy = _c(x, torch.float32), _c(weight, torch.float32), _c(dy, torch.float32)
    N, cin = x.shape
    cout = weight.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(weight) if need_dw else None
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if need_db else None
    L = _C.lib()
    ws = _ws(L.gpn_linear_bwd_ws_bytes(i64(N), i32(cin), i32(cout)), dev) if (need_dw or need_db) else None
    if rows is not None:
        check(L.gpn_linear_bwd_dev(ptr(x), ptr(weight), ptr(dy), i64(N), *rows.args(), i32(cin), i32(cout), ptr(dx), ptr(dw), ptr(db),
                                   ptr(ws), szt(ws.numel() if ws is not None else 0), _stream()), "gpn_linear_bwd_dev")
        return dx, dw, db
    check(L.gpn_linear_bwd(ptr(x), ptr(weight), ptr(dy), i64(N), i32(cin), i32(cout), ptr(dx), ptr(dw), ptr(db), ptr(ws),
                           szt(ws.numel() if ws is not None else 0), _stream()), "gpn_linear_bwd")
    return dx, dw, db


def score_loss(logits, cls_source, proposal_offsets, ious, fg_thresh, bg_thresh, rows: Optional[DevCount] = None):
    """-> (loss [1] f32, score_preds [P] f32, d_logits [P, C1] f32) in one launch (gpn_score_loss)"""
    dev = _dev(logits, ious)
    logits, ious = _c(logits, torch.float32), _c(ious, torch.float32)
    po = _c(proposal_offsets, torch.int32)
    cls_source = cls_source.contiguous()
    assert cls_source.dtype in (torch.int64, torch.int32)
    P, C1 = logits.shape
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    preds = torch.empty((P,), dtype=torch.float32, device=dev)
    d_logits = torch.empty_like(logits)
    is64 = cls_source.dtype == torch.int64
    if rows is not None:  # the proposal count is a device counter (P = its bound; rows past it stay unwritten and unread)
        check(_C.lib().gpn_score_loss_dev(ptr(logits), i32(C1), ptr(cls_source if is64 else None), ptr(None if is64 else cls_source),
                                          ptr(po), ptr(ious), i32(ious.shape[1]), i64(P), rows.ptr, f32(fg_thresh), f32(bg_thresh),
                                          ptr(loss), ptr(preds), ptr(d_logits), _stream()), "gpn_score_loss_dev")
        return loss, preds, d_logits
    check(_C.lib().gpn_score_loss(ptr(logits), i32(C1), ptr(cls_source if is64 else None), ptr(None if is64 else cls_source),
                                  ptr(po), ptr(ious), i32(ious.shape[1]), i64(P), f32(fg_thresh), f32(bg_thresh), ptr(loss),
                                  ptr(preds), ptr(d_logits), _stream()), "gpn_score_loss")
    return loss, preds, d_logits
