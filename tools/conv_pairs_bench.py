"""The pair-compacted conv kernel (csrc/spconv_pairs.hip) against the output-stationary kernels (masked-tile / direct / split) at
the bench's level shapes (8 x 20k-point scenes, voxel 0.01): us per launch through the C-ABI (events around back-to-back
launches, packed weights and rulebooks prebuilt), useful TFLOP/s against the fp32 MFMA peak, and bit-equality of the outputs.

  python tools/conv_pairs_bench.py            # rows per wave 32 / 64 / 128 x column tiles per wave 1 .. all"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import _C, hip_ops as H
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud

dev = torch.device("cuda:0")
MFMA_PEAK = 157.3
L = _C.lib()


def timeit(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def ref_call(x, packed, rb, cin, cout, out):
    ws_ptr, ws_size, stream = H._fast_ws(dev)
    rc = L.gpn_spconv_fwd_ordered(H.ptr(x), H.ptr(packed), H.ptr(rb.nbr), H.ptr(rb.nbr_p), H.ptr(rb.perm), H.i32(rb.K),
                                  H.i64(rb.n_dst), H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(ws_ptr),
                                  ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def pairs_call(x, packed, rb, cin, cout, out):
    _, _, stream = H._fast_ws(dev)
    rc = L.gpn_spconv_fwd_pairs(H.ptr(x), H.ptr(packed), H.ptr(rb.pair_src), H.ptr(rb.pair_dst), H.ptr(rb.tile_off), H.i32(rb.K),
                                H.i64(rb.n_dst), H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def main():
    torch.manual_seed(0)
    n_levels = int(os.environ.get("BENCH_LEVELS", 5))
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    levels = []
    for lvl in range(n_levels):
        rb = H.rulebook_subm3(idx, shape)
        idx2, shape2, rbd, rbu = H.rulebook_down(idx, shape, 8)
        levels.append((rb, rbd, rbu))
        idx, shape = idx2, shape2
    print(f"{'shape':>28s} {'reference':>10s} | pairs kernel, us per launch at (rows per wave, column tiles per wave)")
    for lvl, (rb, rbd, rbu) in enumerate(levels):
        c = 16 * (lvl + 1)
        cases = [("subm", rb, c, c), ("subm", rb, 2 * c, c), ("down", rbd, c, c + 16), ("inv", rbu, c + 16, c)]
        for kind, r, cin, cout in cases:
            if cin > 128:
                continue
            K, pairs = r.K, int(r.num_pairs.item())
            x = torch.randn(r.n_src, cin, device=dev)
            w = torch.randn(K, cin, cout, device=dev) / (K * cin) ** 0.5
            packed = H.pack_weights(w, 0)
            ref = torch.empty(r.n_dst, cout, device=dev)
            ref_call(x, packed, r, cin, cout, ref)
            us_ref = timeit(lambda: ref_call(x, packed, r, cin, cout, ref))
            flops = 2.0 * pairs * cin * cout
            line = f"L{lvl} {r.n_dst:7d} {kind:>4s} {cin:3d}->{cout:<3d} {pairs / r.n_dst:5.1f}p/r {us_ref:7.1f} ({flops / us_ref / 1e6 / MFMA_PEAK:.3f}) |"
            nt_total = cout // 16
            best = None
            for rw in (32, 64, 128):
                for nt in sorted({1, min(2, nt_total), min(4, nt_total)}):
                    if nt_total % nt:
                        continue
                    L.gpn_spconv_pairs_config(1, rw, nt, ctypes.c_int64(0), ctypes.c_int64(1 << 40))
                    out = torch.full((r.n_dst, cout), float("nan"), device=dev)
                    pairs_call(x, packed, r, cin, cout, out)
                    same = torch.equal(out, ref)
                    us = timeit(lambda: pairs_call(x, packed, r, cin, cout, out))
                    line += f" ({rw},{nt}) {us:6.1f}{'' if same else ' DIFF %.1e' % float((out - ref).abs().max())}"
                    if best is None or us < best[0]:
                        best = (us, rw, nt)
            line += f" | best {best[0]:.1f} us = {flops / best[0] / 1e6 / MFMA_PEAK:.3f} of MFMA at {best[1:]}: x{us_ref / best[0]:.2f}"
            print(line, flush=True)
    L.gpn_spconv_pairs_config(0, 64, 0, ctypes.c_int64(4096), ctypes.c_int64(1 << 30))


if __name__ == "__main__":
    main()
