"""The direct conv kernel on the bench's small levels (L2 .. L5 of 8 x 20k-point scenes) unsplit, 2-way and 4-way tap-split:
us per launch through the C-ABI (events around back-to-back launches; packed weights, prebuilt rulebooks).
Output committed as profiles/r03_conv_split.txt."""
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
L = _C.lib()
BIG = 1 << 40


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


def conv_call(x, packed, rb, cin, cout, out):
    ws_ptr, ws_size, stream = H._fast_ws(dev)
    rc = L.gpn_spconv_fwd_ordered(H.ptr(x), H.ptr(packed), H.ptr(rb.nbr), H.ptr(rb.nbr_p), H.ptr(rb.perm), H.i32(rb.K),
                                  H.i64(rb.n_dst), H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(ws_ptr),
                                  ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def main():
    torch.manual_seed(0)
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    L.gpn_spconv_tiles_min_tiles(BIG)  # the direct kernel everywhere
    print(f"{'level rows':>14s} {'conv':>12s} {'units':>7s} {'unsplit':>8s} {'2-way':>8s} {'4-way':>8s}   (us per launch)")
    for lvl in range(7):
        rb = H.rulebook_subm3(idx, shape)
        c = 16 * (lvl + 1)
        if lvl >= 1 and rb.n_dst >= 256:
            for cin, cout in ((c, c), (2 * c, c), (c, 2 * c)):
                x = torch.randn(rb.n_src, cin, device=dev)
                w = torch.randn(27, cin, cout, device=dev) / (27 * cin) ** 0.5
                packed = H.pack_weights(w, 0)
                out = torch.empty(rb.n_dst, cout, device=dev)
                ts = []
                for s4, s2 in ((0, 0), (0, BIG), (BIG, 0)):
                    L.gpn_spconv_direct_split(s4, s2)
                    ts.append(timeit(lambda: conv_call(x, packed, rb, cin, cout, out)))
                units = (rb.n_dst + 15) // 16 * (cout // 16)
                print(f"L{lvl} {rb.n_dst:10d} {cin:4d}->{cout:<4d} {units:7d} {ts[0]:8.1f} {ts[1]:8.1f} {ts[2]:8.1f}")
        if lvl < 6:
            idx, shape, _, _ = H.rulebook_down(idx, shape, 8)
    L.gpn_spconv_direct_split(12000, 0)


if __name__ == "__main__":
    main()
