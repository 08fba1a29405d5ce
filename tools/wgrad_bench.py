"""The weight-gradient contraction at the bench's level shapes (8 x 20k-point scenes, voxel 0.01): time per layer through
gpn_spconv_wgrad (contraction + slice sums), useful TFLOP/s (2 P Cin Cout) against the fp32 MFMA peak, for both kernels -
gathered rows straight into MFMA operands (csrc/spconv_wgrad.hip) and the LDS-staged one (csrc/spconv.hip) - and the
largest difference between their results.

  python tools/wgrad_bench.py        # BENCH_LEVELS=5 levels by default
"""
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


def timeit(fn, iters=30, warm=4):
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


def main():
    torch.manual_seed(0)
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    print(f"{'level rows':>14s} {'pairs/row':>9s} {'conv':>12s} | {'rows us':>8s} {'TF':>6s} {'/mfma':>6s} | {'LDS us':>8s} {'TF':>6s} {'/mfma':>6s} | max rel diff")
    for lvl in range(int(os.environ.get("BENCH_LEVELS", 5))):
        rb = H.rulebook_subm3(idx, shape)
        c = 16 * (lvl + 1)
        cases = [("subm", rb, c, c), ("subm", rb, 2 * c, c)]
        nxt = None
        if lvl < 4:
            idx2, shape2, rbd, rbu = H.rulebook_down(idx, shape, 8)
            cases += [("down", rbd, c, c + 16), ("inv", rbu, c + 16, c)]
            nxt = (idx2, shape2)
        for kind, r, cin, cout in cases:
            pairs = int(r.num_pairs.item())
            x = torch.randn(r.n_src, cin, device=dev)
            g = torch.randn(r.n_dst, cout, device=dev)
            res, us = {}, {}
            for mode in (2, 0):
                L.gpn_spconv_wgrad_rows(mode)
                res[mode] = H.conv_wgrad(x, g, r)
                us[mode] = timeit(lambda: H.conv_wgrad(x, g, r))
            L.gpn_spconv_wgrad_rows(0)
            flops = 2.0 * pairs * cin * cout
            d = ((res[0] - res[2]).abs().max() / res[0].abs().max()).item()
            tf = {m: flops / us[m] / 1e6 for m in us}
            print(f"L{lvl} {r.n_dst:10d} {pairs / r.n_dst:9.2f} {kind:>4s} {cin:3d}->{cout:<3d} | {us[2]:8.1f} {tf[2]:6.2f} {tf[2] / MFMA_PEAK:6.3f} | "
                  f"{us[0]:8.1f} {tf[0]:6.2f} {tf[0] / MFMA_PEAK:6.3f} | {d:.1e}")
        if nxt is None:
            break
        idx, shape = nxt


if __name__ == "__main__":
    main()
