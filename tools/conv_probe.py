"""us per launch of the k=3 submanifold conv forward on the bench's level shapes (8 scenes x 20k points): the table-driven
kernels of csrc/spconv_fwd.hip through gpn_spconv_fwd.  Used by tools/conv_ablation.sh.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import hip_ops as H
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
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


pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
levels = []
for _ in range(4):
    levels.append((idx, H.rulebook_subm3(idx, shape)))
    idx, shape, _, _ = H.rulebook_down(idx, shape, 8)
for lvl, cin, cout in ((0, 16, 16), (0, 32, 16), (0, 16, 32), (1, 32, 32), (1, 64, 32), (2, 48, 48), (3, 64, 64)):
    idx, rb = levels[lvl]
    n = idx.shape[0]
    x = torch.randn(n, cin, device=dev)
    packed = H.pack_weights(torch.randn(27, cin, cout, device=dev) * 0.05, 0)
    t = timeit(lambda: H._conv_packed(x, packed, rb, cin, cout))
    print(f"rows {n:7d}  {cin:2d}->{cout:2d}: {t:6.1f} us", flush=True)
