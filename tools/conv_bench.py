"""micro-benchmark of the fused sparse conv on synthetic surface voxels shaped like the U-Net levels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gapartnet_amd import hip_ops as H
from tests import synth
dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


cases = [((256, 256, 256), 19000, 16, 16), ((128, 128, 128), 10500, 32, 32), ((128, 128, 128), 10500, 64, 32),
         ((64, 64, 64), 3400, 48, 48), ((32, 32, 32), 900, 64, 64)]
rng = np.random.default_rng(0)
for shape, per_scene, cin, cout in cases:
    idx = torch.from_numpy(synth.surface_indices(rng, 8, list(shape), per_scene)).to(dev)
    n = idx.shape[0]
    rb = H.rulebook_subm3(idx, list(shape))
    pairs = int(rb.num_pairs.item())
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    t = timeit(lambda: H.conv_fwd(x, w, rb))
    g = torch.randn(n, cout, device=dev)
    tw = timeit(lambda: H.conv_wgrad(x, g, rb))
    flops = 2.0 * pairs * cin * cout
    dense = 2.0 * n * 27 * cin * cout
    print(f"rows {n:7d} pairs/row {pairs / n:5.2f} {cin:3d}->{cout:3d}: fwd {t:7.1f} us {flops / t / 1e6:6.2f} TF useful {dense / t / 1e6:6.2f} TF incl. empty | wgrad {tw:7.1f} us {flops / tw / 1e6:6.2f} TF")
