"""micro-benchmark of the fused BatchNorm entry points over the (N, C) shapes of the U-Net levels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import hip_ops as H
dev = torch.device("cuda:0")


def timeit(fn, iters=int(os.environ.get('ITERS', 200))):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print(f"{'N':>8s} {'C':>4s} {'fwd us':>8s} {'bwd us':>8s}")
for C in (16, 48, 112):
    for N in (40, 600, 1024, 1025):
        x = torch.randn(N, C, device=dev)
        r = torch.randn(N, C, device=dev)
        w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, mean, invstd = H.bn_fwd(x, r, w, b, rm, rv, True, 0.1, 1e-4, True)
        dy = torch.randn(N, C, device=dev)
        f = timeit(lambda: H.bn_fwd(x, r, w, b, rm, rv, True, 0.1, 1e-4, True))
        g = timeit(lambda: H.bn_bwd(x, y, dy, w, mean, invstd, True, True, True))
        print(f"{N:8d} {C:4d} {f:8.1f} {g:8.1f}")
