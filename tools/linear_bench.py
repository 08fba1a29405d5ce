"""micro-benchmark of the dense-head kernels (csrc/linear.hip) through the Python wrappers: forward, full backward, dW / db only"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gapartnet_amd import hip_ops as H
dev = torch.device("cuda:0")
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
for n, cin, cout in ((160000, 16, 10), (160000, 16, 16), (160000, 16, 3), (10000, 16, 27), (300, 16, 9), (4096, 64, 64)):
    x = torch.randn(n, cin, device=dev); w = torch.randn(cout, cin, device=dev); b = torch.randn(cout, device=dev); dy = torch.randn(n, cout, device=dev)
    print(n, cin, cout, "fwd %.1f us" % t(lambda: H.linear_fwd(x, w, b)), "bwd (dx+dW+db) %.1f us" % t(lambda: H.linear_bwd(x, w, dy, True, True, True)), "bwd dW only %.1f" % t(lambda: H.linear_bwd(x, w, dy, False, True, True)))
