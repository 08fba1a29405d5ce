import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from gapartnet_amd import functional as GF
dev = torch.device("cuda:0")
for n, cin, cout in ((160000, 16, 10), (160000, 16, 16), (160000, 16, 3), (60000, 16, 27)):
    x = torch.randn(n, cin, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, device=dev, requires_grad=True)
    b = torch.randn(cout, device=dev, requires_grad=True)
    dy = torch.randn(n, cout, device=dev)
    for name, fn in (("F.linear", F.linear), ("GF.linear", GF.linear)):
        def run():
            y = fn(x, w, b)
            torch.autograd.grad(y, [x, w, b], dy)
        for _ in range(5): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): run()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{n:7d} {cin}->{cout:2d} {name:10s} host {1e6 * (t1 - t0) / 50:7.1f} us  total {1e6 * (t2 - t0) / 50:7.1f} us")
