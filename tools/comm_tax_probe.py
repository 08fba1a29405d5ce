"""What an RCCL communicator's existence changes for a single process: host cost of launches, small allocations, device->host
reads and library calls, with and without `init_process_group(..., device_id=...)`.  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if os.environ.get("PROBE_GROUP", "1") == "1":
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from gapartnet_amd import hip_ops as H

x = torch.randn(1 << 16, device=dev)
y = torch.empty_like(x)
side = torch.cuda.Stream()


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


def read_back():
    y.add_(1.0)
    return y[0].item()


def two_streams():
    y.add_(1.0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)


idx = torch.randint(0, 1000, (4096,), device=dev, dtype=torch.int32)
table = torch.randn(1000, 16, device=dev)
print(f"group={os.environ.get('PROBE_GROUP', '1')}")
for name, fn, n in (("torch add_ (launch)", lambda: y.add_(1.0), 3000), ("torch.empty 1 MB", lambda: torch.empty(1 << 18, device=dev), 3000),
                    ("library call gather_rows", lambda: H.gather_rows(table, idx), 3000), ("add_ + .item()", read_back, 500),
                    ("cross-stream wait pair", two_streams, 1000)):
    host, wall = per_call(fn, n)
    print(f"  {name:28s} host {host:7.2f} us/call   wall {wall:7.2f} us/call", flush=True)
