"""Host cost of copies / scalar reads in one training step, grouped by call site (torch profiler, python stacks):
which `aten::copy_` (H2D tables, D2H sizes) and `item()` calls the glue still pays for, and how long the host sits in each."""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]


def step(i):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(pool[i % 2], i)
    loss.backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
STEPS = 3
import time
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

WATCH = {"aten.copy_.default", "aten._to_copy.default", "aten._local_scalar_dense.default", "aten.nonzero.default",
         "aten.fill_.Scalar", "aten.cat.default", "aten.index.Tensor", "aten.zero_.default", "aten.lift_fresh.default",
         "aten.zeros.default", "aten.zeros_like.default", "aten.full.default", "aten.ones.default", "aten.ones_like.default",
         "aten.clone.default", "aten.new_zeros.default", "aten.fill_.Tensor", "aten.constant_pad_nd.default"}
sites = defaultdict(lambda: [0, 0.0])


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name not in WATCH:
            return func(*args, **(kwargs or {}))
        t0 = time.perf_counter()
        out = func(*args, **(kwargs or {}))
        dt = time.perf_counter() - t0
        frames = [f for f in traceback.extract_stack() if "/gapartnet_amd/" in f.filename]
        where = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in reversed(frames[-2:])) or "(outside)"
        kind = ""
        if name.startswith("aten.copy_") or name.startswith("aten._to_copy"):
            src = args[1] if name.startswith("aten.copy_") else args[0]
            dst_dev = args[0].device.type if name.startswith("aten.copy_") else str((kwargs or {}).get("device", src.device))[:4]
            kind = f" {src.device.type}->{dst_dev} n={src.numel()}"
        rec = sites[(name + kind, where)]
        rec[0] += 1
        rec[1] += dt * 1e6
        return out


with Spy():
    for i in range(STEPS):
        step(i)
    torch.cuda.synchronize()
rows = sorted(sites.items(), key=lambda kv: -kv[1][1])
print(f"{'us/step':>9} {'calls/step':>10}  op  site")
for (name, where), (n, us) in rows[:60]:
    print(f"{us / STEPS:9.1f} {n / STEPS:10.1f}  {name}  {where}")
