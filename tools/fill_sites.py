"""Which Python call sites the fill / copy kernels of a training step come from (torch profiler with stacks)."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
model.train()
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(12)), model, dev))


def step(i):
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()


for i in range(6):
    step(i)
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    for i in range(STEPS):
        step(6 + i)
    torch.cuda.synchronize()
WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::constant_pad_nd", "aten::_to_copy", "aten::clone",
         "aten::contiguous")
sites = Counter()
for ev in prof.events():
    if ev.name not in WATCH or ev.cpu_parent is not None and ev.cpu_parent.name in WATCH:
        continue
    stack = [s for s in (ev.stack or []) if "gapartnet_amd" in s]
    where = " <- ".join(s.split("/root/repo/")[-1].split(ROOT + "/")[-1][:70] for s in stack[:2]) or "(autograd engine / outside)"
    sites[(ev.name, where)] += 1
for (name, where), n in sites.most_common(45):
    print(f"{n / STEPS:6.1f}/step  {name:24s} {where}")
