"""Host time inside loss.backward(), by custom autograd Function (their backward runs on autograd's device thread, where
cProfile does not look): seconds between entry and return of each Function.backward, per step; the rest of backward() is
the engine itself and torch's own nodes."""
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import functional as GF
from gapartnet_amd.network import net_exec
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(4)]
model.train()
WARM, STEPS = 10, 30
feed = iter(DevicePrefetcher((pool[i % 4] for i in range(WARM + STEPS + 1)), model, dev))
acc = defaultdict(lambda: [0.0, 0, 0.0])
seq = []


def wrap(cls):
    orig = cls.backward

    def timed(ctx, *grads):
        t0 = time.perf_counter()
        out = orig(ctx, *grads)
        t1 = time.perf_counter()
        a = acc[cls.__name__]
        a[0] += t1 - t0
        a[1] += 1
        seq.append((cls.__name__, t0, t1))
        return out
    cls.backward = staticmethod(timed)


for mod in (GF, net_exec):
    for name in dir(mod):
        obj = getattr(mod, name)
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            wrap(obj)
total = [0.0]


def step(i):
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    seq.append(("<backward begins>", time.perf_counter(), time.perf_counter()))
    t0 = time.perf_counter()
    loss.backward()
    total[0] += time.perf_counter() - t0
    seq.append(("<backward returns>", time.perf_counter(), time.perf_counter()))
    opt.step()


for i in range(WARM):
    step(i)
torch.cuda.synchronize()
for a in acc.values():
    a[0], a[1] = 0.0, 0
total[0] = 0.0
seq.clear()
for i in range(STEPS):
    step(WARM + i)
torch.cuda.synchronize()
print(f"loss.backward(): {total[0] / STEPS * 1e3:.3f} ms of host time per step")
inside = 0.0
for name, (s, n, _) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {s / STEPS * 1e3:7.3f} ms/step  {n / STEPS:5.1f} x  {name}.backward")
    inside += s
print(f"  {(total[0] - inside) / STEPS * 1e3:7.3f} ms/step  engine, torch's own nodes, AccumulateGrad")
# the last step as a timeline
last = [i for i, s in enumerate(seq) if s[0] == "<backward begins>"][-1]
t0 = seq[last][1]
print("last step, ms since backward() was called:")
prev_end = t0
for name, a, b in seq[last:]:
    print(f"  {1e3 * (a - t0):7.3f} .. {1e3 * (b - t0):7.3f}  (+{1e3 * (a - prev_end):6.3f} idle before)  {name}")
    prev_end = b
