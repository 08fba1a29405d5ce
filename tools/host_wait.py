"""How long the host WAITS for the device in a training step (un-instrumented apart from a timer around the two
`tolist()` reads the step makes): step time - waits = time the host is busy issuing work.  If the waits are a small part of
the step, the host is the critical resource (the GPU idles behind it); if they are large, the GPU is."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(4)]
model.train()
WARM, STEPS = 10, 40
feed = iter(DevicePrefetcher((pool[i % 4] for i in range(WARM + STEPS + 1)), model, dev))
waits = {}
orig = torch.Tensor.tolist


def timed(self):
    if not self.is_cuda:
        return orig(self)
    t0 = time.perf_counter()
    out = orig(self)
    dt = time.perf_counter() - t0
    f = sys._getframe(1)
    key = f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}"
    w = waits.setdefault(key, [0, 0.0])
    w[0] += 1
    w[1] += dt
    return out


def step(i):
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()


for i in range(WARM):
    step(i)
torch.cuda.synchronize()
torch.Tensor.tolist = timed
t0 = time.perf_counter()
for i in range(STEPS):
    step(WARM + i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
torch.Tensor.tolist = orig
print(f"step {t_all / STEPS * 1e3:.3f} ms (host loop returns after {t_host / STEPS * 1e3:.3f} ms per step)")
tot = 0.0
for k, (n, s) in waits.items():
    print(f"  wait in {k}: {s / STEPS * 1e3:.3f} ms/step over {n / STEPS:.1f} reads/step")
    tot += s
print(f"host busy (step - waits): {(t_all - tot) / STEPS * 1e3:.3f} ms/step; waits {tot / STEPS * 1e3:.3f} ms/step")
