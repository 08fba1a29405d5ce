"""Wall time of every step of a bench-like run (no synchronisation inside the step; the time stamp is taken when the host
returns from the step): finds one-off stalls that a mean over the run hides."""
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
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
model.train()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(N + 1)), model, dev))
import gc
MODE = os.environ.get("GC_MODE", "")
if MODE == "callback":
    def _cb(phase, info):
        if phase == "start":
            _cb.t0 = time.perf_counter()
        else:
            d = (time.perf_counter() - _cb.t0) * 1e3
            if d > 5:
                print(f"    gc generation {info['generation']}: {d:.1f} ms, collected {info['collected']}", flush=True)
    gc.callbacks.append(_cb)
stamps = [time.perf_counter()]
for i in range(N):
    if i == 20 and MODE == "freeze":
        gc.collect()
        gc.freeze()
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()
    stamps.append(time.perf_counter())
torch.cuda.synchronize()
dt = [(b - a) * 1e3 for a, b in zip(stamps, stamps[1:])]
med = sorted(dt[20:])[len(dt[20:]) // 2]
print(f"median step (after 20): {med:.2f} ms")
for i, d in enumerate(dt):
    if i >= 5 and d > 1.8 * med:
        print(f"  step {i}: {d:.1f} ms")
for a, b in ((5, 10), (10, 20), (20, 30), (30, 50), (50, 80), (80, 120), (120, 200), (200, 300)):
    if b <= len(dt):
        print(f"  steps {a:3d}-{b:3d}: mean {sum(dt[a:b]) / (b - a):6.2f} ms")
print("memory reserved MiB:", torch.cuda.memory_reserved() >> 20, "allocated:", torch.cuda.memory_allocated() >> 20)
