"""Which host call stalls for ~50 ms once, around step 23 of a cold run (tools/warmup_curve.py)?  Profiles steps 20-26 and
prints the calls with the largest single duration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(40)), model, dev))


def step(i):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(next(feed), i)
    loss.backward(); opt.step()


for i in range(20):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(20, 27):
        t0 = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
evs = sorted(prof.events(), key=lambda e: -e.self_cpu_time_total)[:12]
for e in evs:
    print(f"{e.self_cpu_time_total / 1e3:9.2f} ms self  {e.name[:90]}  shapes={getattr(e, 'input_shapes', '')}")
