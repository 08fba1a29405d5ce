"""cProfile of the host side of training steps (bench feed: prefetcher + model + FusedAdam): top functions by own time."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + j * 8)] for j in range(2)]
model.train()
N = 30
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(N + 11)), model, dev))


def step(i):
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()


for i in range(10):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(10 + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
total = sum(v[2] for v in st.stats.values())
print(f"profiled {N} steps, {total / N * 1e3:.2f} ms of own time per step (profiler overhead included)")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:45]
for (fn, line, name), (cc, nc, tt, ct, callers) in rows:
    short = os.path.relpath(fn, ROOT) if fn.startswith(ROOT) else os.path.basename(fn)
    print(f"{tt / N * 1e3:7.3f} ms/step  {nc / N:7.1f} calls/step  {short}:{line} {name}")
# who calls the allocator / the small tensor ops (calls per step by caller)
for wanted in ("torch.empty", "'view'", "'data_ptr'", "'to' of"):
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        if wanted in name:
            print(f"\ncallers of {name}:")
            for (cfn, cline, cname), (ccc, cnc, ctt, cct) in sorted(callers.items(), key=lambda kv: -kv[1][1])[:10]:
                short = os.path.relpath(cfn, ROOT) if cfn.startswith(ROOT) else os.path.basename(cfn)
                print(f"   {cnc / N:7.1f} calls/step  {ctt / N * 1e3:6.3f} ms/step  {short}:{cline} {cname}")
