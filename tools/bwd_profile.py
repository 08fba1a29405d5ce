"""host-side profile of loss.backward() alone (autograd nodes and the ops they dispatch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
batch = [pc.to(dev) for pc in make_batch(8, 20000)]
for i in range(3):
    opt.zero_grad(set_to_none=True); model.training_step(batch, i).backward(); opt.step()
N = 4
losses = []
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(N):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, i)
        torch.cuda.synchronize()
        with torch.profiler.record_function("BACKWARD"):
            loss.backward()
        torch.cuda.synchronize()
        opt.step()
ev = prof.key_averages()
rows = [(e.key, e.self_cpu_time_total / N, e.count / N) for e in ev if "Backward" in e.key or e.key == "BACKWARD" or "evaluate_function" in e.key]
rows.sort(key=lambda r: -r[1])
for k, t, c in rows[:40]:
    print(f"{k[:70]:70s} {t:9.1f} us/step  x{c:.0f}")
