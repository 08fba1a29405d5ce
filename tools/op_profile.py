"""torch.profiler (CPU side only) over a few training steps: which aten / custom ops the host spends its time in."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
batch = [pc.to(dev) for pc in make_batch(8, 20000)]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=48))
