"""List every host<->device synchronisation of one training step (torch's sync debug mode), grouped by call site."""
import os
import sys
import traceback
import warnings
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
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


for _ in range(2):
    step()
sites = Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    full = traceback.extract_stack()
    stack = [f for f in full if "/gapartnet_amd/" in f.filename] or [f for f in full if "sync_sites" not in f.filename][-4:]
    key = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in reversed(stack[-3:]))
    sites[key] += 1


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
print("total syncs in one step:", sum(sites.values()))
for k, v in sites.most_common():
    print(f"{v:3d}  {k}")
