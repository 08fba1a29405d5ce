"""count the aten ops (host dispatches) each phase of the training step issues"""
import os, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from gapartnet_amd.smoke import make_batch, make_model

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
batch = [pc.to(dev) for pc in make_batch(8, 20000)]
PHASES = ["_collate", "forward_backbone", "forward_sem_seg", "loss_sem_seg", "forward_offset", "loss_offset",
          "proposal_clustering_and_revoxelize", "forward_proposal_score", "loss_proposal_score", "forward_proposal_npcs",
          "loss_proposal_npcs"]
current = ["other-forward"]
counts = Counter()
ops_by_phase = {}


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        counts[current[0]] += 1
        ops_by_phase.setdefault(current[0], Counter())[str(func).replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))


def wrap(name):
    fn = getattr(model, name)

    def inner(*a, **k):
        prev = current[0]
        current[0] = name
        try:
            return fn(*a, **k)
        finally:
            current[0] = prev
    setattr(model, name, inner)


for p in PHASES:
    wrap(p)
for _ in range(2):
    opt.zero_grad(set_to_none=True)
    model.training_step(batch, 0).backward()
    opt.step()
with Count():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    current[0] = "backward"
    loss.backward()
    current[0] = "optimizer"
    opt.step()
for k, v in counts.most_common():
    top = ", ".join(f"{n} x{c}" for n, c in ops_by_phase[k].most_common(8))
    print(f"{k:38s} {v:5d}   {top}")
