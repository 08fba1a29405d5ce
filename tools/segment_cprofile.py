"""cProfile of the host-bound stretch of the training step (after the proposal stage's host read: paired proposal U-Nets,
heads, proposal losses): cumulative time per function, callees of the big ones."""
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
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(4)]
model.train()
WARM, STEPS = 10, 40
feed = iter(DevicePrefetcher((pool[i % 4] for i in range(WARM + STEPS + 1)), model, dev))
pr = cProfile.Profile()
names = ["forward_proposal_unets", "forward_proposal_score", "loss_proposal_score", "forward_proposal_npcs", "loss_proposal_npcs"]
for n in names:
    fn = getattr(model, n)

    def wrapped(*a, _fn=fn, **k):
        pr.enable()
        try:
            return _fn(*a, **k)
        finally:
            pr.disable()
    setattr(model, n, wrapped)


def step(i):
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()


for i in range(WARM):
    step(i)
torch.cuda.synchronize()
pr.clear()
for i in range(STEPS):
    step(WARM + i)
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.strip_dirs()
rows = []
for (f, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((ct / STEPS * 1e3, tt / STEPS * 1e3, nc / STEPS, f"{f}:{line} {name}"))
rows.sort(reverse=True)
print(f"{'cum ms/step':>11s} {'own ms/step':>11s} {'calls/step':>10s}")
for ct, tt, nc, name in rows[:45]:
    print(f"{ct:11.3f} {tt:11.3f} {nc:10.1f}  {name[:110]}")
