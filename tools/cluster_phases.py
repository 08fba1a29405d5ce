"""host time (with a sync after each part) of the pieces of proposal_clustering_and_revoxelize"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.network import model as M, grouping_utils as G
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
batch = [pc.to(dev) for pc in make_batch(8, 20000)]
for i in range(3):
    opt.zero_grad(set_to_none=True); model.training_step(batch, i).backward(); opt.step()
captured = {}
orig = model.proposal_clustering_and_revoxelize
def grab(**kw):
    captured.update(kw); return orig(**kw)
model.proposal_clustering_and_revoxelize = grab
with torch.no_grad():
    model.training_step(batch, 0)
acc = {}
def timed(name, fn):
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        h, t = acc.get(name, (0, 0)); acc[name] = (h + t1 - t0, t + t2 - t0)
        return r
    return inner
M.cluster_proposals = timed("cluster_proposals (x2)", G.cluster_proposals)
M.segmented_voxelize = timed("segmented_voxelize", G.segmented_voxelize)
N = 10
with torch.no_grad():
    for _ in range(3):
        orig(**captured)
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        orig(**captured)
    torch.cuda.synchronize(); total = (time.perf_counter() - t0) / N * 1e3
print(f"whole function: {total:.2f} ms per call (with the syncs added by this tool)")
for k, (h, t) in acc.items():
    print(f"  {k:28s} host {h / N * 1e3:.2f} ms   host+gpu {t / N * 1e3:.2f} ms")
print("valid points:", int((captured['sem_preds'] > 0).sum()), "of", captured['sem_preds'].shape[0])
