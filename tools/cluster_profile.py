"""kernel-level profile of the clustering / re-voxelisation phase alone"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
batch = [pc.to(dev) for pc in make_batch(8, 20000)]
captured = {}
orig = model.proposal_clustering_and_revoxelize
def grab(**kw):
    captured.update(kw)
    return orig(**kw)
model.proposal_clustering_and_revoxelize = grab
with torch.no_grad():
    model.training_step(batch, 0)
    for _ in range(3):
        orig(**captured)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        orig(**captured)
    torch.cuda.synchronize()
    print("ms per call", (time.perf_counter() - t0) / 5 * 1e3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        orig(**captured)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
