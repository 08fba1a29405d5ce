"""cProfile of the host side of a few bench steps (which Python / ctypes / torch calls the step spends its CPU time in)."""
import cProfile
import pstats
import sys
import os
import time

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


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)

# host-side time of each phase (no sync inside the phase; the queue is drained between phases)
acc = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0, "gpu_total": 0.0}
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    acc["fwd"] += t1 - t0
    acc["bwd"] += t2 - t1
    acc["opt"] += t3 - t2
    acc["gpu_total"] += t4 - t0
print({k: round(v / 5 * 1e3, 2) for k, v in acc.items()})
