import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.network import net_exec
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
batch = [pc.to(dev) for pc in make_batch(8, 20000)]
acc = {"call": 0.0, "fwd": 0.0, "bwd": 0.0, "run": 0.0}
oc = net_exec._call
def tc(*a, **k):
    t = time.perf_counter(); r = oc(*a, **k); acc["call"] += time.perf_counter() - t; return r
net_exec._call = tc
of, ob = net_exec._NetFn.forward, net_exec._NetFn.backward
def tf(ctx, *a):
    t = time.perf_counter(); r = of(ctx, *a); acc["fwd"] += time.perf_counter() - t; return r
def tb(ctx, *a):
    t = time.perf_counter(); r = ob(ctx, *a); acc["bwd"] += time.perf_counter() - t; return r
net_exec._NetFn.forward = staticmethod(tf); net_exec._NetFn.backward = staticmethod(tb)
orun = net_exec.run
def trun(*a, **k):
    t = time.perf_counter(); r = orun(*a, **k); acc["run"] += time.perf_counter() - t; return r
net_exec.run = trun
for i in range(3):
    opt.zero_grad(set_to_none=True); model.training_step(batch, i).backward(); opt.step()
for k in acc: acc[k] = 0.0
N = 10
for i in range(N):
    opt.zero_grad(set_to_none=True); model.training_step(batch, i).backward(); opt.step()
torch.cuda.synchronize()
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms per step (3 nets): run = whole net_exec.run incl. rulebooks + fwd; fwd/bwd = autograd Function bodies; call = library calls")
