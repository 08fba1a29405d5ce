"""Where the forced 1-rank gradient exchange costs time: per-phase wall time with the queue drained around each phase, with and
without the exchange (same process group either way).  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
if os.environ.get("PROBE_ENV") == "early":
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
import torch.distributed as dist

if os.environ.get("PROBE_ENV") == "late":  # after `import torch`, before the first HIP call
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mode = os.environ.get("PROBE_GROUP", "1")
if mode == "1":
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
elif mode == "lazy":   # communicator created at the first collective (none happens without the exchange)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
elif mode == "gloo":
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
elif mode == "warm":   # NCCL group + one collective up front
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher

model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
gs = None
if os.environ.get("PROBE_SYNC", "1") == "1":
    from gapartnet_amd.grad_sync import GradSync
    gs = GradSync(model); gs.broadcast_parameters()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + j * 8)] for j in range(2)]
model.train()
N = 40
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(N + 11)), model, dev))
acc = {"fwd": 0.0, "bwd": 0.0, "sync": 0.0, "opt": 0.0}
drain = os.environ.get("PROBE_DRAIN", "1") == "1"


def tick():
    if drain:
        torch.cuda.synchronize()
    return time.perf_counter()


for i in range(N + 10):
    b = next(feed)
    if i == 10:
        for k in acc:
            acc[k] = 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
    t0 = tick(); opt.zero_grad(set_to_none=True); loss = model.training_step(b, i)
    t1 = tick(); loss.backward()
    t2 = tick()
    if gs is not None:
        gs.sync()
    t3 = tick(); opt.step()
    t4 = tick()
    for k, d in (("fwd", t1 - t0), ("bwd", t2 - t1), ("sync", t3 - t2), ("opt", t4 - t3)):
        acc[k] += d
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / N * 1e3
print(f"group={os.environ.get('PROBE_GROUP', '1')} sync={os.environ.get('PROBE_SYNC', '1')} drain={int(drain)}: step {total:.2f} ms  " +
      "  ".join(f"{k} {v / N * 1e3:.2f}" for k, v in acc.items()), flush=True)
