"""Per-phase timing of one training step: host dispatch time (no sync) vs wall time with the queue drained around
each phase.  Shows which phases are bound by the GPU and which by the host."""
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd.smoke import make_batch, make_model

dev = torch.device("cuda:0")
if os.environ.get("PROBE_GROUP") == "1":  # with an RCCL communicator alive (what every rank of a multi-GPU run has)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
batch = [pc.to(dev) for pc in make_batch(8, 20000)]
PHASES = ["_collate", "forward_backbone", "forward_sem_seg", "loss_sem_seg", "forward_offset", "loss_offset",
          "proposal_clustering_and_revoxelize", "forward_proposal_score", "loss_proposal_score", "forward_proposal_npcs",
          "loss_proposal_npcs"]
acc = defaultdict(float)
SYNC = [False]


def wrap(name):
    fn = getattr(model, name)

    def inner(*a, **k):
        if SYNC[0]:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        if SYNC[0]:
            torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0
        return out
    setattr(model, name, inner)


for p in PHASES:
    wrap(p)


def wrap_fn(owner, attr, label):
    fn = getattr(owner, attr)

    def inner(*a, **k):
        if SYNC[0]:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        if SYNC[0]:
            torch.cuda.synchronize()
        acc[label] += time.perf_counter() - t0
        return out
    setattr(owner, attr, inner)


from gapartnet_amd import hip_ops
from gapartnet_amd.network import net_exec
SUB = ["  rulebook_subm3", "  rulebook_down", "  voxelize", "  net forward call", "  net backward call", "  ball_query", "  ccl"]
wrap_fn(hip_ops, "rulebook_subm3", SUB[0])
wrap_fn(hip_ops, "rulebook_down", SUB[1])
wrap_fn(hip_ops, "voxelize", SUB[2])
wrap_fn(hip_ops, "ball_query", SUB[5])
wrap_fn(hip_ops, "ccl", SUB[6])
_orig_call = net_exec._call


def _timed_call(fn_name, *a, **k):
    label = SUB[3] if fn_name == "gpn_net_forward" else SUB[4]
    if SYNC[0]:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = _orig_call(fn_name, *a, **k)
    if SYNC[0]:
        torch.cuda.synchronize()
    acc[label] += time.perf_counter() - t0
    return out


net_exec._call = _timed_call


def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    if SYNC[0]:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    if SYNC[0]:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    if SYNC[0]:
        torch.cuda.synchronize()
    t3 = time.perf_counter()
    acc["TOTAL forward"] += t1 - t0
    acc["backward"] += t2 - t1
    acc["optimizer"] += t3 - t2


for _ in range(3):
    step()
res = {}
for sync in (False, True):
    SYNC[0] = sync
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    res[sync] = (dict(acc), (time.perf_counter() - t0) / 5 * 1e3)
print(f"step: {res[False][1]:.2f} ms free-running, {res[True][1]:.2f} ms with a sync around every phase")
print(f"{'phase':40s} {'host ms':>9s} {'synced ms':>10s}")
for k in PHASES + ["TOTAL forward", "backward", "optimizer"] + SUB:
    print(f"{k:40s} {res[False][0].get(k, 0) / 5 * 1e3:9.2f} {res[True][0].get(k, 0) / 5 * 1e3:10.2f}")
