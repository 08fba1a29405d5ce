"""Who is ahead, host or GPU, phase by phase of an un-profiled training step?  Around every phase of the step a pair of
events is recorded on the training stream and the host clock is read (no synchronisation inside the step); afterwards each
phase gets: host time spent issuing it, GPU time between its two events, and the LAG of the GPU behind the host at the end
of the phase (event time on the host's clock - host time).  A phase whose lag stays near zero is host-bound (the queue is
empty, the GPU waits for launches); a growing lag means the GPU is the one being waited for."""
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
from gapartnet_amd.network import net_exec

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(4)]
model.train()
WARM, STEPS = 10, 30
feed = iter(DevicePrefetcher((pool[i % 4] for i in range(WARM + STEPS + 1)), model, dev))
marks = []  # (label, kind, host time, event)
ON = [False]


def mark(label, kind):
    if ON[0]:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((label, kind, time.perf_counter(), ev))


def wrap(owner, attr, label):
    fn = getattr(owner, attr)

    def inner(*a, **k):
        mark(label, 0)
        out = fn(*a, **k)
        mark(label, 1)
        return out
    setattr(owner, attr, inner)


for name in ["forward_backbone", "forward_sem_seg", "forward_offset", "proposal_clustering_and_revoxelize",
             "forward_proposal_unets", "forward_proposal_score", "loss_proposal_score", "forward_proposal_npcs",
             "loss_proposal_npcs", "log"]:
    wrap(model, name, name)
_orig_call = net_exec._call
_orig_call_pair = net_exec._call_pair
counter = defaultdict(int)
CALLS_PER_STEP = {"gpn_net_forward": 1, "gpn_net_backward": 1}  # (the proposal networks run as a pair)


def _call(fn_name, *a, **k):
    counter[fn_name] += 1
    label = f"  {fn_name[8:]} call {(counter[fn_name] - 1) % CALLS_PER_STEP.get(fn_name, 1)}"
    mark(label, 0)
    out = _orig_call(fn_name, *a, **k)
    mark(label, 1)
    return out


def _call_pair(fn_name, *a, **k):
    label = f"  {fn_name[8:]}"
    mark(label, 0)
    out = _orig_call_pair(fn_name, *a, **k)
    mark(label, 1)
    return out


net_exec._call = _call
net_exec._call_pair = _call_pair


def step(i):
    mark("step", 0)
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    mark("backward", 0)
    loss.backward()
    mark("backward", 1)
    mark("optimizer", 0)
    opt.step()
    mark("optimizer", 1)
    mark("step", 1)


for i in range(WARM):
    step(i)
torch.cuda.synchronize()
ON[0] = True
ref = torch.cuda.Event(enable_timing=True)
ref.record()
torch.cuda.synchronize()
t_ref = time.perf_counter()
t0 = time.perf_counter()
for i in range(STEPS):
    step(WARM + i)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / STEPS * 1e3
stats = defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0])  # host ms, gpu ms, lag at start, lag at end, n
open_marks = {}
order = []
for label, kind, th, ev in marks:
    tg = ref.elapsed_time(ev)  # ms since ref on the GPU
    th_ms = (th - t_ref) * 1e3
    if kind == 0:
        open_marks[label] = (th_ms, tg)
    else:
        h0, g0 = open_marks.pop(label)
        s = stats[label]
        s[0] += th_ms - h0
        s[1] += tg - g0
        s[2] += g0 - h0
        s[3] += tg - th_ms
        s[4] += 1
        if label not in order:
            order.append(label)
print(f"step {wall:.3f} ms (with ~{len(marks) / STEPS:.0f} event records per step)")
print(f"{'phase':40s} {'host ms':>8s} {'GPU ms':>8s} {'lag@start':>10s} {'lag@end':>9s}")
for label in order:
    h, g, l0, l1, n = stats[label]
    per = n / STEPS
    print(f"{label:40s} {h / STEPS:8.3f} {g / STEPS:8.3f} {l0 / n:10.3f} {l1 / n:9.3f}   x{per:.0f}")
