"""Every torch (aten) operator a training step still dispatches, by Python call site: the launches of the step that are not
hand-written kernels.  A TorchDispatchMode sees what reaches the backend (a no-op `.to()` does not, a `zeros` is one
`aten.zeros`); it is thread-local, so the autograd engine's own thread is not covered - backward ops are listed from a
CPU-side torch profile below the table, by name only."""
import os
import sys
import traceback
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher

dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
model.train()
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(14)), model, dev))


def step(i):
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()


for i in range(6):
    step(i)
torch.cuda.synchronize()
STEPS = 3
NO_KERNEL = ("view", "reshape", "as_strided", "select.", "slice.", "empty", "detach", "unsqueeze", "squeeze", "aten.t.",
             "transpose", "expand", "alias", "_unsafe_view", "narrow", "permute", "unbind", "split", "resize_", "set_",
             "_local_scalar_dense", "lift_fresh", "record_stream", "is_pinned", "_reshape_alias", "stride", "sym_")
sites = Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if any(k in name for k in NO_KERNEL):
            return out
        on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or \
            (isinstance(out, torch.Tensor) and out.is_cuda)
        if not on_gpu:
            return out
        frames = [f for f in traceback.extract_stack() if "/gapartnet_amd/" in f.filename]
        where = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in reversed(frames[-2:])) or "(outside)"
        sites[(name, where)] += 1
        return out


with Spy():
    for i in range(STEPS):
        step(6 + i)
    torch.cuda.synchronize()
print(f"aten ops on device tensors reaching the backend (forward + optimizer, main thread): {sum(sites.values()) / STEPS:.1f} per step")
for (name, where), n in sites.most_common(150):
    print(f"{n / STEPS:6.1f}/step  {name:34s} {where}")

from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(STEPS):
        step(9 + i)
    torch.cuda.synchronize()
names = Counter()
main_tid = None
for ev in prof.events():
    if ev.name.startswith("aten::") and (ev.cpu_parent is None or not ev.cpu_parent.name.startswith("aten::")):
        names[(ev.thread, ev.name)] += 1
print("\ntop-level aten ops by thread (the autograd engine's thread is the one with the *_backward ops):")
for (tid, name), n in sorted(names.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if not any(k in name for k in ("view", "reshape", "empty", "as_strided", "select", "slice", "detach", "squeeze", "expand",
                                   "transpose", "aten::t", "alias", "narrow", "permute", "item", "_local_scalar", "aten::to",
                                   "lift_fresh", "result_type", "resolve_")):
        print(f"  thread {tid}: {n / STEPS:6.1f}/step  {name}")
