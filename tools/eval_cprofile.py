"""cProfile of the host side of validation steps (tools/eval_bench.py's loop: config 2 shape, bs 4 x 20k, eval mode): where the
host's time per step goes once the GPU is no longer the bound behind the backbone.  Prints the top functions by own time and by
cumulative time, per validation step."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
from gapartnet_amd.smoke import make_batch, make_model
from tests.golden import recipe

dev = torch.device("cuda:0")
model = make_model((0, 0)).eval()
model.load_state_dict(recipe.name_keyed_state(model))
model = model.to(dev)
model._log_sink = lambda name, value, bs, sync: None
model.defer_validation_outputs = True
pool = [[pc.to(dev) for pc in make_batch(4, 20000, seed0=2000 + 10 * j)] for j in range(2)]
STEPS = 48


def epoch():
    feed = DevicePrefetcher((pool[i % 2] for i in range(STEPS)), model, dev)
    for i, batch in enumerate(feed):
        model.validation_step(batch, i, 0)
    model._resolve_pending_outputs()
    model.validation_step_outputs = [[] for _ in model.validation_step_outputs]


with torch.no_grad():
    epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    epoch()
    torch.cuda.synchronize()
    print(f"un-profiled: {(time.perf_counter() - t0) / STEPS * 1e3:.3f} ms per validation step")
    pr = cProfile.Profile()
    pr.enable()
    epoch()
    torch.cuda.synchronize()
    pr.disable()
for key in ("tottime", "cumulative"):
    st = pstats.Stats(pr)
    st.sort_stats(key)
    print(f"\n== top by {key} (totals over {STEPS} steps; divide by {STEPS})")
    st.print_stats(28)
