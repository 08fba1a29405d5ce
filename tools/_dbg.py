import cProfile, pstats, sys, os, io, time
sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gapartnet_amd.smoke import make_batch, make_model
from tests.golden import recipe
dev = torch.device("cuda:0")
model = make_model((0, 0)).eval()
model.load_state_dict(recipe.name_keyed_state(model))
model = model.to(dev)
model._log_sink = lambda name, value, bs, sync: None
pools = [[pc.to(dev) for pc in make_batch(4, 20000, seed0=2000 + 10 * j)] for j in range(2)]
with torch.no_grad():
    for i in range(6): model.validation_step(pools[i % 2], i, 0)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for i in range(20): model.validation_step(pools[i % 2], i, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pr.disable()
print("ms/step under cProfile", (t1 - t0) / 20 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60); print(s.getvalue()[:9000])
