"""ms per training step in windows of 10 steps from a cold process (same feed as bench.py): how long the step takes to reach its
steady state, and what the caching allocator does meanwhile (hipMalloc count, reserved bytes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
N = int(os.environ.get("STEPS", 200))
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(N)), model, dev))
torch.cuda.synchronize()
t0 = t_step = time.perf_counter()
for i in range(N):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(next(feed), i)
    loss.backward(); opt.step()
    if os.environ.get("PER_STEP") and 18 <= i < 32:
        torch.cuda.synchronize()
        print(f"  step {i}: {(time.perf_counter() - t_step) * 1e3:.1f} ms", flush=True)
    t_step = time.perf_counter()
    if i % 10 == 9:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = torch.cuda.memory_stats()
        print(f"steps {i - 9:3d}-{i:3d}: {(t1 - t0) * 100:.2f} ms/step  reserved {torch.cuda.memory_reserved() >> 20} MB  "
              f"hipMalloc calls {st['num_device_alloc']}  frees {st['num_device_free']}  alloc retries {st['num_alloc_retries']}")
        t0 = time.perf_counter()
