import os, sys, time, resource
sys.path.insert(0, "/root/repo")
import torch
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(4)]
N = 300
feed = iter(DevicePrefetcher((pool[i % 4] for i in range(N)), model, dev))
t0 = time.time()
for i in range(N):
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(next(feed), i)
    loss.backward(); opt.step()
    if i in (20, 100, 299):
        torch.cuda.synchronize()
        print(i, f"loss {float(loss):.9f} cuda alloc {torch.cuda.memory_allocated() >> 20} MB reserved {torch.cuda.memory_reserved() >> 20} MB "
              f"peak {torch.cuda.max_memory_allocated() >> 20} MB host rss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10} MB  {(time.time() - t0) / (i + 1) * 1e3:.1f} ms/step")
