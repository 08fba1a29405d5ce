import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.trainer import _TrainStep
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
mod = DDP(_TrainStep(model), device_ids=[0], find_unused_parameters=True, broadcast_buffers=False)
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
USE_PF = bool(os.environ.get("USE_PF"))
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(100)), model, dev)) if USE_PF else None
def step(i):
    batch = next(feed) if USE_PF else pool[i % 2]
    opt.zero_grad(set_to_none=True); loss = mod(batch, i); loss.backward(); opt.step()
for i in range(4): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): step(i)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 10 * 1e3)
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(3): step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
dist.destroy_process_group()
