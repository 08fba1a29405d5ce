"""Cost of the gradient exchange wrapper on ONE GPU (world_size 1 over RCCL): ms/step and the host-side op table for
MODE=none | sync (grad_sync.GradSync) | ddp (torch DistributedDataParallel, find_unused_parameters), USE_PF=1 to feed
through the DevicePrefetcher like bench.py.  This is the measurement behind grad_sync.py's header."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from torch.profiler import ProfilerActivity, profile
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.trainer import _TrainStep

MODE, USE_PF = os.environ.get("MODE", "sync"), bool(os.environ.get("USE_PF"))
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
if MODE != "none":
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
model = make_model((0, 0)).to(dev)
opt = model.configure_optimizers()
mod, sync = model.training_step, None
if MODE == "ddp":
    from torch.nn.parallel import DistributedDataParallel as DDP
    mod = DDP(_TrainStep(model), device_ids=[0], find_unused_parameters=True, broadcast_buffers=False)
elif MODE == "sync":
    from gapartnet_amd.grad_sync import GradSync
    sync = GradSync(model)
    if os.environ.get("BCAST"):
        sync.broadcast_parameters()
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(2)]
feed = iter(DevicePrefetcher((pool[i % 2] for i in range(100)), model, dev)) if USE_PF else None


def step(i):
    batch = next(feed) if USE_PF else pool[i % 2]
    opt.zero_grad(set_to_none=True)
    loss = mod(batch, i)
    loss.backward()
    if sync is not None:
        sync.sync()
    opt.step()


WARM, STEPS = int(os.environ.get('WARM', 6)), int(os.environ.get('STEPS', 20))
for i in range(WARM): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(STEPS): step(i)
torch.cuda.synchronize(); print(f"MODE={MODE} USE_PF={USE_PF} ms/step", (time.perf_counter() - t0) / STEPS * 1e3, sync.stats if sync else "")
if os.environ.get("TABLE"):
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for i in range(3): step(i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
if MODE != "none":
    dist.destroy_process_group()
