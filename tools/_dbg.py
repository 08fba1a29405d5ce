import cProfile, pstats, sys, os, io, time, tempfile, shutil
sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.argv = ['x']
from tools.pth_loader_bench import write_dataset
from gapartnet_amd.dataset.gapartnet import GAPartNetInst
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
from gapartnet_amd.smoke import make_model
device = torch.device("cuda:0")
root = tempfile.mkdtemp(prefix="gpn_pth_")
write_dataset(root, 128, 8, 20000)
dm = GAPartNetInst(root, max_points=20000, train_batch_size=8, val_batch_size=8, test_batch_size=8, num_workers=8, pos_jitter=0.1,
                   color_jitter=0.3, flip_prob=0.3, rotate_prob=0.3, packed_cache=True)
dm.setup("fit")
model = make_model((0, 0)).to(device).train()
opt = model.configure_optimizers()
def epoch(prof=None):
    feed = DevicePrefetcher(dm.train_dataloader(), model, device, augmentation=dm.aug)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    if prof: prof.enable()
    for i, batch in enumerate(feed):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, i); loss.backward(); opt.step(); n += 1
    torch.cuda.synchronize()
    if prof: prof.disable()
    return (time.perf_counter() - t0) / n * 1e3
print("warm", epoch()); print("ms/step", epoch(), epoch())
pr = cProfile.Profile(); print("profiled", epoch(pr))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:7000])
shutil.rmtree(root, ignore_errors=True)
