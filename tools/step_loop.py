"""A plain training-step loop of the bench's default workload with diagnostic switches, meant to be run under
`rocprofv3 --kernel-trace` and read with tools/fwd_bwd_split.py:
  --freeze-convs   conv weights do not require a gradient: no weight-gradient launches (what do the dgrad kernels cost
                   without the contractions running beside them on the second stream?)
  --no-bn-fuse     BatchNorm sums as separate launches (what do the conv epilogues cost?)
  --steps N"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import _C
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.dataset.prefetch import DevicePrefetcher
from gapartnet_amd.spconv import pytorch as spconv

ap = argparse.ArgumentParser()
ap.add_argument("--freeze-convs", action="store_true")
ap.add_argument("--no-bn-fuse", action="store_true")
ap.add_argument("--steps", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda:0")
model = make_model((0, 0)).to(dev)
if args.freeze_convs:
    for m in model.modules():
        if isinstance(m, (spconv.SubMConv3d, spconv.SparseConv3d, spconv.SparseInverseConv3d)):
            m.weight.requires_grad_(False)
if args.no_bn_fuse:
    _C.lib().gpn_net_bn_fusion(0)
opt = model.configure_optimizers()
pool = [[pc.to(dev) for pc in make_batch(8, 20000, seed0=1000 + 8 * j)] for j in range(4)]
model.train()
feed = iter(DevicePrefetcher((pool[i % 4] for i in range(args.steps + 1)), model, dev))
import time
for i in range(args.steps):
    if i == args.steps - 6:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    b = next(feed)
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print(f"last 6 steps: {(time.perf_counter() - t0) / 6 * 1e3:.3f} ms/step")
