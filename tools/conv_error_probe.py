"""max error of the fused conv (forward and dgrad) against a float64 evaluation of the same pair lists"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gapartnet_amd import hip_ops as H
from tests import synth
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for shape, per, cin, cout in [((128,)*3, 4000, 16, 16), ((64,)*3, 2000, 32, 32), ((40,)*3, 1000, 48, 48), ((24,)*3, 350, 64, 64),
                              ((24,)*3, 350, 128, 64), ((24,)*3, 350, 64, 128), ((16,)*3, 150, 80, 80), ((64,)*3, 2000, 64, 32)]:
    idx = torch.from_numpy(synth.surface_indices(rng, 2, list(shape), per)).to(dev)
    n = idx.shape[0]
    rb = H.rulebook_subm3(idx, list(shape))
    P = int(rb.num_pairs.item())
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) / np.sqrt(27 * cin)
    out = H.conv_fwd(x, w, rb)
    src, dst = rb.pair_src[:P].long(), rb.pair_dst[:P].long()
    toff = rb.tile_off.long()
    tap = torch.zeros(P, dtype=torch.long, device=dev)
    ends = toff[:, -1]
    tap = torch.searchsorted(ends, torch.arange(P, device=dev), right=True)
    ref = torch.zeros(n, cout, dtype=torch.float64, device=dev)
    contrib = torch.einsum("pi,pio->po", x[src].double(), w.double()[tap])
    ref.index_add_(0, dst, contrib)
    e_f = float((out.double() - ref).abs().max() / ref.abs().max())
    g = torch.randn(n, cout, device=dev)
    din = H.conv_dgrad(g, w, rb, rb, True)
    refd = torch.zeros(n, cin, dtype=torch.float64, device=dev)
    refd.index_add_(0, src, torch.einsum("po,pio->pi", g[dst].double(), w.double()[tap]))
    e_d = float((din.double() - refd).abs().max() / refd.abs().max())
    print(f"rows {n:6d} {cin:3d}->{cout:3d}: fwd err {e_f:.2e}  dgrad err {e_d:.2e}")
