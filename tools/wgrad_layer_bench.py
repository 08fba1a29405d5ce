"""Per-layer timing of the SubM weight-gradient contraction: the destination-stretch kernel against the pair-list kernel
(gpn_spconv_wgrad_stretch toggles them) on sheet-like voxel sets of the backbone's level sizes.  Times gpn_spconv_wgrad
(contraction + slice sums) with events on torch's stream, 30 calls after 5 warm-up calls."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gapartnet_amd import hip_ops as H, _C
from tests import synth


def main():
    cuda = torch.device("cuda:0")
    L = _C.lib()
    rng = np.random.default_rng(0)
    cases = [("L0 16->16", 8, [320, 320, 320], 30000, 16, 16), ("L1 32->32", 8, [160, 160, 160], 14000, 32, 32),
             ("L0 32->16", 8, [320, 320, 320], 30000, 32, 16), ("L1 16->32", 8, [160, 160, 160], 14000, 16, 32)]
    for name, batch, shape, n_per, cin, cout in cases:
        idx = synth.surface_indices(rng, batch, shape, n_per)
        N = idx.shape[0]
        rb = H.rulebook_subm3(torch.from_numpy(idx).to(cuda), shape)
        f = torch.randn(N, cin, device=cuda)
        g = torch.randn(N, cout, device=cuda)
        res = {}
        for on in (0, 1, 0, 1):
            L.gpn_spconv_wgrad_stretch(on)
            for _ in range(5):
                H.conv_wgrad(f, g, rb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                H.conv_wgrad(f, g, rb)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(on, []).append(e0.elapsed_time(e1) / 30 * 1e3)
        L.gpn_spconv_wgrad_stretch(1)
        pairs = int(rb.num_pairs.sum()) if hasattr(rb, "num_pairs") else -1
        print(f"{name}: rows {N} pairs {pairs}  pair-list {min(res[0]):.1f} us  stretch {min(res[1]):.1f} us", flush=True)


if __name__ == "__main__":
    main()
