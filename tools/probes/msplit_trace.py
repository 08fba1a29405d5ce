#!/usr/bin/env python
"""What a conv launch of the levels below the masked-tile kernel's size is made of, wave by wave: the direct kernel's 4-way
tap-split form (csrc/spconv_fwd.hip, GPN_SPLIT_TRACE) and the masked tap-split kernel (csrc/spconv_msplit.hip, GPN_MSPLIT_TRACE)
on the bench's levels L2 .. L5 (25k / 7k / 1.9k / 489 rows).  Every wave records s_memrealtime (100 MHz) at its start, when its
table entries (direct) / compacted offsets (masked) are there, when its first operands have arrived (masked only), at the end
of its stage loop, after the workgroup's barrier and at its end, plus HW_ID / XCC_ID.

    tools/probes/build_variants.sh "trace spconv_fwd.hip,spconv_msplit.hip -DGPN_SPLIT_TRACE=1 -DGPN_MSPLIT_TRACE=1"
    GPN_PROBE_SO=tools/probes/_build/libgpn_trace.so python tools/probes/msplit_trace.py [--cold]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gapartnet_amd import _C, hip_ops as H

_C.SO_PATH = os.path.abspath(os.environ.get("GPN_PROBE_SO", "tools/probes/_build/libgpn_trace.so"))
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud

dev = torch.device("cuda:0")
L = _C.lib()
COLD = "--cold" in sys.argv


def conv_call(x, packed, rb, cin, cout, out):
    ws_ptr, ws_size, stream = H._fast_ws(dev)
    rc = L.gpn_spconv_fwd_ordered(H.ptr(x), H.ptr(packed), H.ptr(rb.nbr), H.ptr(rb.nbr_p), H.ptr(rb.perm), H.i32(rb.K),
                                  H.i64(rb.n_dst), H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(ws_ptr),
                                  ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def report(name, tr, mfma_per_stage_clks):
    tr = tr[tr[:, 5] > 0]
    t0, t1, t1b, t2, t3, t4, taps, hw, xcc = (tr[:, i].astype(np.int64) for i in range(9))
    base = t0.min()
    us = lambda t: (t - base) / 100.0
    span = us(t4).max()
    cu = (xcc & 0xf) * (1 << 20) + (hw & 0xff00)
    print(f"== {name}: {len(tr)} waves on {len(np.unique(cu))} CUs; launch span {span:.1f} us (first wave start -> last wave end)")
    print(f"   wave start: median {np.median(us(t0)):.1f} us, 90% {np.percentile(us(t0), 90):.1f}, max {us(t0).max():.1f}")
    ph = lambda a, b: f"{np.mean(b - a) / 100:.2f} (max {np.max(b - a) / 100:.2f})"
    print(f"   per wave, us: table/offsets {ph(t0, t1)}  first operands {ph(t1, t1b)}  stage loop {ph(t1b, t2)}  barrier wait {ph(t2, t3)}  "
          f"sum+store+stats {ph(t3, t4)}  total {ph(t0, t4)}")
    print(f"   taps per wave (live, masked kernel; walked, direct kernel): mean {taps.mean():.1f} max {taps.max()};  "
          f"MFMA time of a wave's chain at the pipe's rate, per column tile it carries: mean {taps.mean() * mfma_per_stage_clks / 2400:.2f} us")
    ts = np.linspace(0, span, 9)[1:-1]
    alive = [(int(((us(t0) <= t) & (us(t4) > t)).sum())) for t in ts]
    print("   waves alive at", " ".join(f"{t:.0f}us:{a}" for t, a in zip(ts, alive)))


def main():
    torch.manual_seed(0)
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    junk = torch.empty(128 << 20, dtype=torch.float32, device=dev) if COLD else None
    print(f"# {'cold: a 512 MB fill before the traced launch' if COLD else 'warm: the traced launch follows five launches of the same layer'}")
    for lvl in range(6):
        rb = H.rulebook_subm3(idx, shape)
        c = 16 * (lvl + 1)
        if lvl >= 2:
            x = torch.randn(rb.n_src, c, device=dev)
            w = torch.randn(rb.K, c, c, device=dev) / (rb.K * c) ** 0.5
            packed = H.pack_weights(w, 0)
            out = torch.empty(rb.n_dst, c, device=dev)
            n_tiles = (rb.n_dst + 15) // 16
            for kind in ("direct 4-way", "masked tap-split"):
                L.gpn_spconv_msplit(0 if kind.startswith("direct") else 1, 0, 0)
                probe = L.gpn_probe_split_trace if kind.startswith("direct") else L.gpn_probe_msplit_trace
                trace = torch.zeros(n_tiles * (c // 16) * 16, 10, dtype=torch.int64, device=dev)
                for _ in range(5):
                    conv_call(x, packed, rb, c, c, out)
                torch.cuda.synchronize()
                if COLD:
                    junk.fill_(1.0)
                    torch.cuda.synchronize()
                assert probe(ctypes.c_void_p(trace.data_ptr())) == 0
                conv_call(x, packed, rb, c, c, out)
                torch.cuda.synchronize()
                probe(ctypes.c_void_p(0))
                report(f"L{lvl} {c}->{c}, {rb.n_dst} rows, {kind}", trace.cpu().numpy(), (c // 16) * 4 * 32)
        idx, shape, _, _ = H.rulebook_down(idx, shape, 8)


if __name__ == "__main__":
    main()
