#!/usr/bin/env python
"""Where a masked-tile conv launch spends its time, wave by wave (a GPN_TILES_TRACE build of the library:
tools/probes/tiles_ablation.sh "trace -DGPN_TILES_TRACE=1"): per wave its start, end of prologue, end of the tap loop, end
(s_memrealtime, 100 MHz), live taps and the SIMD it ran on.  Prints the launch's span, the phases, and the SIMD balance.

    GPN_PROBE_SO=tools/probes/_build/libgpn_trace.so python tools/probes/tiles_trace.py
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


def conv_call(x, packed, rb, cin, cout, out):
    ws_ptr, ws_size, stream = H._fast_ws(dev)
    rc = L.gpn_spconv_fwd_ordered(H.ptr(x), H.ptr(packed), H.ptr(rb.nbr), H.ptr(rb.nbr_p), H.ptr(rb.perm), H.i32(rb.K),
                                  H.i64(rb.n_dst), H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(ws_ptr),
                                  ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def report(name, tr):
    tr = tr[tr[:, 3] > 0]
    t0, t1, t2, t3, taps, hw, xcc = (tr[:, i].astype(np.int64) for i in range(7))
    base = t0.min()
    us = lambda t: (t - base) / 100.0
    span = us(t3).max()
    simd = (xcc & 0xf) * (1 << 20) + (hw & 0xff30)  # XCC_ID, HW_ID's se / sh / cu (15:8) and simd (5:4) fields
    ids, inv = np.unique(simd, return_inverse=True)
    busy = np.zeros(len(ids)); np.add.at(busy, inv, (t3 - t0) / 100.0)
    tap_sum = np.zeros(len(ids)); np.add.at(tap_sum, inv, taps)
    last = np.zeros(len(ids)); np.maximum.at(last, inv, us(t3))
    first = np.full(len(ids), 1e9); np.minimum.at(first, inv, us(t0))
    nwave = np.bincount(inv)
    print(f"== {name}: {len(tr)} waves on {len(ids)} SIMDs; launch span {span:.1f} us (first start -> last end)")
    print(f"   wave start: median {np.median(us(t0)):.1f} us, 90% {np.percentile(us(t0), 90):.1f}, max {us(t0).max():.1f}")
    print(f"   per wave (us): prologue {np.mean(t1 - t0) / 100:.2f}  tap loop {np.mean(t2 - t1) / 100:.2f}  epilogue {np.mean(t3 - t2) / 100:.2f}  "
          f"total mean {np.mean(t3 - t0) / 100:.2f} max {np.max(t3 - t0) / 100:.2f}")
    lt = (t2 - t1) / 100.0
    print(f"   tap loop per live tap: mean {np.sum(lt) / np.sum(taps):.3f} us;  live taps per wave mean {taps.mean():.1f} max {taps.max()}")
    print(f"   per SIMD: waves mean {nwave.mean():.1f} max {nwave.max()};  live taps mean {tap_sum.mean():.1f} max {tap_sum.max():.0f} "
          f"({tap_sum.max() / tap_sum.mean():.2f}x);  last wave ends: median {np.median(last):.1f} us, 10% {np.percentile(last, 10):.1f}, max {last.max():.1f}")
    mfma_us = tap_sum * float(os.environ.get("MFMA_PER_TAP", 16)) * 32 / 2400.0
    print(f"   per SIMD MFMA time at peak: mean {mfma_us.mean():.1f} us max {mfma_us.max():.1f};  MFMA share of the SIMD's busy span: mean {np.mean(mfma_us / (last - first)):.2f}")
    c = np.corrcoef(tap_sum, last - first)[0, 1]
    print(f"   correlation (SIMD's live taps, SIMD's busy span) {c:.2f}")
    xs = np.unique(xcc & 0xf)
    per_x = [us(t3)[(xcc & 0xf) == x].max() for x in xs]
    print("   last end per XCD (us):", " ".join(f"{v:.1f}" for v in per_x))
    # how many waves are alive over time
    ts = np.linspace(0, span, 9)[1:-1]
    alive = [(int(((us(t0) <= t) & (us(t3) > t)).sum())) for t in ts]
    print("   waves alive at", " ".join(f"{t:.0f}us:{a}" for t, a in zip(ts, alive)))


def main():
    torch.manual_seed(0)
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    for lvl in range(2):
        rb = H.rulebook_subm3(idx, shape)
        c = 16 * (lvl + 1)
        x = torch.randn(rb.n_src, c, device=dev)
        w = torch.randn(rb.K, c, c, device=dev) / (rb.K * c) ** 0.5
        packed = H.pack_weights(w, 0)
        out = torch.empty(rb.n_dst, c, device=dev)
        n_units = (rb.n_dst + 15) // 16 * 4  # generous: column groups <= 4
        trace = torch.zeros(n_units, 8, dtype=torch.int64, device=dev)
        for _ in range(5):
            conv_call(x, packed, rb, c, c, out)
        torch.cuda.synchronize()
        assert L.gpn_probe_tiles_trace(ctypes.c_void_p(trace.data_ptr())) == 0
        conv_call(x, packed, rb, c, c, out)
        torch.cuda.synchronize()
        L.gpn_probe_tiles_trace(ctypes.c_void_p(0))
        os.environ["MFMA_PER_TAP"] = str((c // 16) * (c // 16) * 4)
        report(f"L{lvl} {c}->{c}, {rb.n_dst} rows", trace.cpu().numpy())
        idx, shape, _, _ = H.rulebook_down(idx, shape, 8)


if __name__ == "__main__":
    main()
