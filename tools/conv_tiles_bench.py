"""The conv forward kernels at the bench's level shapes (8 x 20k-point scenes, voxel 0.01): time per launch through the C-ABI
(events around back-to-back launches with packed weights and prebuilt rulebooks - no wrapper, no packing in the timed region),
useful TFLOP/s (2 P Cin Cout) against the fp32 MFMA peak, and how many MFMA row-slots a launch executes per useful pair.

  python tools/conv_tiles_bench.py            # masked-tile kernel, tile order on for >= GPN_TILE_ORDER_MIN_ROWS rows
  GPN_CONV_TILES=0 python tools/conv_tiles_bench.py   # the round-2 direct kernel on the same inputs

Every (level, channels) case also checks the output against the plain-table launch of the same kernel (bit-equal) and, in the
default mode, prints the largest difference to the other kernel's summation order when GPN_CONV_REF=path holds its outputs."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import _C, hip_ops as H
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud

dev = torch.device("cuda:0")
MFMA_PEAK = 157.3
if os.environ.get("GPN_PROBE_SO"):  # a measurement build of the library (tools/probes/tiles_ablation.sh)
    _C.SO_PATH = os.path.abspath(os.environ["GPN_PROBE_SO"])
L = _C.lib()


def timeit(fn, iters=int(os.environ.get('BENCH_ITERS', 40)), warm=int(os.environ.get('BENCH_WARM', 5))):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def popcount(t):
    t = t.to(torch.int64) & 0xFFFFFFFF
    c = torch.zeros_like(t)
    for k in range(27):
        c += (t >> k) & 1
    return c


def waste(rb, R):
    """MFMA row-slots executed per useful pair: tiles (R = 1) or pairs of tiles (R = 2) times their live taps times 16 rows"""
    table = (rb.nbr_p if rb.nbr_p is not None else rb.nbr)[:rb.K * rb.n_dst].view(rb.K, rb.n_dst) >= 0
    pad = (-rb.n_dst) % 16
    if pad:
        table = torch.cat([table, table.new_zeros(rb.K, pad)], 1)
    live = table.view(rb.K, -1, 16).any(2)  # [K, tiles]
    m = (live.to(torch.int64) << torch.arange(rb.K, device=live.device)[:, None]).sum(0)
    if R == 2:
        if m.numel() % 2:
            m = torch.cat([m, m.new_zeros(1)])
        m = m.view(-1, 2)
        m = (m[:, 0] | m[:, 1])
        return float(popcount(m).sum().item() * 32) / float(rb.num_pairs.item())
    return float(popcount(m).sum().item() * 16) / float(rb.num_pairs.item())


def conv_call(x, packed, rb, cin, cout, out, ordered=True):
    ws_ptr, ws_size, stream = H._fast_ws(dev)
    nbr_p = rb.nbr_p if ordered else None
    perm = rb.perm if ordered else None
    rc = L.gpn_spconv_fwd_ordered(H.ptr(x), H.ptr(packed), H.ptr(rb.nbr), H.ptr(nbr_p), H.ptr(perm), H.i32(rb.K),
                                  H.i64(rb.n_dst), H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(ws_ptr),
                                  ctypes.c_size_t(ws_size), ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def main():
    torch.manual_seed(0)
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    print(f"# GPN_CONV_TILES={os.environ.get('GPN_CONV_TILES', '1')} tile order from {H.TILE_ORDER_MIN_ROWS} rows, "
          f"blocks of {H.TILE_ORDER_BLOCK}")
    print(f"{'level rows':>14s} {'pairs/row':>9s} {'conv':>9s} {'us':>8s} {'TF':>7s} {'/mfma':>6s} {'slots/pair R1':>13s} {'R2':>6s}")
    ref_path = os.environ.get("GPN_CONV_REF")
    ref = torch.load(ref_path) if ref_path and os.path.exists(ref_path) else {}
    outs = {}
    levels = []
    for lvl in range(int(os.environ.get('BENCH_LEVELS', 5))):
        rb = H.rulebook_subm3(idx, shape)
        levels.append((idx, shape, rb))
        if lvl < 4 and lvl + 1 < int(os.environ.get('BENCH_LEVELS', 5)) + 1:
            idx2, shape2, rbd, rbu = H.rulebook_down(idx, shape, 8)
            levels[-1] = (idx, shape, rb, rbd, rbu)
            idx, shape = idx2, shape2
    for lvl, lv in enumerate(levels):
        rb = lv[2]
        n, P = rb.n_dst, int(rb.num_pairs.item())
        c = 16 * (lvl + 1)
        cases = [("subm", rb, c, c), ("subm", rb, 2 * c, c), ("subm", rb, c, 2 * c)]
        if len(lv) > 3:
            cases += [("down", lv[3], c, c + 16), ("inv", lv[4], c + 16, c)]
        for kind, r, cin, cout in cases:
            K = r.K
            pairs = int(r.num_pairs.item())
            x = torch.randn(r.n_src, cin, device=dev)
            w = torch.randn(K, cin, cout, device=dev) / (K * cin) ** 0.5
            packed = H.pack_weights(w, 0)
            out = torch.empty(r.n_dst, cout, device=dev)
            conv_call(x, packed, r, cin, cout, out)
            if r.nbr_p is not None and not os.environ.get('BENCH_NO_CHECK'):
                plain = torch.empty_like(out)
                conv_call(x, packed, r, cin, cout, plain, ordered=False)
                assert torch.equal(out, plain), (lvl, kind, cin, cout, (out - plain).abs().max().item())
            key = f"{lvl}/{kind}/{cin}/{cout}"
            outs[key] = out.cpu()
            extra = ""
            if key in ref:
                d = (outs[key] - ref[key]).abs().max().item()
                extra = f"  max|d| vs ref {d:.2e} (|out| max {ref[key].abs().max().item():.2f})"
            us = timeit(lambda: conv_call(x, packed, r, cin, cout, out))
            tf = 2.0 * pairs * cin * cout / us / 1e6
            print(f"L{lvl} {r.n_dst:10d} {pairs / r.n_dst:9.2f} {kind:>4s} {cin:3d}->{cout:<3d} {us:8.1f} {tf:7.2f} {tf / MFMA_PEAK:6.3f} "
                  f"{waste(r, 1):13.2f} {waste(r, 2):6.2f}{extra}")
    if ref_path and not ref:
        torch.save(outs, ref_path)


if __name__ == "__main__":
    main()
