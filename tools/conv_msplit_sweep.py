"""The conv launches of the levels below the masked-tile kernel's size (L2 .. L6 of 8 x 20k-point scenes, and the k = 8 stride
convs between them): the direct kernel's 4-way tap-split form (round 3) against the masked tap-split kernel (round 6,
csrc/spconv_msplit.hip) at every (column tiles per workgroup, waves per row tile) cut - us per launch through the C-ABI
(events around back-to-back launches; packed weights, prebuilt rulebooks), and the largest difference of the results.
Output committed as profiles/r06_conv_msplit_sweep.txt.

    python tools/conv_msplit_sweep.py [--cold]      (--cold: a 512 MB fill between launches, timed per launch)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import _C, hip_ops as H

if os.environ.get("GPN_PROBE_SO"):
    _C.SO_PATH = os.path.abspath(os.environ["GPN_PROBE_SO"])
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud

dev = torch.device("cuda:0")
L = _C.lib()
L.gpn_spconv_msplit.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
COLD = "--cold" in sys.argv


def timeit(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if COLD:
        junk = torch.empty(128 << 20, dtype=torch.float32, device=dev)
        tot = 0.0
        for _ in range(12):
            junk.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        return tot / 12 * 1e3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def conv_call(x, packed, tab, cin, cout, out):
    nbr, nbr_p, perm, K, n_dst = tab
    ws_ptr, ws_size, stream = H._fast_ws(dev)
    rc = L.gpn_spconv_fwd_ordered(H.ptr(x), H.ptr(packed), H.ptr(nbr), H.ptr(nbr_p), H.ptr(perm), H.i32(K), H.i64(n_dst),
                                  H.i32(cin), H.i32(cout), H.ptr(out), ctypes.c_void_p(ws_ptr), ctypes.c_size_t(ws_size),
                                  ctypes.c_void_p(stream))
    assert rc == 0, L.gpn_last_error()


def sweep(label, tab, n_src, cin, cout):
    K, n_dst = tab[3], tab[4]
    x = torch.randn(n_src, cin, device=dev)
    w = torch.randn(K, cin, cout, device=dev) / (K * cin) ** 0.5
    packed = H.pack_weights(w, 0)
    out = torch.empty(n_dst, cout, device=dev)
    L.gpn_spconv_msplit(0, 0, 0)
    t_old = timeit(lambda: conv_call(x, packed, tab, cin, cout, out))
    ref = out.clone()
    cells = []
    nt_total = cout // 16
    best = (1e9, None)
    for sp in (4, 9):
        if K < sp:
            continue
        for nt in (1, 2, 3, 4):
            if nt_total % nt or (sp == 9 and (cin // 16) * (1 + nt) > 28):
                continue
            L.gpn_spconv_msplit(1, nt, sp)
            out.zero_()
            t = timeit(lambda: conv_call(x, packed, tab, cin, cout, out))
            err = float((out - ref).abs().max())
            cells.append(f"nt{nt}/sp{sp} {t:6.1f}{'' if err == 0 else f' (d {err:.1e})'}")
            best = min(best, (t, f"nt{nt}/sp{sp}"))
    L.gpn_spconv_msplit(1, 0, 0)
    t_auto = timeit(lambda: conv_call(x, packed, tab, cin, cout, out))
    print(f"{label:>22s} {cin:4d}->{cout:<4d} K={K:<2d} direct-split {t_old:6.1f} | auto {t_auto:6.1f} | best {best[1]} {best[0]:6.1f} | " + "  ".join(cells), flush=True)


def main():
    torch.manual_seed(0)
    pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    print(f"# us per launch, {'cold (512 MB fill between launches)' if COLD else 'back to back'}; (d x) = largest difference from the direct kernel's result")
    for lvl in range(7):
        rb = H.rulebook_subm3(idx, shape)
        c = 16 * (lvl + 1)
        if lvl >= 2:
            tab = (rb.nbr, rb.nbr_p, rb.perm, rb.K, rb.n_dst)
            for cin, cout in ((c, c), (2 * c, c)):
                sweep(f"L{lvl} {rb.n_dst} rows", tab, rb.n_src, cin, cout)
        if lvl < 6:
            idx2, shape2, rb_f, rb_b = H.rulebook_down(idx, shape, 8)
            if lvl >= 1:
                sweep(f"L{lvl}->L{lvl + 1} down {rb_f.n_dst}", (rb_f.nbr, rb_f.nbr_p, rb_f.perm, rb_f.K, rb_f.n_dst), rb_f.n_src, c, c + 16)
                sweep(f"L{lvl + 1}->L{lvl} up {rb_b.n_dst}", (rb_b.nbr, rb_b.nbr_p, rb_b.perm, rb_b.K, rb_b.n_dst), rb_b.n_src, c + 16, c)
            idx, shape = idx2, shape2


if __name__ == "__main__":
    main()
