"""GPU parity of the masked tap-split conv kernel (csrc/spconv_msplit.hip, round 6): the layers below the masked-tile kernel's
size.  Against the oracle at north_star's 1e-4, and the properties its header states: a row's result depends neither on the
column tiles per workgroup nor on the tile order; with four waves per row tile it is bit-equal to the direct kernel's 4-way
form; every cut is deterministic; the device-counted form is bit-equal to the exactly-sized one under any plan."""
import ctypes

import numpy as np
import pytest
import torch

import oracle as O
from tests import synth
from tests.test_gpu_ops import CONV_SHAPES, FP_TOL, dev, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from gapartnet_amd import hip_ops
    return hip_ops


@pytest.fixture
def knob():
    """gpn_spconv_msplit(mode, force_nt, force_sp); restored to (on, table, table) afterwards; the masked-tile kernel kept away"""
    from gapartnet_amd import _C
    L = _C.lib()
    L.gpn_spconv_msplit.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    prev_tiles = L.gpn_spconv_tiles_min_tiles(1 << 40)
    yield L.gpn_spconv_msplit
    L.gpn_spconv_msplit(1, 0, 0)
    L.gpn_spconv_tiles_min_tiles(prev_tiles)
    L.gpn_spconv_direct_split(12000, 0)


def _cuts(cin, cout):
    nt_total = cout // 16
    for sp in (4, 9):
        for nt in (1, 2, 3, 4):
            if nt_total % nt == 0 and (sp == 4 or (cin // 16) * (1 + nt) <= 28):
                yield nt, sp


@pytest.mark.parametrize("cin,cout", CONV_SHAPES + [(224, 112)])
def test_masked_split_kernel_fwd_dgrad_every_cut(H, cuda, knob, cin, cout):
    from gapartnet_amd import _C
    rng = np.random.default_rng(cin * 1000 + cout + 13)
    shape = [40, 40, 40]
    idx = synth.surface_indices(rng, 2, shape, 1500)
    N = idx.shape[0]
    f = rng.normal(size=(N, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    g = rng.normal(size=(N, cout)).astype(np.float32)
    rb_ref = O.rulebook_subm3(idx, shape)
    rb = H.rulebook_subm3(dev(idx, cuda), shape)
    ref_out, ref_din = O.spconv_fwd(f, W, rb_ref, N), O.spconv_dgrad(g, W, rb_ref, N, N)
    fd, gd, Wd = dev(f, cuda), dev(g, cuda), dev(W, cuda)
    dgrad = H.PACK_TRANSPOSE | H.PACK_REVERSE
    # the direct kernel's 4-way form on the same inputs
    knob(0, 0, 0)
    _C.lib().gpn_spconv_direct_split(1 << 40, 0)
    direct_out, direct_din = H.conv_fwd_ordered(fd, Wd, rb), H.conv_fwd_ordered(gd, Wd, rb, flags=dgrad)
    by_sp = {}
    for nt, sp in _cuts(cin, cout):
        knob(1, nt, sp)
        out, din = H.conv_fwd_ordered(fd, Wd, rb), H.conv_fwd_ordered(gd, Wd, rb, flags=dgrad)
        assert np.allclose(host(out), ref_out, atol=FP_TOL, rtol=1e-4), (nt, sp)
        assert np.allclose(host(din), ref_din, atol=FP_TOL, rtol=1e-4), (nt, sp)
        assert torch.equal(out, H.conv_fwd_ordered(fd, Wd, rb)), "fixed summation order"
        if sp in by_sp:
            assert torch.equal(out, by_sp[sp][0]) and torch.equal(din, by_sp[sp][1]), "column tiles per workgroup must not change a bit"
        by_sp[sp] = (out, din)
    if cin // 16 in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12) and cout // 16 in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12):  # (widths the direct kernel is instantiated for)
        assert torch.equal(by_sp[4][0], direct_out) and torch.equal(by_sp[4][1], direct_din), "four waves per row tile = the direct kernel's 4-way form"
    # the built-in table, voxel order and tile order
    knob(1, 0, 0)
    a = H.conv_fwd_ordered(fd, Wd, rb)
    rb.perm, rb.nbr_p = H.tile_order(rb.nbr, 27, N)
    b = H.conv_fwd_ordered(fd, Wd, rb)
    assert torch.equal(a, b), "the tile order must not change a bit"
    assert any(torch.equal(a, v[0]) for v in by_sp.values())


def test_masked_split_kernel_down_inverse_ragged_tail_tiny(H, cuda, knob):
    """K = 8 tables (stride-2 conv and its inverse, rows without any tap), row counts that are not multiples of 16, a level of
    fewer than 16 row tiles (the lock-step kernel's until round 6)"""
    rng = np.random.default_rng(199)
    for shape, n in (([33, 40, 37], 4001), ([12, 10, 9], 150)):
        idx = synth.random_sparse_indices(rng, 2, shape, n)
        N = idx.shape[0]
        d = O.rulebook_down(idx, shape)
        No = d["out_indices"].shape[0]
        _, _, rb_f, rb_b = H.rulebook_down(dev(idx, cuda), shape, 2)
        for cin, cout in ((16, 32), (48, 64), (96, 112)):
            f = rng.normal(size=(N, cin)).astype(np.float32)
            W = (rng.normal(size=(8, cin, cout)) / np.sqrt(8 * cin)).astype(np.float32)
            Wi = (rng.normal(size=(8, cout, cin)) / np.sqrt(cout)).astype(np.float32)
            ref = O.spconv_fwd(f, W, d["fwd"], No)
            ref_up = O.spconv_fwd(ref, Wi, d["bwd"], N)
            for ordered in (False, True):
                for rb in (rb_f, rb_b):
                    rb.perm, rb.nbr_p = H.tile_order(rb.nbr, 8, rb.n_dst) if ordered else (None, None)
                for nt in (0, 1, 2):
                    knob(1, nt, 0)
                    out = host(H.conv_fwd_ordered(dev(f, cuda), dev(W, cuda), rb_f))
                    assert np.allclose(out, ref, atol=FP_TOL, rtol=1e-4)
                    up = host(H.conv_fwd_ordered(dev(ref, cuda), dev(Wi, cuda), rb_b))
                    assert np.allclose(up, ref_up, atol=FP_TOL, rtol=1e-4)
        # SubM on the same (ragged, tiny) rows
        rb_ref = O.rulebook_subm3(idx, shape)
        rb = H.rulebook_subm3(dev(idx, cuda), shape)
        for cin, cout in ((32, 32), (112, 112)):
            f = rng.normal(size=(N, cin)).astype(np.float32)
            W = (rng.normal(size=(27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
            knob(1, 0, 0)
            assert np.allclose(host(H.conv_fwd_ordered(dev(f, cuda), dev(W, cuda), rb)), O.spconv_fwd(f, W, rb_ref, N), atol=FP_TOL, rtol=1e-4)


def test_masked_split_kernel_in_the_executor_matches_the_direct_kernel(cuda, knob):
    """a small U-Net, training mode, through the network executor (BatchNorm sums in the conv epilogues, dgrad launches that carry
    the backward sums, accumulate-in-place second gradients, paired launches are covered by the model tests): features and every
    gradient with the masked tap-split kernel against the run with the direct kernel, at 1e-4 / 1e-3 of the tensor's scale - the
    two differ in the grouping of the taps' sums only"""
    from gapartnet_amd.network.backbone import SparseUNet
    from gapartnet_amd.spconv import pytorch as spconv
    import functools
    rng = np.random.default_rng(5)
    shape = [48, 48, 48]
    idx = synth.surface_indices(rng, 2, shape, 2500)
    N = idx.shape[0]
    torch.manual_seed(3)
    norm_fn = functools.partial(torch.nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    net = SparseUNet.build(16, [16, 32, 48, 64], 2, norm_fn, without_stem=True).to(cuda).train()
    x0 = torch.from_numpy(rng.normal(size=(N, 16)).astype(np.float32)).to(cuda)
    w_out = torch.from_numpy(rng.normal(size=(N, 16)).astype(np.float32)).to(cuda)

    def run():
        x = x0.clone().requires_grad_(True)
        st = spconv.SparseConvTensor(x, dev(idx, cuda), shape, 2)
        for p in net.parameters():
            p.grad = None
        y = net(st).features
        (y * w_out).sum().backward()
        return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()]

    knob(0, 0, 0)
    y0, dx0, g0 = run()
    knob(1, 0, 0)
    y1, dx1, g1 = run()
    scale = lambda t: float(t.abs().max()) + 1e-30
    assert float((y1 - y0).abs().max()) <= 1e-4 * scale(y0)
    assert float((dx1 - dx0).abs().max()) <= 1e-3 * scale(dx0)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 1e-3 * scale(b)
    y2, dx2, g2 = run()
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and all(torch.equal(a, b) for a, b in zip(g1, g2)), "deterministic"


def test_masked_split_kernel_at_the_bench_levels_full_size(H, cuda, knob):
    """the kernel on the levels it owns in the bench's batch (8 x 20k-point scenes: 25k rows x 48 channels, 7k x 64, 1.8k x 80, 487 x
    96, 107 x 112, and the stride-2 / inverse tables between them) against the oracle at north_star's 1e-4, forward and dgrad;
    the rulebooks of those levels bit-equal to the oracle's on the way"""
    from gapartnet_amd.smoke import make_batch
    from gapartnet_amd.structure.point_cloud import PointCloud
    rng = np.random.default_rng(4242)
    pcs = [pc.to(cuda) for pc in make_batch(8, 20000)]
    batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
    idx, shape = batch.voxel_tensor.indices, list(batch.voxel_tensor.spatial_shape)
    knob(1, 0, 0)
    dgrad = H.PACK_TRANSPOSE | H.PACK_REVERSE
    checked = 0
    for lvl in range(7):
        c = 16 * (lvl + 1)
        idx_h = host(idx)
        if lvl >= 2:
            rb = H.rulebook_subm3(idx, shape)
            rb_ref = O.rulebook_subm3(idx_h, shape)
            N = idx_h.shape[0]
            f = rng.normal(size=(N, c)).astype(np.float32)
            W = (rng.normal(size=(27, c, c)) / np.sqrt(27 * c)).astype(np.float32)
            out = H.conv_fwd_ordered(dev(f, cuda), dev(W, cuda), rb)
            assert np.allclose(host(out), O.spconv_fwd(f, W, rb_ref, N), atol=FP_TOL, rtol=1e-4), lvl
            din = H.conv_fwd_ordered(dev(f, cuda), dev(W, cuda), rb, flags=dgrad)
            assert np.allclose(host(din), O.spconv_dgrad(f, W, rb_ref, N, N), atol=FP_TOL, rtol=1e-4), lvl
            checked += 1
        if lvl < 6:
            d = O.rulebook_down(idx_h, shape)
            idx2, shape2, rb_f, rb_b = H.rulebook_down(idx, shape, 8)
            assert np.array_equal(host(idx2), d["out_indices"]), "coarse voxels: bit-exact"
            if lvl >= 1:
                No, N = idx2.shape[0], idx_h.shape[0]
                f = rng.normal(size=(N, c)).astype(np.float32)
                W8 = (rng.normal(size=(8, c, c + 16)) / np.sqrt(8 * c)).astype(np.float32)
                down = H.conv_fwd_ordered(dev(f, cuda), dev(W8, cuda), rb_f)
                ref = O.spconv_fwd(f, W8, d["fwd"], No)
                assert np.allclose(host(down), ref, atol=FP_TOL, rtol=1e-4), lvl
                Wi = (rng.normal(size=(8, c + 16, c)) / np.sqrt(c + 16)).astype(np.float32)
                up = H.conv_fwd_ordered(dev(ref, cuda), dev(Wi, cuda), rb_b)
                if N < 4096 * 16:  # (the inverse conv into a level of >= 4096 row tiles is the masked-tile kernel's)
                    checked += 1
                assert np.allclose(host(up), O.spconv_fwd(ref, Wi, d["bwd"], N), atol=FP_TOL, rtol=1e-4), lvl
            idx, shape = idx2, shape2
    assert checked >= 9
