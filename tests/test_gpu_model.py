"""GPU tests above the operator level: the whole pipeline on the HIP path vs the same pipeline over the CPU oracle
(small sizes), and size-independent properties at BASELINE.json's full sizes (20k-point scenes, batch 8; 50k-point
scene) where the oracle would be too slow."""
import copy

import numpy as np
import pytest
import torch

from gapartnet_amd import backend
from gapartnet_amd.smoke import make_batch, make_model
from tests.smoke_check import run_smoke
from gapartnet_amd.structure.point_cloud import PointCloud

pytestmark = pytest.mark.gpu


def test_smoke_train_step_matches_oracle(cuda):
    res = run_smoke(cuda)
    assert abs(res["loss"] - res["oracle_loss"]) < 2e-3 * max(1.0, abs(res["oracle_loss"]))


def test_backbone_features_within_1e4_of_oracle(cuda):
    """north_star tolerance: fp features within 1e-4 (eval mode, so BatchNorm uses fixed statistics)."""
    from oracle import torch_ops
    model = make_model((999, 999), channels=[16, 32, 48, 64]).eval()
    batch = make_batch(2, 3000)
    with torch.no_grad():
        with backend.using(torch_ops):
            ref_batch = PointCloud.collate(batch, voxel_size=(0.01, 0.01, 0.01))
            ref = model.forward_backbone(ref_batch)
        gpu_model = copy.deepcopy(model).to(cuda)
        gpu_batch = PointCloud.collate([pc.to(cuda) for pc in batch], voxel_size=(0.01, 0.01, 0.01))
        got = gpu_model.forward_backbone(gpu_batch)
    assert torch.equal(gpu_batch.voxel_tensor.indices.cpu(), ref_batch.voxel_tensor.indices), "voxel indices: bit-exact"
    assert torch.equal(gpu_batch.pc_voxel_id.cpu(), ref_batch.pc_voxel_id)
    assert torch.equal(gpu_batch.voxel_tensor.features.cpu(), ref_batch.voxel_tensor.features), "ordered means: bit-exact"
    err = (got.cpu() - ref).abs().max().item()
    assert err < 1e-4, err


def test_cluster_labels_bit_exact_vs_oracle(cuda):
    from gapartnet_amd.network import grouping_utils as G
    from oracle import torch_ops
    from tests import synth
    rng = np.random.default_rng(4)
    pts, batch = synth.clustered_points(rng, 4, 5000, n_clusters=10)
    offs = torch.tensor([0, 5000, 10000, 15000, 20000], dtype=torch.int32)
    sem = torch.from_numpy(rng.integers(1, 4, 20000).astype(np.int32))
    with backend.using(torch_ops):
        ref = G.cluster_proposals(torch.from_numpy(pts), torch.from_numpy(batch), offs, sem, 0.04, 50)
    got = G.cluster_proposals(torch.from_numpy(pts).to(cuda), torch.from_numpy(batch).to(cuda), offs.to(cuda), sem.to(cuda), 0.04, 50)
    assert torch.equal(got[0].cpu(), ref[0]) and torch.equal(got[1].cpu(), ref[1])


@pytest.mark.parametrize("n_scenes,n_points", [(8, 20000), (1, 50000)])
def test_full_size_properties(cuda, n_scenes, n_points):
    """BASELINE sizes: voxel/rulebook invariants that need no oracle."""
    from gapartnet_amd import hip_ops as H
    batch = [pc.to(cuda) for pc in make_batch(n_scenes, n_points)]
    data = PointCloud.collate(batch, voxel_size=(0.01, 0.01, 0.01))
    vt, pid = data.voxel_tensor, data.pc_voxel_id
    V = vt.indices.shape[0]
    assert pid.min() >= 0 and int(pid.max()) == V - 1 and torch.unique(pid).numel() == V  # every voxel has a point
    idx = vt.indices.long()
    key = ((idx[:, 0] * vt.spatial_shape[0] + idx[:, 1]) * vt.spatial_shape[1] + idx[:, 2]) * vt.spatial_shape[2] + idx[:, 3]
    assert bool((key[1:] > key[:-1]).all()), "voxels sorted by (scene,x,y,z), unique"
    # voxel feature = mean of its points: the count-weighted sum of voxel means reproduces the point sum
    counts = torch.bincount(pid.long(), minlength=V).float()
    assert torch.allclose((vt.features * counts[:, None]).sum(0), data.points.sum(0), rtol=1e-4, atol=1e-2)
    # point -> voxel map agrees with the coordinates
    order, starts = data.pc_voxel_csr
    assert int(starts[-1]) == data.points.shape[0] and torch.equal(pid[order.long()], torch.repeat_interleave(torch.arange(V, device=cuda), counts.long()).int())
    # SubM rulebook: centre tap is the identity, pair (i -> o) at tap k implies (o -> i) at tap 26 - k
    rb = H.rulebook_subm3(vt.indices, vt.spatial_shape)
    nbr = rb.nbr[:27 * V].reshape(27, V)
    assert torch.equal(nbr[13], torch.arange(V, device=cuda, dtype=torch.int32))
    k = 5
    o = torch.nonzero(nbr[k] >= 0)[:, 0]
    assert torch.equal(nbr[26 - k][nbr[k][o].long()], o.int())
    P = int(rb.num_pairs.item())
    assert P == int((nbr >= 0).sum()) and torch.equal(rb.tile_off[:, -1], torch.cumsum((nbr >= 0).sum(1), 0).int())
    # linearity of the conv kernel at full size: conv(a x + y) == a conv(x) + conv(y) up to fp32 rounding
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(V, 16, generator=g).to(cuda)
    y = torch.randn(V, 16, generator=g).to(cuda)
    W = (torch.randn(27, 16, 32, generator=g) / 20).to(cuda)
    lhs = H.conv_fwd(2.5 * x + y, W, rb)
    rhs = 2.5 * H.conv_fwd(x, W, rb) + H.conv_fwd(y, W, rb)
    assert torch.allclose(lhs, rhs, atol=1e-4, rtol=1e-4)
    # dgrad is the adjoint of fwd: <conv(x), g> == <x, dgrad(g)>
    gg = torch.randn(V, 32, generator=g).to(cuda)
    a = (H.conv_fwd(x, W, rb) * gg).sum().item()
    b = (x * H.conv_dgrad(gg, W, rb, rb, True)).sum().item()
    assert abs(a - b) <= 1e-3 * max(1.0, abs(a))
    # down conv: every fine voxel maps to exactly one coarse voxel; inverse restores the fine set
    out_idx, out_shape, rb_f, rb_b = H.rulebook_down(vt.indices, vt.spatial_shape, n_scenes)
    assert out_shape == [s // 2 for s in vt.spatial_shape]
    kept = ((idx[:, 1:] // 2) < torch.tensor(out_shape, device=cuda)).all(1)  # an odd extent drops its last plane
    Pd = int(rb_f.num_pairs.item())
    assert Pd == int(kept.sum()) and V - Pd < 0.01 * V
    assert torch.equal(torch.sort(rb_b.pair_dst[:Pd])[0], torch.nonzero(kept)[:, 0].int())
    assert int(torch.unique(rb_f.pair_dst[:Pd]).numel()) == out_idx.shape[0]


def test_full_size_train_step_runs_and_is_deterministic(cuda):
    """BASELINE config 3's per-GPU shape at FULL size: batch 8 x 20k-point scenes, every head on - finite loss, every
    parameter receives a finite gradient, two runs are bit-identical (losses and every gradient)"""
    batch = [pc.to(cuda) for pc in make_batch(8, 20000)]
    jitter = (torch.tensor([0.3, 0.6, 0.1], device=cuda), torch.tensor([0.5, 0.2, 0.9], device=cuda))
    losses, grads = [], []
    for _ in range(2):
        model = make_model((0, 0)).to(cuda)
        model.revoxelize_jitter = jitter
        loss = model.training_step(batch, 0)
        loss.backward()
        losses.append(float(loss))
        missing = [k for k, p in model.named_parameters() if p.grad is None]
        assert not missing, f"parameters without a gradient in the full pipeline: {missing[:5]}"
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    assert np.isfinite(losses[0]) and losses[0] == losses[1], losses
    for k in grads[0]:
        assert bool(torch.isfinite(grads[0][k]).all()), k
        assert torch.equal(grads[0][k], grads[1][k]), f"{k}: conv / BatchNorm sums / wgrad are order-independent: bitwise reproducible"


@pytest.mark.parametrize("epoch,expect", [(0, (False, False)), (5, (True, False)), (10, (True, True))])
def test_training_schedule_phases_on_the_full_model(cuda, epoch, expect):
    """gapartnet.yaml's training_schedule [5, 10] on the full 7-level model (network/model.py:466-659 of the reference):
    epoch < 5 trains backbone + point heads only, 5-9 adds ScoreNet, >= 10 adds NPCS-Net - which sub-networks receive
    gradients and which loss terms are logged"""
    score_on, npcs_on = expect
    model = make_model((5, 10)).to(cuda)
    model._current_epoch = epoch
    model.revoxelize_jitter = (torch.tensor([0.3, 0.6, 0.1], device=cuda), torch.tensor([0.5, 0.2, 0.9], device=cuda))
    logged = {}
    model._log_sink = lambda name, value, bs, sync: logged.setdefault(name, float(value))
    batch = [pc.to(cuda) for pc in make_batch(4, 20000)]
    loss = model.training_step(batch, 0)
    loss.backward()
    assert np.isfinite(float(loss))

    def has_grad(prefix):
        ps = [p for k, p in model.named_parameters() if k.startswith(prefix)]
        assert ps, prefix
        got = [p.grad is not None and bool((p.grad != 0).any()) for p in ps]
        assert all(got) or not any(got), f"{prefix}: some parameters trained, some not"
        return all(got)

    assert has_grad("backbone.") and has_grad("sem_seg_head.") and has_grad("offset_head.")
    assert has_grad("score_unet.") == score_on and has_grad("score_head.") == score_on
    assert has_grad("npcs_unet.") == npcs_on and has_grad("npcs_head.") == npcs_on
    # the reference logs all five loss terms in every phase (a switched-off term is the constant 0.0), model.py:572-590 there
    want = {f"train_loss/{k}" for k in ("total_loss", "loss_sem_seg", "loss_offset_dist", "loss_offset_dir", "loss_prop_score",
                                        "loss_prop_npcs")} | {"train/all_accu", "train/pixel_accu"}
    assert want <= set(logged), sorted(logged)
    assert (logged["train_loss/loss_prop_score"] > 0) == score_on, logged
    assert (logged["train_loss/loss_prop_npcs"] > 0) == npcs_on, logged
    assert logged["train_loss/loss_sem_seg"] > 0 and logged["train_loss/loss_offset_dist"] > 0
    assert all(np.isfinite(v) for v in logged.values()), logged


def test_executor_gradient_hand_over_contract(cuda):
    """network/net_exec.py hands U-Net parameter gradients over directly (persistent flat buffer, ``.grad`` assigned, no
    AccumulateGrad nodes).  What that must not break: a gradient somebody keeps is not overwritten by the next backward, a
    frozen parameter gets none, a tensor hook fires (the call switches to the autograd form), and with
    ``set_autograd_parameters(True)`` ``torch.autograd.grad`` works and agrees."""
    from gapartnet_amd.network import net_exec
    net, idx, feats, spconv = _unet_case(cuda, True)
    net.train(True)
    names = [k for k, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]

    def run(scale=1.0, retain=False):
        f = feats.clone().requires_grad_(True)
        y = net(spconv.SparseConvTensor(f, idx, [64, 64, 64], 3)).features
        loss = (y.square().sum()) * scale
        return loss, f

    loss, _ = run()
    loss.backward()
    ref = {k: p.grad.clone() for k, p in zip(names, params)}
    # the consumer released the gradients (what FusedAdam.step / the Trainer do): the next backward reuses the persistent buffer
    # (no allocation, same addresses)
    first_ptrs = [p.grad.data_ptr() for p in params]
    net.zero_grad(set_to_none=True)
    net_exec.release_gradients(net)
    loss, _ = run()
    loss.backward()
    assert [p.grad.data_ptr() for p in params] == first_ptrs, "the persistent gradient buffer must be reused step after step"
    kept = params[0].grad                       # nobody released this pass's gradients (a foreign training loop): they stay
    kept_copy = kept.clone()
    net.zero_grad(set_to_none=True)
    loss, _ = run(scale=3.0)
    loss.backward()
    assert torch.equal(kept, kept_copy), "an unreleased gradient must survive the next backward"
    assert params[0].grad is not kept and torch.allclose(params[0].grad, 3.0 * kept_copy, rtol=1e-4, atol=1e-6)
    # frozen parameter
    net.zero_grad(set_to_none=True)
    params[0].requires_grad_(False)
    loss, _ = run()
    loss.backward()
    assert params[0].grad is None and all(p.grad is not None for p in params[1:])
    assert torch.allclose(params[1].grad, ref[names[1]], rtol=1e-4, atol=1e-5 * float(ref[names[1]].abs().max()))
    params[0].requires_grad_(True)
    # a hook switches the call to the autograd form and fires
    net.zero_grad(set_to_none=True)
    fired = []
    h = params[2].register_hook(lambda g: fired.append(g.shape))
    loss, _ = run()
    loss.backward()
    h.remove()
    assert fired == [params[2].shape]
    for k, p in zip(names, params):
        assert torch.allclose(p.grad, ref[k], rtol=1e-4, atol=1e-5 * float(ref[k].abs().max()) + 1e-7), k
    # autograd form on request: torch.autograd.grad on U-Net parameters
    net.zero_grad(set_to_none=True)
    prev = net_exec.set_autograd_parameters(True)
    try:
        loss, _ = run()
        got = torch.autograd.grad(loss, params)
    finally:
        net_exec.set_autograd_parameters(prev)
    assert all(p.grad is None for p in params), "autograd.grad must not populate .grad"
    for k, g in zip(names, got):
        assert torch.allclose(g, ref[k], rtol=1e-4, atol=1e-5 * float(ref[k].abs().max()) + 1e-7), k
    # ... and in the default form the same request fails loudly (documented contract), it does not return garbage
    loss, _ = run()
    with pytest.raises(RuntimeError):
        torch.autograd.grad(loss, params[:1])


def _unet_case(cuda, without_stem):
    import functools
    import torch.nn as nn
    from gapartnet_amd.network.backbone import SparseUNet
    from gapartnet_amd.spconv import pytorch as spconv
    from tests import synth
    norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    torch.manual_seed(3)
    cin = 16 if without_stem else 6
    net = SparseUNet.build(cin, [16, 32, 48, 64], 2, norm_fn, without_stem=without_stem).to(cuda)
    rng = np.random.default_rng(7)
    idx = torch.from_numpy(synth.surface_indices(rng, 3, [64, 64, 64], 2500)).to(cuda)
    feats = torch.from_numpy(rng.normal(size=(idx.shape[0], cin)).astype(np.float32)).to(cuda)
    return net, idx, feats, spconv


@pytest.fixture
def bn_fusion(request):
    """conv-epilogue BatchNorm sums (csrc/bn_stats.h) on / off for one test"""
    from gapartnet_amd import _C
    prev = _C.lib().gpn_net_bn_fusion(1 if request.param else 0)
    yield bool(request.param)
    _C.lib().gpn_net_bn_fusion(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("without_stem", [False, True])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("bn_fusion", [False, True], indirect=True)
def test_native_executor_matches_per_layer_path(cuda, without_stem, training, bn_fusion):
    """kernel family U: one-call forward/backward of the whole U-Net == the module-by-module walk (same kernels).  With the
    BatchNorm sums taken in the conv epilogues (fixed-point accumulation, bn_stats.h) the training-mode statistics differ
    from the separate statistics pass in the last bits: equal to 1e-5 instead of bit-equal."""
    from gapartnet_amd.network import net_exec
    net, idx, feats, spconv = _unet_case(cuda, without_stem)
    exact = not (bn_fusion and training)
    assert net_exec.program_for(net) is not None, "the reference U-Net must be expressible as a program"
    ref = copy.deepcopy(net)
    ref.use_native_executor = False
    net.train(training), ref.train(training)
    outs, grads, in_grads = [], [], []
    for model in (net, ref):
        f = feats.clone().requires_grad_(True)
        x = spconv.SparseConvTensor(f, idx, [64, 64, 64], 3)
        y = model(x)
        w = torch.linspace(-1, 1, y.features.numel(), device=cuda).view_as(y.features)
        (y.features * w).sum().backward()
        outs.append(y.features.detach())
        in_grads.append(f.grad)
        grads.append({k: p.grad for k, p in model.named_parameters()})
        assert torch.equal(y.indices, idx)
    if exact:
        assert torch.equal(outs[0], outs[1]), "forward: identical kernels in identical order -> bit-equal"
    else:
        assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-5), (outs[0] - outs[1]).abs().max()
    scale = float(in_grads[1].abs().max())
    assert torch.allclose(in_grads[0], in_grads[1], rtol=1e-5, atol=1e-6 if exact else 1e-5 * scale)
    for k in grads[1]:
        assert grads[0][k] is not None, k
        gs = float(grads[1][k].abs().max())
        assert torch.allclose(grads[0][k], grads[1][k], rtol=1e-4, atol=1e-6 if exact else 1e-4 * gs + 1e-6), k
    for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        if exact:
            assert torch.equal(a, b), f"buffer / parameter {k} diverged (running statistics, num_batches_tracked)"
        else:
            assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("bn_fusion", [False, True], indirect=True)
@pytest.mark.parametrize("one_sided", [False, True])
def test_paired_passes_equal_one_network_after_the_other(cuda, training, bn_fusion, one_sided):
    """net_exec.run_pair (gpn_net_forward_pair / gpn_net_backward_pair: layer i of two structurally identical U-Nets in ONE
    launch per kernel, as ScoreNet + NPCS-Net run in the model): outputs, input gradient, every parameter gradient and every
    buffer are BIT-equal to running the two networks one after the other (same kernels, same summation orders; the
    BatchNorm sums are order-independent integers).  ``one_sided``: only one output reaches the loss - that network
    alone runs backward, the other's parameters keep ``grad is None``."""
    from gapartnet_amd.network import net_exec
    net_a, idx, feats, spconv = _unet_case(cuda, True)
    torch.manual_seed(11)
    net_b = copy.deepcopy(net_a)
    with torch.no_grad():
        for p in net_b.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ref_a, ref_b = copy.deepcopy(net_a), copy.deepcopy(net_b)
    for m in (net_a, net_b, ref_a, ref_b):
        m.train(training)
    results = []
    for paired in (True, False):
        a, b = (net_a, net_b) if paired else (ref_a, ref_b)
        f = feats.clone().requires_grad_(True)
        x = spconv.SparseConvTensor(f, idx, [64, 64, 64], 3)
        if paired:
            out = net_exec.run_pair(a, b, x)
            assert out is not None, "two copies of one architecture must pair"
            ya, yb = out
        else:
            ya, yb = a(x), b(x)
        wa = torch.linspace(-1, 1, ya.features.numel(), device=cuda).view_as(ya.features)
        wb = torch.linspace(2, -1, yb.features.numel(), device=cuda).view_as(yb.features)
        loss = (ya.features * wa).sum() if one_sided else (ya.features * wa).sum() + (yb.features * wb).sum()
        loss.backward()
        results.append((ya.features.detach(), yb.features.detach(), f.grad,
                        {k: p.grad for k, p in a.named_parameters()}, {k: p.grad for k, p in b.named_parameters()},
                        a.state_dict(), b.state_dict()))
    (ya, yb, din, ga, gb, sa, sb), (ra, rb, rdin, rga, rgb, rsa, rsb) = results
    assert torch.equal(ya, ra) and torch.equal(yb, rb)
    assert torch.equal(din, rdin)
    for k in rga:
        assert ga[k] is not None and torch.equal(ga[k], rga[k]), k
    for k in rgb:
        if one_sided:
            assert gb[k] is None and rgb[k] is None, k
        else:
            assert gb[k] is not None and torch.equal(gb[k], rgb[k]), k
    for got, want in ((sa, rsa), (sb, rsb)):
        for k in want:
            assert torch.equal(got[k], want[k]), f"buffer / parameter {k} diverged"
    # different architectures do not pair
    other, _, _, _ = _unet_case(cuda, False)
    assert net_exec.run_pair(net_a, other, spconv.SparseConvTensor(feats, idx, [64, 64, 64], 3)) is None


@pytest.mark.gpu
def test_conv_epilogue_batchnorm_sums_are_deterministic_and_used(cuda):
    """the fused form (BatchNorm sums accumulated by the producing conv launch as fixed-point integers): two runs are
    bit-identical although thousands of waves add to one channel in arbitrary order, and the profile shows no separate
    statistics launch for the levels above 1024 rows"""
    from gapartnet_amd import _C
    assert _C.lib().gpn_net_bn_fusion(-1) == 1, "fusion is the default"
    net, idx, feats, spconv = _unet_case(cuda, True)
    net.train(True)
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        f = feats.clone().requires_grad_(True)
        y = net(spconv.SparseConvTensor(f, idx, [64, 64, 64], 3))
        y.features.square().sum().backward()
        runs.append((y.features.detach().clone(), f.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][2], runs[1][2]))
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        f = feats.clone().requires_grad_(True)
        net(spconv.SparseConvTensor(f, idx, [64, 64, 64], 3)).features.square().sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    reduces = sum(e.count for e in prof.key_averages() if "bn_reduce_kernel" in e.key)
    applies = sum(e.count for e in prof.key_averages() if "bn_apply_fwd_fold_kernel" in e.key)
    assert applies > 0, names
    # only BatchNorms behind a k = 1 conv (lock-step kernel, no sum epilogue) keep their statistics launches
    assert reduces <= applies, (reduces, applies, names)


@pytest.mark.gpu
def test_native_executor_no_grad_and_frozen_input(cuda):
    from gapartnet_amd.network import net_exec
    net, idx, feats, spconv = _unet_case(cuda, True)
    ref = copy.deepcopy(net)
    ref.use_native_executor = False
    with torch.no_grad():
        a = net(spconv.SparseConvTensor(feats, idx, [64, 64, 64], 3)).features
        b = ref(spconv.SparseConvTensor(feats, idx, [64, 64, 64], 3)).features
    # (training-mode BatchNorm: the executor takes the statistics from the conv epilogues, csrc/bn_stats.h - equal to
    # rounding, bit-equal with gpn_net_bn_fusion(0): test_native_executor_matches_per_layer_path)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    # input that does not require grad: parameters still get their gradients
    net(spconv.SparseConvTensor(feats, idx, [64, 64, 64], 3)).features.square().sum().backward()
    ref(spconv.SparseConvTensor(feats, idx, [64, 64, 64], 3)).features.square().sum().backward()
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-4 * float(q.grad.abs().max()) + 1e-6), k


@pytest.mark.gpu
def test_device_prefetcher_feeds_identical_steps(cuda):
    """batches prepared one step ahead on a second stream (dataset/prefetch.py) train exactly like batches collated
    inside the step: same losses, same parameters after 3 steps"""
    from gapartnet_amd.dataset.prefetch import DevicePrefetcher
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    scenes = [[pc.to(cuda) for pc in make_batch(2, 4000, seed0=50 + 10 * j)] for j in range(3)]
    results = []
    for use_prefetch in (False, True):
        model = copy.deepcopy(base)
        model.revoxelize_jitter = (torch.full((3,), 0.3, device=cuda), torch.full((3,), 0.6, device=cuda))
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        feed = DevicePrefetcher(scenes, model, cuda) if use_prefetch else scenes
        losses = []
        for i, batch in enumerate(feed):
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(batch, i)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        results.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    assert torch.equal(results[0][0], results[1][0]), (results[0][0], results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_device_prefetcher_iterated_twice_without_a_host_sync(cuda):
    """the SAME DevicePrefetcher object walked for two epochs back to back (no synchronisation in between): the second pass's
    first preparation must wait for the training stream as it stands then, not for the stale mark of the first pass - the
    side stream would otherwise reuse blocks the last step's kernels still read.  Same losses / parameters as the plain feed."""
    from gapartnet_amd.dataset.prefetch import DevicePrefetcher
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    scenes = [[pc.to(cuda) for pc in make_batch(2, 4000, seed0=70 + 10 * j)] for j in range(3)]
    results = []
    for use_prefetch in (False, True):
        model = copy.deepcopy(base)
        model.revoxelize_jitter = (torch.full((3,), 0.3, device=cuda), torch.full((3,), 0.6, device=cuda))
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        feed = DevicePrefetcher(scenes, model, cuda) if use_prefetch else scenes
        losses = []
        for epoch in range(2):
            for i, batch in enumerate(feed):
                opt.zero_grad(set_to_none=True)
                loss = model.training_step(batch, i)
                loss.backward()
                opt.step()
                losses.append(loss.detach())
        torch.cuda.synchronize()
        results.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    assert torch.equal(results[0][0], results[1][0]), (results[0][0], results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_rank_bench_shares_one_device(cuda, launcher):
    """bench.py's multi-rank path (GradSync over the executor's flat gradient buffers, device prefetcher, per-rank scene shards,
    max-over-ranks timing) with two ranks time-slicing ONE GPU over gloo (GPN_DIST_SHARE_DEVICE): the code path the driver
    runs at N = 2/4/8 over RCCL, minus the performance.  "self": the plain command `python bench.py --gpus 2` - no RANK in the
    environment, the script starts its ranks itself (round 6; before, that command died on an assertion); "torchrun": wrapped in
    `python -m torch.distributed.run` by the caller, as the driver's contract says."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPN_DIST_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    bench = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--points", "6000",
             "--no-cpu-baseline"]
    if launcher == "self":
        cmd = [sys.executable] + bench
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29533"] + bench
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 4
    assert res["distributed"]["world_size"] == 2
    ex = res["grad_exchange"]  # 4 steps x (3 U-Nets reduced in place in the executor's buffer + the heads via one cat)
    assert ex["steps"] == 4 and ex["in_place"] >= ex["steps"] and ex["flattened"] >= ex["steps"], ex


@pytest.mark.gpu
def test_bench_with_the_gradient_exchange_forced_on_one_rank(cuda):
    """GPN_BENCH_FORCE_GRAD_SYNC=1 (one of the three environment switches left, with GPN_DIST_SHARE_DEVICE and GPN_NO_PIN): bench.py
    brings up a ONE-rank RCCL group and runs the gradient exchange of the multi-GPU path on it - the executor's flat gradient
    buffers all-reduced in place by the collective library the 8-GPU run uses"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPN_BENCH_FORCE_GRAD_SYNC="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "2", "--points", "6000", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["distributed"]["world_size"] == 1 and res["distributed"]["backend"] == "nccl"
    ex = res["grad_exchange"]
    assert ex["backend"] == "nccl" and ex["steps"] == 5 and ex["in_place"] >= ex["steps"], ex


@pytest.mark.gpu
def test_training_step_without_any_proposal(cuda):
    """every point predicted as background: clustering has nothing to cluster, the proposal nets do not run, the step
    still produces a finite loss and gradients for the backbone and the point heads only"""
    model = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    with torch.no_grad():
        model.sem_seg_head.weight.zero_()
        model.sem_seg_head.bias.fill_(-10.0)
        model.sem_seg_head.bias[0] = 10.0
    batch = [pc.to(cuda) for pc in make_batch(2, 4000)]
    loss = model.training_step(batch, 0)
    assert torch.isfinite(loss)
    loss.backward()
    assert model.backbone.stem[0].weight.grad is not None and torch.isfinite(model.backbone.stem[0].weight.grad).all()
    assert all(p.grad is None for p in model.score_unet.parameters())
    assert all(p.grad is None for p in model.npcs_unet.parameters())


@pytest.mark.gpu
def test_validation_epoch_on_the_gpu_matches_the_oracle_path(cuda):
    """the eval path (validation_step -> filter / NMS -> AP, mIoU at epoch end) on the GPU gives the metrics the same
    model gives over the CPU oracle (same proposals: clustering is bit-exact; scores within fp tolerance)"""
    from oracle import torch_ops
    from gapartnet_amd.trainer import MetricLog
    model = make_model((0, 0), channels=[16, 32, 48]).eval()
    model.revoxelize_jitter = (torch.full((3,), 0.3), torch.full((3,), 0.6))
    results = []
    for on_gpu in (False, True):
        m = copy.deepcopy(model)
        if on_gpu:
            m = m.to(cuda)
            m.revoxelize_jitter = tuple(t.to(cuda) for t in model.revoxelize_jitter)
        log = MetricLog()
        m._log_sink = log
        with torch.no_grad():
            for loader_idx in range(3):
                batch = make_batch(2, 2500, seed0=2000 + 10 * loader_idx)
                if on_gpu:
                    m.validation_step([pc.to(cuda) for pc in batch], 0, loader_idx)
                else:
                    with backend.using(torch_ops):
                        m.validation_step(batch, 0, loader_idx)
            if on_gpu:
                m.on_validation_epoch_end()
            else:
                with backend.using(torch_ops):
                    m.on_validation_epoch_end()
        results.append(log.reduce(cuda if on_gpu else torch.device("cpu")))
    cpu, gpu = results
    assert set(cpu) == set(gpu)
    for key in cpu:
        assert np.isfinite(gpu[key]), key
        assert abs(cpu[key] - gpu[key]) <= 1e-3 * max(1.0, abs(cpu[key])), (key, cpu[key], gpu[key])


@pytest.mark.gpu
@pytest.mark.parametrize("augment", [False, True])
def test_prefetcher_prepares_raw_scenes_on_the_gpu(cuda, augment):
    """a device_pipeline dataset hands over raw scenes; label compaction, augmentation, per-instance
    statistics and voxelisation run per batch on the prefetch stream.  Checked against the per-scene CPU loader functions
    (dataset/gapartnet.py, the restatement of the reference loader), then trained on for one step."""
    from gapartnet_amd.dataset import gapartnet as ds
    from gapartnet_amd.dataset import synthetic
    from gapartnet_amd.dataset.prefetch import DevicePrefetcher
    aug = dict(pos_jitter=0.05, color_jitter=0.1, flip_prob=0.5, rotate_prob=0.5) if augment else None
    scenes = [synthetic.make_scene(5100 + i, 6000) for i in range(3)]
    np.random.seed(5)
    want = []
    for pc in scenes:
        pc = ds.compact_instance_labels(pc)
        if aug:
            pc = ds.apply_augmentations(pc, **aug)
        want.append(ds.generate_inst_info(pc).to_tensor())
    model = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    np.random.seed(5)
    feed = iter(DevicePrefetcher([[pc.to_tensor() for pc in scenes]], model, cuda, augmentation=aug))
    batch = next(feed)
    torch.cuda.synchronize()
    assert batch.points.is_cuda and batch.num_instances == [pc.num_instances for pc in want]
    assert torch.equal(batch.instance_labels.cpu(), torch.cat([pc.instance_labels for pc in want]))
    assert torch.allclose(batch.points.cpu(), torch.cat([pc.points for pc in want]), rtol=0, atol=2e-6)
    assert torch.allclose(batch.instance_regions.cpu(), torch.cat([pc.instance_regions for pc in want]), rtol=0, atol=3e-6)
    for s, pc in enumerate(want):
        assert torch.equal(batch.num_points_per_instance[s, :pc.num_instances].cpu(), pc.num_points_per_instance)
        assert torch.equal(batch.instance_sem_labels[s, :pc.num_instances].cpu(), pc.instance_sem_labels)
    if not augment:  # identical coordinates => identical voxels as the per-scene-prepared batch
        ref = PointCloud.collate([pc.to(cuda) for pc in want], voxel_size=model.voxel_size)
        assert torch.equal(batch.voxel_tensor.indices, ref.voxel_tensor.indices)
        assert torch.equal(batch.pc_voxel_id, ref.pc_voxel_id)
    loss = model.training_step(batch, 0)
    loss.backward()
    assert torch.isfinite(loss)


@pytest.mark.gpu
def test_fused_adam_follows_moved_storage_and_changed_switches(cuda):
    """the device tables hold raw addresses of value / gradient / moments: replacing a parameter's storage (``p.data = ...``
    as ``model.float()`` / ``load_state_dict(assign=True)`` do), replacing a moment tensor, clearing the state of one
    parameter and switching weight decay on mid-run must all be picked up (the step signature covers them) - same
    trajectory as torch.optim.Adam, no write through a stale pointer"""
    from gapartnet_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(16, 27, 16), (32,), (4096, 3)]
    base = [torch.randn(s, generator=g).to(cuda) for s in shapes]
    a = [torch.nn.Parameter(t.clone()) for t in base]
    b = [torch.nn.Parameter(t.clone()) for t in base]
    opt_a = FusedAdam(a, lr=1e-2)
    opt_b = torch.optim.Adam(b, lr=1e-2, foreach=False, fused=False)
    grad_bufs = [torch.empty_like(p) for p in a]  # persistent gradient buffers, like the executor's

    def both_step():
        for p, q, buf in zip(a, b, grad_bufs):
            buf.copy_(torch.randn(p.shape, generator=g).to(cuda))
            p.grad, q.grad = buf, buf.clone()
        opt_a.step()
        opt_b.step()

    both_step()
    both_step()
    old_storage = a[0].data
    a[0].data = a[0].data.clone()                                   # the parameter's storage moves
    both_step()
    assert torch.equal(old_storage, old_storage.clone()) and not torch.equal(old_storage, a[0].data), "old storage untouched"
    opt_a.state[a[2]]["exp_avg"] = opt_a.state[a[2]]["exp_avg"].clone()  # a moment tensor is replaced
    both_step()
    for group in (opt_a.param_groups[0], opt_b.param_groups[0]):
        group["weight_decay"] = 0.01                                # needs torch's path from now on
    both_step()
    both_step()
    opt_a.state_dict()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-6), float((p - q).abs().max())
        assert float(opt_a.state[p]["step"]) == float(opt_b.state[q]["step"]) == 6.0
        assert torch.allclose(opt_a.state[p]["exp_avg"], opt_b.state[q]["exp_avg"], rtol=1e-5, atol=1e-7)
    # a checkpoint of torch's fused Adam (fused=True, device step tensors) loads and keeps training on either path
    opt_f = torch.optim.Adam([torch.nn.Parameter(t.clone()) for t in base], lr=1e-2, fused=True)
    for p in opt_f.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    opt_f.step()
    opt_d = FusedAdam([torch.nn.Parameter(t.clone()) for t in base], lr=1e-2)
    opt_d.load_state_dict(opt_f.state_dict())
    assert not opt_d.param_groups[0].get("fused") and not opt_d.param_groups[0].get("capturable")
    for p in opt_d.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    opt_d.step()
    opt_d.param_groups[0]["weight_decay"] = 0.1
    opt_d.step()  # torch's single-tensor path with the loaded (now CPU) step tensors


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam(cuda):
    """gapartnet_amd.optim.FusedAdam (one launch for all tensors) against torch.optim.Adam's single-tensor implementation:
    same parameters and state after four steps, including a tensor that receives its first gradient two steps late (the
    training schedule switches ScoreNet / NPCS-Net on at epochs 5 / 10) and a gradient buffer that moves."""
    from gapartnet_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(16, 27, 6), (16,), (48, 27, 48), (5000, 3), (1,), (112, 27, 112)]
    base = [torch.randn(s, generator=g).to(cuda) for s in shapes]
    a = [torch.nn.Parameter(t.clone()) for t in base]
    b = [torch.nn.Parameter(t.clone()) for t in base]
    opt_a = FusedAdam(a, lr=1e-3)
    opt_b = torch.optim.Adam(b, lr=1e-3, foreach=False, fused=False)
    assert isinstance(opt_a, torch.optim.Adam)
    for step in range(4):
        for i, (p, q) in enumerate(zip(a, b)):
            if i == 3 and step < 2:
                p.grad = q.grad = None
                continue
            grad = torch.randn(p.shape, generator=g).to(cuda)
            p.grad, q.grad = grad.clone(), grad.clone()
        opt_a.step()
        opt_b.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=2e-7, atol=1e-7), float((p - q).abs().max())  # an ulp: torch contracts a*b+c
        opt_a.state_dict()  # refreshes the per-parameter step tensors
        sa, sb = opt_a.state[p], opt_b.state[q]
        assert float(sa["step"]) == float(sb["step"])
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-6, atol=1e-7)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-6, atol=1e-9)
    # state_dict round trip into a plain torch Adam
    opt_c = torch.optim.Adam([torch.nn.Parameter(t.clone()) for t in base], lr=1e-3)
    opt_c.load_state_dict(opt_a.state_dict())


@pytest.mark.gpu
def test_fused_adam_tensors_that_sit_out_steps_keep_their_own_step_count(cuda):
    """(round 5) batches without proposals alternate with batches that have some: ScoreNet / NPCS-Net parameters then have no
    gradient on some steps and torch.optim.Adam leaves their step counts behind the backbone's.  FusedAdam caches its device
    tables by signature (which tensors have a gradient) - a table set cached BEFORE a tensor sat out a step groups it with the
    backbone's step number and must be regrouped when it is used again (it was not: wrong bias corrections from then on)."""
    from gapartnet_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(s, generator=g).to(cuda) for s in [(64, 27, 16), (300,), (16,), (48, 48)]]
    a = [torch.nn.Parameter(t.clone()) for t in base]
    b = [torch.nn.Parameter(t.clone()) for t in base]
    opt_a = FusedAdam(a, lr=1e-2)
    opt_b = torch.optim.Adam(b, lr=1e-2, foreach=False, fused=False)
    hold = []
    for step in range(9):
        for i, (p, q) in enumerate(zip(a, b)):
            if step in (3, 6, 7) and i in (1, 2):
                p.grad = q.grad = None
                continue
            grad = torch.randn(p.shape, generator=g).to(cuda)
            if step % 2:
                hold.append(torch.empty(p.shape, device=cuda))  # (the allocator hands out other blocks: gradient addresses move)
            p.grad, q.grad = grad.clone(), grad.clone()
        opt_a.step()
        opt_b.step()
    opt_a.state_dict()
    for p, q in zip(a, b):
        assert float(opt_a.state[p]["step"]) == float(opt_b.state[q]["step"])
        assert torch.allclose(p, q, rtol=1e-6, atol=2e-7), float((p - q).abs().max())


def _bn_launches(lib):
    import ctypes
    launches, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    assert lib.gpn_prof_get(6, ctypes.byref(launches), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) == 0  # GPN_K_BN
    return int(launches.value)


@pytest.mark.gpu
@pytest.mark.parametrize("without_stem", [False, True])
def test_inference_pass_applies_batchnorm_in_the_conv_epilogue(cuda, without_stem):
    """(round 6) eval mode without a backward pass to follow (GPN_NET_INFERENCE: the executor called with gradients disabled):
    every BatchNorm behind a conv is applied by that conv's launch - the arithmetic of the stand-alone eval pass per element, so
    the output is BIT-equal to the eval pass with gradients enabled (which keeps its BatchNorm launches for the backward pass),
    and the pass has no BatchNorm launch left but a stem norm that follows no conv.  Single and paired passes, masked-tile,
    masked tap-split and k = 1 kernels (the shortcut convs of the decoder blocks)."""
    from gapartnet_amd import _C
    from gapartnet_amd.network import net_exec
    lib = _C.lib()
    net, idx, feats, spconv = _unet_case(cuda, without_stem)
    twin = copy.deepcopy(net)
    with torch.no_grad():
        for m in list(net.modules()) + list(twin.modules()):  # non-trivial running statistics and affine parameters
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0.0, 0.3)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.2)
        for p in twin.parameters():
            p.mul_(0.7)
    net.eval(), twin.eval()
    prev_tiles = lib.gpn_spconv_tiles_min_tiles(256)  # (the largest level of this case on the masked-tile kernel)
    try:
        def run(no_grad):
            lib.gpn_prof_reset(); lib.gpn_prof_enable(1)
            ctx = torch.no_grad() if no_grad else torch.enable_grad()
            with ctx:
                y = net(spconv.SparseConvTensor(feats.clone(), idx, [64, 64, 64], 3)).features
                single_bn = _bn_launches(lib)
                lib.gpn_prof_reset()
                # (paired passes exist for networks without a stem conv - the proposal networks; the backbone's 6-channel stem runs
                # as a module in front of the program)
                pair = net_exec.run_pair(net, twin, spconv.SparseConvTensor(feats.clone(), idx, [64, 64, 64], 3)) if without_stem else None
            torch.cuda.synchronize()
            pair_bn = _bn_launches(lib) if pair is not None else single_bn
            lib.gpn_prof_enable(0)
            if pair is None:
                return y.detach(), y.detach(), y.detach(), single_bn, pair_bn
            return y.detach(), pair[0].features.detach(), pair[1].features.detach(), single_bn, pair_bn

        y0, a0, b0, n0, m0 = run(False)
        y1, a1, b1, n1, m1 = run(True)
    finally:
        lib.gpn_spconv_tiles_min_tiles(prev_tiles)
        lib.gpn_prof_enable(0)
    n_bn = sum(isinstance(m, torch.nn.BatchNorm1d) for m in net.modules())
    assert n0 == n_bn and m0 == n_bn, (n0, m0, n_bn)                 # eval with a backward to come: a launch per BatchNorm (pair: per pair)
    # inference: only the BatchNorm that follows no conv of the program is left - the stem norm (of a network without a stem conv, or
    # behind the 6-channel stem conv that runs as a module in front of the program)
    assert n1 == 1 and m1 == 1, (n1, m1)
    assert torch.equal(y0, y1) and torch.equal(a0, a1) and torch.equal(b0, b1)
    assert torch.equal(y0, a0), "the pair's first network is the single pass's"


def test_gated_adam_counts_skips_per_table(cuda):
    """(round 6) gated tensors that started at different times - training_schedule [5, 10]: ScoreNet's from epoch 5,
    NPCS-Net's from epoch 10 - are two device tables; a step the gate suppresses must count ONCE for each, and only against
    tensors that had started (one shared counter counted it twice and charged it to both).  Against torch.optim.Adam with
    ``grad = None`` on the suppressed steps (what the reference does on a batch without proposals)."""
    from gapartnet_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(7)
    base = [torch.randn(s, generator=g).to(cuda) for s in [(32, 27, 16), (200,), (16, 16), (48,)]]
    a = [torch.nn.Parameter(t.clone()) for t in base]  # a[0] ungated (backbone), a[1] early gated, a[2], a[3] late gated
    b = [torch.nn.Parameter(t.clone()) for t in base]
    opt_a = FusedAdam(a, lr=1e-2)
    opt_b = torch.optim.Adam(b, lr=1e-2, foreach=False, fused=False)
    gate = torch.ones((2,), dtype=torch.int64, device=cuda)
    opt_a.set_gate(a[1:], lambda: (gate, 1))
    closed_steps = (3, 4, 8)
    for step in range(11):
        started = [True, True, step >= 2, step >= 2]
        gate[1] = 0 if step in closed_steps else 1
        for i, (p, q) in enumerate(zip(a, b)):
            if not started[i]:
                p.grad = q.grad = None
                continue
            grad = torch.randn(p.shape, generator=g).to(cuda)
            # the device-counted step hands the gated tensors a (zero) gradient although nothing ran; the reference leaves None
            p.grad = grad.clone() if (i == 0 or step not in closed_steps) else torch.zeros_like(p)
            q.grad = grad.clone() if (i == 0 or step not in closed_steps) else None
        opt_a.step()
        opt_b.step()
        if step == 6:
            opt_a.state_dict()  # (a checkpoint in the middle: the counters are folded into the host counts and start again)
    opt_a.state_dict()
    for i, (p, q) in enumerate(zip(a, b)):
        assert float(opt_a.state[p]["step"]) == float(opt_b.state[q]["step"]), (i, float(opt_a.state[p]["step"]), float(opt_b.state[q]["step"]))
        assert torch.allclose(p, q, rtol=1e-6, atol=2e-7), (i, float((p - q).abs().max()))
        assert torch.allclose(opt_a.state[p]["exp_avg"], opt_b.state[q]["exp_avg"], rtol=1e-6, atol=1e-7), i


def test_grouped_weight_gradient_contractions_are_bit_equal(cuda):
    """gpn_net_wgrad_group: up to 4 consecutive same-shape layers of a level (and the two networks of a paired pass) share a
    weight-gradient contraction launch on the executor's second stream.  Same per-layer arithmetic for every setting: all
    parameter gradients of a U-Net backward - single and paired passes - are bit-equal at 1, 2 and 4 layers per launch."""
    from gapartnet_amd import _C
    from gapartnet_amd.network import net_exec
    L = _C.lib()
    prev = L.gpn_net_wgrad_group(-1)
    assert prev == 4, "grouping of 4 is the default (csrc/net.hip)"
    try:
        results = {}
        for layers in (1, 2, 4):
            L.gpn_net_wgrad_group(layers)
            assert L.gpn_net_wgrad_group(-1) == layers
            net, idx, feats, spconv = _unet_case(cuda, True)
            twin = copy.deepcopy(net)
            with torch.no_grad():
                for p in twin.parameters():
                    p.mul_(0.5)
            x = feats.clone().requires_grad_(True)
            out = net(spconv.SparseConvTensor(x, idx, [64, 64, 64], 3)).features
            out.square().sum().backward()
            single = [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]
            net.zero_grad(set_to_none=True)
            x2 = feats.clone().requires_grad_(True)
            pair = net_exec.run_pair(net, twin, spconv.SparseConvTensor(x2, idx, [64, 64, 64], 3))
            assert pair is not None
            (pair[0].features.square().sum() + pair[1].features.sum()).backward()
            paired = [p.grad.clone() for p in list(net.parameters()) + list(twin.parameters())] + [x2.grad.clone()]
            torch.cuda.synchronize()
            results[layers] = (single, paired)
        for layers in (2, 4):
            for a, b in zip(results[1][0] + results[1][1], results[layers][0] + results[layers][1]):
                assert torch.equal(a, b), f"grouping {layers}: gradients differ from one layer per launch"
    finally:
        L.gpn_net_wgrad_group(prev)


def test_step_switches_keep_the_step(cuda):
    """the two step-level switches of network/model.py that keep a second formulation reachable on GPU tensors - the proposal
    stage as torch ops (use_fused_proposals = False) and ScoreNet / NPCS-Net one after the other (pair_proposal_unets = False) -
    give the same training step as the default (fused stage, paired passes): loss and every gradient bit-equal."""
    batch = [pc.to(cuda) for pc in make_batch(2, 5000, seed0=321)]
    base = make_model((0, 0), channels=[16, 32, 48]).to(cuda)
    jitter = (torch.tensor([0.3, 0.6, 0.1], device=cuda), torch.tensor([0.5, 0.2, 0.9], device=cuda))
    runs = {}
    for fused, pair in ((True, True), (True, False), (False, True)):
        model = copy.deepcopy(base)
        model.revoxelize_jitter = jitter
        model.use_fused_proposals, model.pair_proposal_unets = fused, pair
        loss = model.training_step(batch, 0)
        loss.backward()
        runs[(fused, pair)] = (loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    ref_loss, ref_grads = runs[(True, True)]
    for key in ((True, False), (False, True)):
        loss, grads = runs[key]
        assert torch.equal(loss, ref_loss), (key, float(loss), float(ref_loss))
        for k in ref_grads:
            assert torch.equal(grads[k], ref_grads[k]), (key, k)
