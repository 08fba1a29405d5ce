"""BASELINE.json's configurations at their FULL sizes on the GPU (configs 2, 4 and 5 had no `-m gpu` test of
their own).  At these sizes a full comparison of integer outputs against the CPU oracle is not robust (80k-640k points:
an fp32 rounding difference of 1e-7 in a distance or a logit flips a handful of radius / arg-max decisions), so each test
splits the pipeline where the arithmetic changes kind:
  * the floating-point stage (backbone U-Net -> point heads) is compared with the oracle at north_star's 1e-4;
  * the integer stage (clustering -> proposal CSR -> re-voxelisation) is fed IDENTICAL inputs on both sides - the GPU's
    own predictions - and must then be bit-exact;
  * whole steps are checked through size-independent properties: finite, bitwise reproducible, proposal CSR consistent.
"""
import copy

import numpy as np
import pytest
import torch

from gapartnet_amd import backend
from gapartnet_amd.dataset import synthetic
from gapartnet_amd.dataset.gapartnet import compact_instance_labels, generate_inst_info
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.structure.point_cloud import PointCloud
from tests.golden import recipe

pytestmark = pytest.mark.gpu
VOXEL = (0.01, 0.01, 0.01)
JITTER = (torch.tensor([0.3, 0.6, 0.1]), torch.tensor([0.5, 0.2, 0.9]))


def _fp_stage(model, batch):
    pc_feature = model.forward_backbone(pc_batch=batch)
    return pc_feature, model.forward_sem_seg(pc_feature), model.forward_offset(pc_feature)


def _integer_stage(model, batch, pc_feature, sem_preds, offsets):
    vt, pid, props = model.proposal_clustering_and_revoxelize(
        pt_xyz=batch.points[:, :3], batch_indices=batch.batch_indices, pt_features=pc_feature, sem_preds=sem_preds,
        offset_preds=offsets, instance_labels=batch.instance_labels, batch_size=batch.batch_size)
    return vt, pid, props


def _check_csr(props, n_points):
    off = props.proposal_offsets.long()
    sizes = off[1:] - off[:-1]
    assert int(off[0]) == 0 and int(off[-1]) == props.sorted_indices.shape[0]
    assert bool((sizes >= 5).all()), "min_num_points_per_proposal"
    assert torch.equal(sizes, props.num_points_per_proposal.long())
    assert torch.equal(props.proposal_indices.long(), torch.repeat_interleave(torch.arange(sizes.shape[0], device=off.device), sizes))
    # a proposal lies in one scene and carries one predicted class
    first = off[:-1]
    assert torch.equal(props.batch_indices, props.batch_indices[first][props.proposal_indices.long()])
    assert torch.equal(props.sem_preds, props.sem_preds[first][props.proposal_indices.long()])
    assert int(props.sorted_indices.max()) < int(props.valid_mask.sum()) <= n_points


def test_config2_eval_batch4_full_pipeline(cuda):
    """config 2: full pipeline, 4 x 20k-point scenes, eval mode (release.ckpt is absent: seeded weights)."""
    from oracle import torch_ops
    model = make_model((0, 0)).eval()
    # seeded weights with non-trivial BatchNorm statistics (torch's default init predicts one class almost everywhere in
    # eval mode, and then there is nothing to cluster)
    model.load_state_dict(recipe.name_keyed_state(model))
    scenes = make_batch(4, 20000, seed0=2200)
    gpu_model = copy.deepcopy(model).to(cuda)
    gpu_model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
    with torch.no_grad():
        gbatch = PointCloud.collate([pc.to(cuda) for pc in scenes], voxel_size=VOXEL)
        g_feat, g_sem, g_off = _fp_stage(gpu_model, gbatch)
        with backend.using(torch_ops):
            cbatch = PointCloud.collate(scenes, voxel_size=VOXEL)
            c_feat, c_sem, c_off = _fp_stage(model, cbatch)
    assert torch.equal(gbatch.voxel_tensor.indices.cpu(), cbatch.voxel_tensor.indices), "80k points: voxel set bit-exact"
    assert torch.equal(gbatch.pc_voxel_id.cpu(), cbatch.pc_voxel_id)
    for name, g, c in (("pc_feature", g_feat, c_feat), ("sem_logits", g_sem, c_sem), ("offsets", g_off, c_off)):
        err = float((g.cpu() - c).abs().max())
        assert err <= 1e-4 * max(1.0, float(c.abs().max())), (name, err)
    # integer stage on identical inputs (the GPU's predictions)
    sem_preds = torch.argmax(g_sem, dim=-1)
    with torch.no_grad():
        g_vt, g_pid, g_props = _integer_stage(gpu_model, gbatch, g_feat, sem_preds, g_off)
        model.revoxelize_jitter = JITTER
        with backend.using(torch_ops):
            c_vt, c_pid, c_props = _integer_stage(model, cbatch, g_feat.cpu(), sem_preds.cpu(), g_off.cpu())
    assert g_props is not None and g_props.proposal_offsets.shape[0] > 10
    for f in ("sorted_indices", "proposal_offsets", "proposal_indices", "batch_indices", "sem_preds", "instance_labels"):
        assert torch.equal(getattr(g_props, f).cpu(), getattr(c_props, f)), f"proposals.{f}: bit-exact on identical inputs"
    assert torch.equal(g_vt.indices.cpu(), c_vt.indices) and torch.equal(g_pid.cpu(), c_pid)
    assert torch.allclose(g_vt.features.cpu(), c_vt.features, rtol=0, atol=1e-5)
    _check_csr(g_props, 80000)
    # the whole validation step + epoch end: finite metrics for every logged key, reproducible
    logs = []
    for _ in range(2):
        m = copy.deepcopy(gpu_model)
        m.revoxelize_jitter = gpu_model.revoxelize_jitter
        rec = {}
        m._log_sink = lambda name, value, bs, sync, rec=rec: rec.__setitem__(name, float(value))
        with torch.no_grad():
            for loader_idx in range(3):
                m.validation_step([pc.to(cuda) for pc in scenes], 0, loader_idx)
            m.on_validation_epoch_end()
        logs.append(rec)
    assert logs[0] == logs[1] or all(np.isnan(logs[0][k]) == np.isnan(logs[1][k]) and
                                     (np.isnan(logs[0][k]) or logs[0][k] == logs[1][k]) for k in logs[0])
    assert "monitor_metrics/mean_mAP" in logs[0] and np.isfinite(logs[0]["val_loss/total_loss"])


def test_config4_dense_50k_train_step(cuda):
    """config 4: one 50k-point scene at voxel 0.01 (rulebook / LDS stress), full train step."""
    from oracle import torch_ops
    scenes = make_batch(1, 50000, seed0=4400)
    model = make_model((0, 0))
    model.load_state_dict(recipe.name_keyed_state(model))
    # fp stage vs the oracle in eval mode (fixed BatchNorm statistics)
    with torch.no_grad():
        gbatch = PointCloud.collate([pc.to(cuda) for pc in scenes], voxel_size=VOXEL)
        g_feat, g_sem, g_off = _fp_stage(copy.deepcopy(model).eval().to(cuda), gbatch)
        with backend.using(torch_ops):
            cbatch = PointCloud.collate(scenes, voxel_size=VOXEL)
            c_feat, c_sem, c_off = _fp_stage(copy.deepcopy(model).eval(), cbatch)
    assert torch.equal(gbatch.voxel_tensor.indices.cpu(), cbatch.voxel_tensor.indices)
    assert gbatch.voxel_tensor.indices.shape[0] > 25000, "dense scene: ~29k active voxels expected"
    for name, g, c in (("pc_feature", g_feat, c_feat), ("sem_logits", g_sem, c_sem), ("offsets", g_off, c_off)):
        err = float((g.cpu() - c).abs().max())
        assert err <= 1e-4 * max(1.0, float(c.abs().max())), (name, err)
    # train step: finite, every sub-network gets gradients, bitwise reproducible
    losses, grads = [], []
    for _ in range(2):
        m = copy.deepcopy(model).to(cuda).train()
        m.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
        loss = m.training_step([pc.to(cuda) for pc in scenes], 0)
        loss.backward()
        losses.append(float(loss))
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert np.isfinite(losses[0]) and losses[0] == losses[1]
    assert set(grads[0]) == set(n for n, _ in model.named_parameters()), "all three U-Nets and all heads trained"
    for n in grads[0]:
        assert torch.isfinite(grads[0][n]).all() and torch.equal(grads[0][n], grads[1][n]), n


def _mixed_category_scenes(n_scenes, n_points=20000):
    """config 5: scenes from four generator 'categories' (object extents and part counts differ)"""
    cats = [dict(extent_range=(0.4, 1.0), parts_range=(3, 12)), dict(extent_range=(0.2, 0.5), parts_range=(8, 12)),
            dict(extent_range=(0.7, 1.0), parts_range=(1, 3)), dict(extent_range=(0.3, 0.9), parts_range=(5, 7))]
    out = []
    for i in range(n_scenes):
        xyz, rgb, sem, ins, npcs, _ = synthetic.make_scene_arrays(5500 + i, n_points, **cats[i % 4])
        pc = PointCloud(pc_id=f"Cat{i % 4}_{i}_0_0", obj_cat=i % 4, points=np.concatenate([xyz, rgb], 1).astype(np.float32),
                        sem_labels=sem.astype(np.int64), instance_labels=ins.astype(np.int32), gt_npcs=npcs.astype(np.float32))
        out.append(generate_inst_info(compact_instance_labels(pc)).to_tensor())
    return out


def test_config5_mixed_batch32_npcs_and_pose_heads(cuda):
    """config 5 (per-rank view of the 4-GPU job: the global batch of 32 on ONE device): NPCS head on, train step, then
    the test path with batched pose fitting on the predicted NPCS of every kept proposal."""
    from gapartnet_amd.misc.pose_fitting_batched import estimate_pose_from_npcs_batched
    scenes = [pc.to(cuda) for pc in _mixed_category_scenes(32)]
    model = make_model((0, 0))
    model.load_state_dict(recipe.name_keyed_state(model))
    model = model.to(cuda)
    model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
    rec = {}
    model._log_sink = lambda name, value, bs, sync: rec.__setitem__(name, float(value))
    loss = model.training_step(scenes, 0)
    loss.backward()
    assert np.isfinite(float(loss)) and rec["train_loss/loss_prop_npcs"] > 0 and rec["train_loss/loss_prop_score"] > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.npcs_unet.parameters())
    assert all(p.grad is not None for p in model.npcs_head.parameters())
    # test path: proposals after score filtering + NMS carry per-point NPCS predictions; fit a pose per proposal
    model.eval()
    with torch.no_grad():
        pc_ids, sem_seg, kept = model.test_step(scenes, 0, 0)
    assert kept is not None and len(pc_ids) == 32 and sem_seg.sem_preds.shape[0] == 32 * 20000
    P = kept.proposal_offsets.shape[0] - 1
    assert P > 0 and kept.npcs_preds is not None
    # NPCS predictions exist for the points that pass the NPCS validity mask; pose fitting runs on those, per proposal
    mask = kept.npcs_valid_mask
    pid = kept.proposal_indices[mask].long()
    assert kept.npcs_preds.shape[0] == int(mask.sum()) == pid.shape[0]
    counts = torch.bincount(pid, minlength=P)
    offsets = torch.zeros(P + 1, dtype=torch.int64, device=cuda)
    offsets[1:] = counts.cumsum(0)
    keep = counts > 0
    np.random.seed(0)
    if int(keep.sum()) > 0:
        nz_off = torch.zeros(int(keep.sum()) + 1, dtype=torch.int64, device=cuda)
        nz_off[1:] = counts[keep].cumsum(0)
        fit = estimate_pose_from_npcs_batched(kept.pt_xyz[mask], kept.npcs_preds, nz_off, max_iters=20)
        assert fit["bbox"].shape == (int(keep.sum()), 8, 3) and fit["valid"].dtype == torch.bool
        ok = fit["valid"]
        assert torch.isfinite(fit["bbox"][ok]).all() and torch.isfinite(fit["scale"][ok]).all()
        if int(ok.sum()):
            r = fit["rotation"][ok]
            eye = torch.eye(3, dtype=r.dtype, device=cuda)
            assert torch.allclose(r @ r.transpose(1, 2), eye.expand_as(r), atol=1e-8), "rotations are orthonormal"


def test_training_mode_backbone_levels_at_full_size_match_the_oracle(cuda):
    """TRAINING-mode floating point at a full-size level (the full-size configs compared the fp stage in eval mode
    only; training-mode BatchNorm at 144k rows was covered by run-to-run determinism alone, which a wrong-but-deterministic
    sum epilogue would pass).  Stem + a two-level residual U-Net (16 / 32 channels: the masked-tile kernel with tile order,
    BatchNorm sums in the conv / dgrad epilogues, stride-2 / inverse convs, weight-gradient contractions) on 8 x 20k-point
    scenes, forward AND every parameter gradient against the CPU oracle running the same modules: features at north_star's
    1e-4, BatchNorm running statistics at 1e-4, gradients by relative L2 error (see below)."""
    import functools
    import torch.nn as nn
    from gapartnet_amd.network.backbone import SparseUNet
    from oracle import torch_ops
    scenes = make_batch(8, 20000, seed0=9100)
    torch.manual_seed(11)
    net = SparseUNet.build(6, [16, 32], 2, functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)).train()
    with torch.no_grad():  # non-trivial affine parameters and running statistics
        for m in net.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.3, 0.3)
    gnet = copy.deepcopy(net).to(cuda)
    gbatch = PointCloud.collate([pc.to(cuda) for pc in scenes], voxel_size=VOXEL)
    assert gbatch.voxel_tensor.features.shape[0] > 130000
    g_out = gnet(gbatch.voxel_tensor).features
    weight = torch.linspace(-1.0, 1.0, g_out.shape[1], device=cuda)
    (g_out * weight).sum().backward()
    with backend.using(torch_ops):
        cbatch = PointCloud.collate(scenes, voxel_size=VOXEL)
        c_out = net(cbatch.voxel_tensor).features
        (c_out * weight.cpu()).sum().backward()
    assert torch.equal(gbatch.voxel_tensor.indices.cpu(), cbatch.voxel_tensor.indices)
    err = float((g_out.detach().cpu() - c_out.detach()).abs().max())
    assert err <= 1e-4 * max(1.0, float(c_out.abs().max())), f"training-mode features: {err:.3e}"
    # Two fp32 evaluations of a ReLU network disagree on the SIGN of the few pre-activations that are zero to rounding (about
    # one in 10^6 of the 7 M here): such a flip moves single gradient entries by O(1) - measured: one entry of one weight
    # off by 3e-3 x max|g| in all three GPU paths alike (tools/probes/train_mode_probe.py), everything else <= 4e-4.  So per
    # tensor: relative L2 error (a wrong sum moves EVERY entry and shows here) and the share of entries off by more than
    # 1e-3 x max|g| (a handful of flips does not).
    worst_l2, worst_share, worst_max = ("", 0.0), ("", 0.0), ("", 0.0)
    for (name, p), (_, q) in zip(gnet.named_parameters(), net.named_parameters()):
        assert (p.grad is None) == (q.grad is None), name
        if q.grad is None:
            continue
        scale = float(q.grad.abs().max())
        d = (p.grad.cpu() - q.grad).abs()
        if scale <= 1e-9:
            assert float(d.max()) <= 1e-6, name
            continue
        l2 = float(d.double().norm() / q.grad.double().norm())
        share = float((d > 1e-3 * scale).double().mean())
        worst_l2 = max(worst_l2, (name, l2), key=lambda t: t[1])
        worst_share = max(worst_share, (name, share), key=lambda t: t[1])
        worst_max = max(worst_max, (name, float(d.max()) / scale), key=lambda t: t[1])
    print("training-mode full-size gradients: worst relative L2 error %.2e at %s; worst share of entries off by > 1e-3 max|g| "
          "%.2e at %s; worst single entry %.2e x max|g| at %s" % (worst_l2[1], worst_l2[0], worst_share[1], worst_share[0],
                                                                  worst_max[1], worst_max[0]))
    assert worst_l2[1] <= 1e-3, worst_l2  # (measured 4.3e-4: sums of ~10^5 cancelling terms behind training-mode BatchNorms)
    assert worst_share[1] <= 0.01, worst_share
    assert worst_max[1] <= 2e-2, worst_max
    for (name, b), (_, c) in zip(gnet.named_buffers(), net.named_buffers()):
        if b.dtype.is_floating_point:
            assert torch.allclose(b.cpu(), c, rtol=1e-4, atol=1e-6), f"running statistic {name}"


def test_full_model_training_step_at_full_size_matches_the_oracle(cuda):
    """BASELINE config 3's per-GPU shape with nothing reduced: the 7-level model, 8 x 20k-point scenes, every head on, training
    mode, forward + backward - against the same step through the CPU oracle (~15 s).  The loss agrees to 1e-6.  Gradients: the
    score branch max-pools ~7800 (proposal, channel) entries over the proposals' voxels, and two voxels whose features agree to
    the last bit but one are a TIE the two fp32 evaluations may break differently - one such entry moves every gradient behind it
    by sqrt(2 / 7800) = 1.6 % in relative L2 (measured, round 5: exactly one flipped entry, gap 1.2e-7, ScoreNet gradients 0.7 -
    1.3 % off, the backbone's 0.25 %).  So the check has two parts: the ties are counted and must be ties (<= 1e-6), and the step
    is run with the max-pool's routing taken from the oracle, after which ScoreNet and the heads agree to 1e-5, NPCS-Net to 2e-3
    and the backbone to 3e-3 in relative L2 per tensor (achieved 1.9e-6 / 4.3e-4 / 1.1e-3: the backbone's ~7 M pre-activations
    that are zero to rounding flip single ReLU masks, as in the two-level test above)."""
    from collections import defaultdict
    from gapartnet_amd import hip_ops as H
    from oracle import torch_ops as oracle_ops
    model = make_model((0, 0))
    jitter = (torch.tensor([0.3, 0.6, 0.1]), torch.tensor([0.5, 0.2, 0.9]))
    batch = make_batch(8, 20000)
    ref_model = copy.deepcopy(model)
    ref_model.revoxelize_jitter = jitter
    seen = {}
    o_fwd, h_fwd = oracle_ops.segmented_maxpool_fwd, H.segmented_maxpool_fwd

    def spy_oracle(values, begin, end, *a, **k):
        out = o_fwd(values, begin, end, *a, **k)
        seen["oracle"] = (out[1].clone(), values.detach().clone())
        return out

    def spy_hip(values, begin, end, *a, **k):
        out = h_fwd(values, begin, end, *a, **k)
        seen["hip"] = out[1].cpu()
        return out[0], seen["oracle"][0].to(out[1].device, out[1].dtype)  # (ties broken as the oracle broke them)

    oracle_ops.segmented_maxpool_fwd, H.segmented_maxpool_fwd = spy_oracle, spy_hip
    try:
        with backend.using(oracle_ops):
            ref_loss = ref_model.training_step(batch, 0)
            ref_loss.backward()
        model = model.to(cuda)
        model.revoxelize_jitter = tuple(j.to(cuda) for j in jitter)
        loss = model.training_step([pc.to(cuda) for pc in batch], 0)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        oracle_ops.segmented_maxpool_fwd, H.segmented_maxpool_fwd = o_fwd, h_fwd
    assert float(loss.detach()) == pytest.approx(float(ref_loss.detach()), rel=1e-6)
    arg_o, values = seen["oracle"]
    arg_h = seen["hip"]
    assert arg_o.shape == arg_h.shape and arg_o.numel() > 5000
    flipped = (arg_o.long() != arg_h.long()).nonzero()
    assert flipped.shape[0] <= 8, f"{flipped.shape[0]} max-pool entries routed differently"
    if flipped.shape[0]:
        r, c = flipped[:, 0], flipped[:, 1]
        gap = (values[arg_o.long()[r, c], c] - values[arg_h.long()[r, c], c]).abs()
        assert float(gap.max()) <= 1e-6, "a max-pool entry routed differently that is not a tie"
    ref = {n: q.grad for n, q in ref_model.named_parameters()}
    top = max(float(g.abs().max()) for g in ref.values() if g is not None)
    worst = defaultdict(float)
    for name, p in model.named_parameters():
        g_ref = ref[name]
        assert (g_ref is None) == (p.grad is None), name
        if g_ref is None or float(g_ref.abs().max()) <= 1e-6 * top:
            continue
        g = p.grad.detach().cpu()
        group = name.split(".")[0]
        worst[group] = max(worst[group], float((g - g_ref).norm() / g_ref.norm()))
    bounds = dict(backbone=3e-3, npcs_unet=2e-3, score_unet=1e-5, score_head=1e-5, npcs_head=1e-5, sem_seg_head=1e-5, offset_head=1e-5)
    assert set(worst) == set(bounds), sorted(worst)
    for group, bound in bounds.items():
        assert worst[group] <= bound, (group, worst[group], dict(worst))
