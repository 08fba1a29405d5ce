"""Host-side logic on CPU: the spconv / epic_ops mirrors, autograd wrappers, collate, model and trainer run over the
CPU oracle (``oracle.torch_ops`` installed as the raw-op backend for the duration of a test — the product default is
the HIP library and is restored afterwards)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gapartnet_amd import backend
from gapartnet_amd import functional as GF
from gapartnet_amd.dataset.gapartnet import (SyntheticGAPartNetDataset, apply_voxelization, compact_instance_labels,
                                             generate_inst_info)
from gapartnet_amd.dataset.synthetic import make_scene, make_scene_arrays
from gapartnet_amd.network import grouping_utils as G
from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.spconv import pytorch as spconv
from gapartnet_amd.structure.point_cloud import PointCloud
from tests import synth


@pytest.fixture(autouse=True)
def oracle_backend():
    from oracle import torch_ops
    with backend.using(torch_ops):
        yield
    assert backend.raw().name == "hip"


def _sparse(rng, batch, shape, n, c):
    idx = synth.random_sparse_indices(rng, batch, shape, n)
    return spconv.SparseConvTensor(torch.from_numpy(rng.normal(size=(idx.shape[0], c)).astype(np.float32)),
                                   torch.from_numpy(idx), shape, batch)


def _dense_of(x):
    return x.dense()


def test_spconv_modules_match_dense_convs_and_autograd():
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    x = _sparse(rng, 2, [10, 12, 8], 400, 6)
    x._features.requires_grad_(True)
    subm = spconv.SubMConv3d(6, 16, 3, padding=1, bias=False, indice_key="subm1")
    down = spconv.SparseConv3d(16, 32, 2, stride=2, bias=False, indice_key="spconv1")
    up = spconv.SparseInverseConv3d(32, 16, 2, bias=False, indice_key="spconv1")
    y1 = subm(x)
    y2 = down(y1)
    y3 = up(y2)
    assert "subm1" in x.indice_dict and "spconv1" in x.indice_dict and y3.indices is x.indices
    assert y2.spatial_shape == [5, 6, 4]
    # dense reference with the same weights
    dx = x.dense().detach().requires_grad_(True)
    w1 = subm.weight.detach().permute(0, 4, 1, 2, 3)
    mask = (x.dense().abs().sum(1, keepdim=True) > 0).float()
    r1 = F.conv3d(dx, w1, padding=1) * mask
    w2 = down.weight.detach().permute(0, 4, 1, 2, 3)
    r2 = F.conv3d(r1, w2, stride=2)
    w3 = up.weight.detach().permute(4, 0, 1, 2, 3)
    r3 = F.conv_transpose3d(r2, w3, stride=2) * mask
    i = x.indices.long()
    assert torch.allclose(y3.features, r3[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]], atol=1e-4)
    g = torch.randn_like(y3.features)
    y3.features.backward(g)
    gd = torch.zeros_like(r3)
    gd[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = g
    r3.backward(gd)
    assert torch.allclose(x.features.grad, dx.grad[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]], atol=1e-4)
    assert subm.weight.grad.shape == subm.weight.shape and float(subm.weight.grad.abs().sum()) > 0


def test_spconv_v1_checkpoint_layout_is_accepted():
    conv = spconv.SubMConv3d(6, 16, 3, padding=1, bias=False)
    v1 = torch.randn(3, 3, 3, 6, 16)
    conv.load_state_dict({"weight": v1})
    assert torch.equal(conv.weight.data, v1.permute(4, 0, 1, 2, 3))
    assert torch.equal(conv.canonical_weight().data, v1.reshape(27, 6, 16))


def test_sparse_sequential_names_children_by_index():
    seq = spconv.SparseSequential(spconv.SubMConv3d(6, 16, 3, padding=1, bias=False), torch.nn.BatchNorm1d(16), torch.nn.ReLU())
    assert list(seq.state_dict().keys())[:2] == ["0.weight", "1.weight"]


def test_state_dict_keys_follow_the_reference_module_tree():
    keys = set(make_model((0, 0)).state_dict().keys())
    expected = [
        "backbone.stem.0.weight", "backbone.stem.1.running_mean", "backbone.ublock.encoder_blocks.0.conv1.0.weight",
        "backbone.ublock.encoder_blocks.1.conv2.1.bias", "backbone.ublock.downsample.0.weight",
        "backbone.ublock.downsample.1.num_batches_tracked", "backbone.ublock.ublock.ublock.encoder_blocks.0.conv1.0.weight",
        "backbone.ublock.upsample.0.weight", "backbone.ublock.decoder_blocks.0.shortcut.0.weight",
        "backbone.ublock.decoder_blocks.0.shortcut.1.weight", "backbone.ublock.decoder_blocks.1.conv2.0.weight",
        "sem_seg_head.weight", "offset_head.0.weight", "offset_head.1.running_var", "offset_head.3.bias",
        "score_unet.stem.0.weight", "score_unet.ublock.downsample.0.weight", "score_head.bias",
        "npcs_unet.ublock.upsample.0.weight", "npcs_head.weight"]
    missing = [k for k in expected if k not in keys]
    assert not missing, missing
    m = make_model((0, 0))
    assert m.backbone.stem[0].weight.shape == (16, 3, 3, 3, 6)
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 7_897_617, n_params  # 7,532,128 backbone + 2 x 182,176 + 1,137 heads
    assert sum(isinstance(mod, (spconv.SubMConv3d, spconv.SparseConv3d, spconv.SparseInverseConv3d)) for mod in m.modules()) == 101


def test_gather_rows_and_maxpool_gradients():
    rng = np.random.default_rng(1)
    table = torch.from_numpy(rng.normal(size=(50, 4)).astype(np.float32)).requires_grad_(True)
    idx = torch.from_numpy(rng.integers(-1, 50, 300).astype(np.int32))
    out = GF.gather_rows(table, idx)
    ref = torch.where((idx >= 0)[:, None], table[idx.clamp(min=0).long()], torch.zeros(()))
    assert torch.equal(out, ref)
    out.sum().backward()
    assert torch.allclose(table.grad[:, 0], torch.bincount(idx[idx >= 0].long(), minlength=50).float())
    vals = torch.from_numpy(rng.normal(size=(300, 4)).astype(np.float32)).requires_grad_(True)
    offs = torch.tensor([0, 100, 180, 300], dtype=torch.int32)
    pooled, arg = GF.segmented_maxpool(vals, offs[:-1], offs[1:])
    pooled.sum().backward()
    assert float(vals.grad.sum()) == 12.0 and torch.equal(vals.grad.nonzero()[:, 0].sort()[0], arg.reshape(-1).long().sort()[0])


def test_batched_collate_equals_per_scene_voxelisation():
    scenes = [generate_inst_info(compact_instance_labels(make_scene(1000 + i, 2500))).to_tensor() for i in range(3)]
    batched = PointCloud.collate(scenes, voxel_size=(0.01, 0.01, 0.01))
    per_scene = PointCloud.collate([apply_voxelization(copy.copy(s), voxel_size=(0.01, 0.01, 0.01)) for s in scenes])
    assert torch.equal(batched.voxel_tensor.indices, per_scene.voxel_tensor.indices)
    assert torch.equal(batched.voxel_tensor.features, per_scene.voxel_tensor.features)
    assert torch.equal(batched.pc_voxel_id, per_scene.pc_voxel_id.to(batched.pc_voxel_id.dtype))
    assert batched.voxel_tensor.spatial_shape == per_scene.voxel_tensor.spatial_shape
    assert batched.voxel_tensor.spatial_shape[0] >= 128 and int(batched.num_points_per_instance.sum()) > 0
    assert batched.batch_indices.dtype == torch.int32 and batched.points.shape == (7500, 6)


def test_dataset_contract():
    arrays = make_scene_arrays(1000, 2000)
    assert [a.shape for a in arrays] == [(2000, 3), (2000, 3), (2000,), (2000,), (2000, 3), (2000, 2)]
    assert abs(np.linalg.norm(arrays[0], axis=1).max() - 1.0) < 1e-5 and (arrays[3] >= 0).any()
    pc = SyntheticGAPartNetDataset(2, n_points=2000)[1]
    assert pc.instance_regions.shape == (2000, 9) and pc.num_instances == int(pc.instance_labels.max()) + 1
    lab = pc.instance_labels.numpy()
    for k in range(pc.num_instances):
        sel = lab == k
        assert np.allclose(pc.instance_regions[sel, 0:3].numpy(), pc.points[sel, :3].mean(0).numpy(), atol=1e-5)
        assert np.allclose(pc.instance_regions[sel, 3:6].numpy(), pc.points[sel, :3].min(0)[0].numpy())
        assert int(pc.num_points_per_instance[k]) == int(sel.sum())


def test_cluster_and_nms_glue():
    rng = np.random.default_rng(2)
    pts, batch = synth.clustered_points(rng, 2, 600, n_clusters=4)
    offs = torch.tensor([0, 600, 1200], dtype=torch.int32)
    sem = torch.ones(1200, dtype=torch.int32)
    labels, order = G.cluster_proposals(torch.from_numpy(pts), torch.from_numpy(batch), offs, sem, 0.06, 50)
    assert torch.all(labels[1:] >= labels[:-1]) and sorted(order.tolist()) == list(range(1200))
    uniq = torch.unique(labels)
    assert 2 <= uniq.numel() <= 60
    # intersections from the incidence list == dense one-hot product
    sorted_indices = torch.cat([torch.arange(40), torch.arange(20, 70), torch.arange(60, 75)])
    prop = torch.cat([torch.zeros(40), torch.ones(50), torch.full((15,), 2)]).long()
    inter = G.proposal_intersections(sorted_indices, prop, 3)
    onehot = torch.zeros(3, 75)
    onehot[prop, sorted_indices] = 1
    assert torch.equal(inter, onehot @ onehot.T)


def test_full_train_step_on_cpu_is_finite_and_deterministic():
    batch = make_batch(2, 2500)
    losses = []
    for _ in range(2):
        model = make_model((0, 0), channels=[16, 32, 48])
        model.revoxelize_jitter = (torch.tensor([0.3, 0.6, 0.1]), torch.tensor([0.5, 0.2, 0.9]))
        logged = {}
        model._log_sink = lambda name, value, bs, sync: logged.__setitem__(name, float(value))
        loss = model.training_step(batch, 0)
        loss.backward()
        losses.append(float(loss))
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        assert {"train_loss/total_loss", "train_loss/loss_sem_seg", "train_loss/loss_offset_dist",
                "train_loss/loss_offset_dir", "train_loss/loss_prop_score", "train_loss/loss_prop_npcs",
                "train/all_accu", "train/pixel_accu"} <= set(logged)
        assert logged["train_loss/loss_prop_score"] > 0 and logged["train_loss/loss_prop_npcs"] > 0
    assert losses[0] == losses[1]


def test_schedule_gates_the_proposal_heads():
    batch = make_batch(1, 2000)
    model = make_model((5, 10), channels=[16, 32])
    logged = {}
    model._log_sink = lambda name, value, bs, sync: logged.__setitem__(name, float(value))
    model.training_step(batch, 0).backward()
    assert logged["train_loss/loss_prop_score"] == 0.0 and logged["train_loss/loss_prop_npcs"] == 0.0
    assert all(p.grad is None for p in model.score_unet.parameters())


def test_validation_epoch_produces_ap_metrics():
    from gapartnet_amd.trainer import MetricLog
    model = make_model((0, 0), channels=[16, 32, 48])
    model.eval()
    log = MetricLog()
    model._log_sink = log
    with torch.no_grad():
        for loader_idx in range(3):
            model.validation_step(make_batch(2, 2500, seed0=2000 + 10 * loader_idx), 0, loader_idx)
    model.on_validation_epoch_end()
    metrics = log.reduce(torch.device("cpu"))
    for key in ("val/AP@50", "val/mAP", "val/miou", "test_intra/all_accu", "test_inter/pixel_accu",
                "monitor_metrics/mean_mAP", "monitor_metrics/mean_AP@50", "monitor_metrics/mean_imou"):
        assert key in metrics and np.isfinite(metrics[key]), key
    assert model.validation_step_outputs == []


def test_sync_free_loss_forms_equal_the_selecting_forms():
    """the masked (no boolean-mask selection) forms used in the training step == the reference's selecting forms"""
    from gapartnet_amd.network.losses import focal_loss
    g = torch.Generator().manual_seed(0)
    # NPCS loss: members selected vs masked
    n, P, m = 400, 9, 4
    pi = torch.sort(torch.randint(0, P, (n,), generator=g))[0]
    pred = torch.rand((n, 3), generator=g, requires_grad=True)
    gt = torch.rand((n, 3), generator=g) - 0.5
    sym = torch.linalg.qr(torch.randn((n, m, 3, 3), generator=g))[0]
    for member in (torch.rand((n,), generator=g) < 0.4, torch.zeros((n,), dtype=torch.bool), pi != 3):
        masked = G.compute_npcs_loss_masked(pred, gt, pi, sym, member, P)
        if member.any():
            want = G.compute_npcs_loss(pred[member], gt[member], pi[member], sym[member])
            assert torch.allclose(masked, want, rtol=1e-5, atol=1e-7)
            ga = torch.autograd.grad(masked, pred)[0]
            gb = torch.autograd.grad(want, pred)[0]
            assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-8)
        else:
            assert masked.item() == 0.0
    # all three symmetry groups at once (the form the training step uses) == the reference's per-group selecting loop
    from gapartnet_amd.misc.info import get_symmetry_matrix
    t1, t2, t3 = get_symmetry_matrix()
    tables = G.SymmetryTables((t1, t2, t3), torch.device("cpu"))
    for case in range(3):
        types = torch.randint(0, 5, (n,), generator=g) if case < 2 else torch.full((n,), 1)
        if case == 1:
            types[pi == 2] = 4
        got = G.compute_npcs_loss_grouped(pred, gt, pi, types, tables, P)
        want = 0
        for mask, table, base in ((types < 3, t1, 0), (types == 3, t2, 3), (types == 4, t3, 4)):
            if mask.any():
                want = want + G.compute_npcs_loss(pred[mask], gt[mask], pi[mask], table[types[mask] - base])
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-7), (case, got, want)
        ga, gb = torch.autograd.grad(got, pred)[0], torch.autograd.grad(want, pred)[0]
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-8)
    # focal loss with ignored rows
    logits = torch.randn((50, 6), generator=g, requires_grad=True)
    target = torch.randint(0, 6, (50,), generator=g)
    target[::7] = -100
    keep = target != -100
    for reduction in ("mean", "sum"):
        got = focal_loss(logits, target, gamma=2.0, reduction=reduction, ignore_index=-100)
        want = focal_loss(logits[keep], target[keep], gamma=2.0, reduction=reduction, ignore_index=None)
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert focal_loss(logits, torch.full((50,), -100), ignore_index=-100).item() == 0.0


def test_clustering_with_known_batch_size_equals_the_compacting_form():
    """proposal_clustering_and_revoxelize(batch_size=B) (CSR over all scenes, no host read) gives the same proposals as
    the reference's form (CSR over the scenes that still have points), also when a scene has no valid point"""
    from oracle import torch_ops
    model = make_model((0, 0), channels=[16, 32])
    model.revoxelize_jitter = (torch.full((3,), 0.25), torch.full((3,), 0.75))
    rng = np.random.default_rng(4)
    pts, batch = synth.clustered_points(rng, 3, 600, n_clusters=4)
    xyz = torch.from_numpy(pts)
    bidx = torch.from_numpy(batch)
    feats = torch.from_numpy(rng.normal(size=(xyz.shape[0], 16)).astype(np.float32))
    sem = torch.from_numpy(rng.integers(1, 3, xyz.shape[0]).astype(np.int64))
    sem[bidx == 1] = 0  # scene 1 contributes no valid point
    off = torch.zeros_like(xyz)
    with backend.using(torch_ops):
        a = model.proposal_clustering_and_revoxelize(xyz, bidx, feats, sem, off, None)
        b = model.proposal_clustering_and_revoxelize(xyz, bidx, feats, sem, off, None, batch_size=3)
    assert torch.equal(a[0].indices, b[0].indices) and torch.equal(a[0].features, b[0].features)
    assert torch.equal(a[1], b[1])
    for f in ("sorted_indices", "proposal_offsets", "proposal_indices", "batch_indices", "sem_preds"):
        assert torch.equal(getattr(a[2], f), getattr(b[2], f)), f


def test_voxel_mean_backpropagates_to_point_features():
    """d feats[i] = d voxel[pc_voxel_id[i]] / count: checked against a plain torch restatement (index_add mean)"""
    from oracle import torch_ops
    rng = np.random.default_rng(11)
    pts = torch.from_numpy(rng.uniform(0, 4, (300, 3)).astype(np.float32))
    pts[:5] = 9.0  # outside the grid: no voxel, no gradient
    feats = torch.from_numpy(rng.normal(size=(300, 5)).astype(np.float32)).requires_grad_(True)
    offs = torch.tensor([0, 120, 300], dtype=torch.int64)
    rmin, rmax = torch.zeros((1, 3)), torch.full((1, 3), 4.0)
    with backend.using(torch_ops):
        vf, vc, vseg, pid, order, starts, _ = GF.voxelize_mean(pts, feats, offs, rmin, rmax, [1.0, 1.0, 1.0], [5, 5, 5])
    w = torch.from_numpy(rng.normal(size=tuple(vf.shape)).astype(np.float32))
    (vf * w).sum().backward()
    ref_feats = feats.detach().clone().requires_grad_(True)
    keep = pid >= 0
    V = vf.shape[0]
    sums = torch.zeros((V, 5)).index_add(0, pid[keep].long(), ref_feats[keep])
    cnt = torch.zeros((V,)).index_add(0, pid[keep].long(), torch.ones(int(keep.sum())))
    ref = sums / cnt[:, None]
    assert torch.allclose(vf, ref, atol=1e-6)
    (ref * w).sum().backward()
    assert torch.allclose(feats.grad, ref_feats.grad, atol=1e-6)
    assert torch.equal(feats.grad[:5], torch.zeros((5, 5)))


def test_affinity_helpers_leave_the_process_alone_without_a_gpu(monkeypatch):
    """affinity.pin_to_gpu_node: no GPU / unreadable topology / GPN_NO_PIN=1 -> affinity untouched, no error"""
    import os
    from gapartnet_amd import affinity
    assert affinity._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert affinity._parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    assert affinity.pin_to_gpu_node(0) is None  # no GPU here: the PCI address cannot be resolved
    monkeypatch.setenv("GPN_NO_PIN", "1")
    monkeypatch.setattr(affinity, "gpu_numa_node", lambda i=0: 0)
    assert affinity.pin_to_gpu_node(0) is None
    assert os.sched_getaffinity(0) == before


def test_persistent_gradient_buffer_ownership():
    """network/net_exec.NetProgram.grad_buffer: an explicit hand-over contract (round 5; no reference-count heuristics).  The
    flat per-network gradient buffer is handed out again only after the consumer of the previous backward pass's gradients
    RELEASED them (FusedAdam.step / zero_grad, the Trainer, net_exec.release_gradients(model)); without a release it is
    REPLACED - never overwritten - and the replacement is counted (retired_total)."""
    from gapartnet_amd.network import net_exec
    from gapartnet_amd.network.backbone import SparseUNet
    from gapartnet_amd.optim import FusedAdam
    import functools
    unet = SparseUNet.build(16, [16, 32], 1, functools.partial(torch.nn.BatchNorm1d, eps=1e-4, momentum=0.1), without_stem=True)
    prog = net_exec.program_for(unet)
    assert prog is not None
    params = prog.params()
    dev = torch.device("cpu")
    opt = FusedAdam(list(unet.parameters()), lr=1e-3)
    assert net_exec.programs_of(unet.parameters()) == [prog]
    ids = []
    for step in range(4):  # the training loop's sequence: zero_grad, backward assigns .grad, optimizer step
        opt.zero_grad(set_to_none=True)
        flat, views = prog.grad_buffer(dev, params, fresh=False)
        flat.zero_()
        ids.append(flat.data_ptr())
        net_exec._hand_over(prog, params, views, flat, False)
        del flat, views
        opt.step()  # (CPU tensors: torch's own Adam; the release is this class's)
    assert len(set(ids)) == 1 and prog.retired_total == 0, (ids, prog.retired_total)
    # a loop that never releases (a foreign optimizer): every pass gets a buffer of its own, the old one stays with its holders
    kept = params[3].grad
    kept.fill_(3.0)
    for p in params:
        p.grad = None
    flat, views = prog.grad_buffer(dev, params, fresh=False)   # released by the last opt.step(): still the persistent pair
    assert flat.data_ptr() == ids[0] and prog.retired_total == 0
    del flat, views
    flat, views = prog.grad_buffer(dev, params, fresh=False)   # nobody released the pass before
    assert flat.data_ptr() != ids[0] and prog.retired_total == 1
    kept_value = kept.clone()
    flat.fill_(7.0)
    assert torch.equal(kept, kept_value), "the retired buffer stays untouched with its holder"
    second = flat.data_ptr()
    del flat, views, kept
    # the module-level acknowledgement for any other training loop
    net_exec.release_gradients(unet)
    assert prog.grad_buffer(dev, params, fresh=False)[0].data_ptr() == second and prog.retired_total == 1
