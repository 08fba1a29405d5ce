"""csrc/proposals.hip (the proposal stage as one library call with one host read) against the torch formulation of the same
stage on the same GPU inputs - which is itself pinned to the reference's model.py by tests/test_golden_pipeline.py.  Every
integer output must be equal, the voxel features (ordered means) too; gradients w.r.t. the point features bit-equal."""
import numpy as np
import pytest
import torch

from gapartnet_amd.smoke import make_batch, make_model
from gapartnet_amd.structure.point_cloud import PointCloud
from tests.golden import recipe

pytestmark = pytest.mark.gpu
JITTER = (torch.tensor([0.3, 0.6, 0.1]), torch.tensor([0.5, 0.2, 0.9]))
FIELDS = ("valid_mask", "valid_indices", "sorted_indices", "pt_xyz", "batch_indices", "proposal_offsets", "proposal_indices",
          "num_points_per_proposal", "sem_preds", "instance_labels")


def _stage(model, batch, feats, sem_preds, offsets, fused, with_labels=True):
    model.use_fused_proposals = fused
    f = feats.clone().requires_grad_(True)
    vt, pid, props = model.proposal_clustering_and_revoxelize(
        pt_xyz=batch.points[:, :3], batch_indices=batch.batch_indices, pt_features=f, sem_preds=sem_preds, offset_preds=offsets,
        instance_labels=batch.instance_labels if with_labels else None, batch_size=batch.batch_size)
    return f, vt, pid, props


@pytest.mark.parametrize("n_scenes,n_points,with_labels", [(2, 4000, True), (8, 20000, True), (3, 6000, False), (1, 50000, True)])
def test_fused_stage_equals_the_torch_formulation(cuda, n_scenes, n_points, with_labels):
    model = make_model((0, 0), channels=[16, 32])
    model.load_state_dict(recipe.name_keyed_state(model))
    model = model.to(cuda).eval()
    model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
    batch = PointCloud.collate([pc.to(cuda) for pc in make_batch(n_scenes, n_points, seed0=900 + n_points)], voxel_size=(0.01,) * 3)
    g = torch.Generator().manual_seed(n_points)
    N = batch.points.shape[0]
    feats = torch.randn(N, 16, generator=g).to(cuda)
    # spatially coherent fake predictions: class from a coarse grid cell, offsets of a few millimetres
    cell = torch.floor(batch.points[:, :3] * 4).long()
    sem_preds = ((cell[:, 0] * 7 + cell[:, 1] * 3 + cell[:, 2]) % 10).clamp(min=0)
    offsets = (0.01 * torch.randn(N, 3, generator=g)).to(cuda)
    out = {}
    for fused in (False, True):
        f, vt, pid, props = _stage(model, batch, feats, sem_preds, offsets, fused, with_labels)
        assert props is not None
        w = torch.linspace(-1, 1, vt.features.numel(), device=cuda).view_as(vt.features)
        (vt.features * w).sum().backward()
        out[fused] = (f.grad, vt, pid, props)
    (g0, vt0, pid0, p0), (g1, vt1, pid1, p1) = out[False], out[True]
    for name in FIELDS:
        a, b = getattr(p0, name), getattr(p1, name)
        if a is None:
            assert b is None, name
            continue
        assert a.dtype == b.dtype and a.shape == b.shape, (name, a.dtype, b.dtype, a.shape, b.shape)
        assert torch.equal(a, b), name
    assert torch.equal(p1.point_indices, p0.valid_indices[p0.sorted_indices])
    assert torch.equal(vt0.indices, vt1.indices) and vt0.batch_size == vt1.batch_size and vt0.spatial_shape == vt1.spatial_shape
    assert torch.equal(pid0, pid1)
    assert torch.equal(vt0.features, vt1.features), "ordered means: bit-equal"
    for a, b in zip(vt0.point_csr, vt1.point_csr):
        assert torch.equal(a, b)
    assert torch.equal(g0, g1), "d point features"


def test_fused_stage_without_proposals(cuda):
    model = make_model((0, 0), channels=[16, 32]).to(cuda).eval()
    model.sync_free_proposals = False  # (the blocking form: a stage that finds nothing says so; tests/test_gpu_sync_free.py has the other)
    batch = PointCloud.collate([pc.to(cuda) for pc in make_batch(2, 3000)], voxel_size=(0.01,) * 3)
    N = batch.points.shape[0]
    feats = torch.zeros(N, 16, device=cuda)
    offsets = torch.zeros(N, 3, device=cuda)
    # nothing valid at all
    none = _stage(model, batch, feats, torch.zeros(N, dtype=torch.int64, device=cuda), offsets, True)
    assert none[1] is None and none[3] is None
    # valid points, but every one its own class neighbourhood: clusters stay below min_num_points_per_proposal
    sem = (torch.arange(N, device=cuda) % 9) + 1
    model.ball_query_radius = 1e-4
    none = _stage(model, batch, feats, sem, offsets, True)
    assert none[3] is None


@pytest.mark.parametrize("n_props,seed", [(1, 0), (37, 1), (600, 2)])
def test_fused_npcs_loss_equals_the_torch_formulation(cuda, n_props, seed):
    """csrc/losses.hip gpn_npcs_loss_* against compute_npcs_loss_grouped (itself pinned to the reference's per-group
    compute_npcs_loss by tests/golden/npcs_loss.npz): value and d logits, all five symmetry types, proposals without any
    valid point, points whose class is wrong / whose target is zero"""
    from gapartnet_amd.structure.instances import Instances
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(5, 200, (n_props,), generator=g)
    M = int(sizes.sum())
    offsets = torch.zeros(n_props + 1, dtype=torch.int32)
    offsets[1:] = sizes.cumsum(0)
    prop = torch.repeat_interleave(torch.arange(n_props), sizes)
    cls_of_prop = torch.randint(1, 10, (n_props,), generator=g)
    sem_preds = cls_of_prop[prop].to(torch.int32)
    sem_labels = sem_preds.long().clone()
    wrong = torch.rand(M, generator=g) < 0.2
    sem_labels[wrong] = (sem_labels[wrong] % 9) + 1
    gt = torch.rand(M, 3, generator=g) - 0.5
    gt[torch.rand(M, generator=g) < 0.15] = 0.0
    if n_props > 5:  # a proposal with no valid point at all
        gt[offsets[3]:offsets[4]] = 0.0
    logits = torch.randn(M, 27, generator=g)
    results = []
    for fused in (False, True):
        model = make_model((0, 0), channels=[16, 32]).to(cuda).train()
        x = logits.clone().to(cuda).requires_grad_(True)
        props = Instances(sem_preds=sem_preds.to(cuda), sem_labels=sem_labels.to(cuda), proposal_offsets=offsets.to(cuda),
                          proposal_indices=prop.to(cuda))
        if not fused:
            from gapartnet_amd import hip_ops
            saved = hip_ops.npcs_loss_fwd
            del hip_ops.npcs_loss_fwd
        try:
            loss = model.loss_proposal_npcs(x, gt.to(cuda), props)
        finally:
            if not fused:
                hip_ops.npcs_loss_fwd = saved
        (loss * 1.7).backward()
        results.append((float(loss), x.grad.cpu(), props.npcs_valid_mask.cpu()))
    (l0, g0, v0), (l1, g1, v1) = results
    assert torch.equal(v0, v1)
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (l0, l1)
    assert torch.allclose(g0, g1, rtol=1e-4, atol=1e-7), float((g0 - g1).abs().max())


@pytest.mark.parametrize("seed,full", [(0, 28.0), (1, 28.0), (2, 7.0)])
def test_revoxelisation_without_a_sort_equals_the_sorting_voxeliser(cuda, seed, full):
    """gpn_proposals_revoxelize (one workgroup per proposal: LDS bitmap, popcount ranks, stable placement) against
    gpn_voxelize_ex (keys + radix sort) on the same grouped points: unique cells, voxel of every point, point order, CSR -
    with points outside their grid, many points per cell, proposals of 1 ... 3000 points, empty tail rows, unused proposal
    slots"""
    import ctypes
    from gapartnet_amd import _C, hip_ops as H
    rng = np.random.default_rng(seed)
    sizes = [1, 5, 7, 3000, 257, 256, 255, 64, 1000] + [int(n) for n in rng.integers(5, 400, size=40)]
    P, T = len(sizes), int(sum(sizes))
    P_ub, T2 = P + 23, T + 517
    off = np.full(P_ub + 1, T, np.int64)
    off[:P + 1] = np.concatenate([[0], np.cumsum(sizes)])
    xyz = rng.uniform(-0.5, full + 0.5, size=(T2, 3)).astype(np.float32)     # some points leave their grid
    xyz[off[3]:off[3] + 1500] = np.float32(3.25)                               # 1500 points in one cell
    xyz[off[8]:off[9]] = np.floor(xyz[off[8]:off[9]] / 4) * 4 + 0.5           # few cells, many points each
    xyz[T:] = -1.0
    scaled = torch.from_numpy(xyz).to(cuda)
    counts = torch.zeros(8, dtype=torch.int64, device=cuda)
    counts[1], counts[2] = T, P
    off32 = torch.from_numpy(off.astype(np.int32)).to(cuda)
    vc = torch.full((T2, 3), -9, dtype=torch.int32, device=cuda)
    vseg = torch.full((T2,), -9, dtype=torch.int32, device=cuda)
    pid = torch.full((T2,), -9, dtype=torch.int32, device=cuda)
    order = torch.full((T2,), -9, dtype=torch.int32, device=cuda)
    vstart = torch.full((T2 + 1,), -9, dtype=torch.int32, device=cuda)
    L = _C.lib()
    ws = torch.empty((int(L.gpn_proposals_revoxelize_ws_bytes(H.i64(P_ub))),), dtype=torch.uint8, device=cuda)
    rc = L.gpn_proposals_revoxelize(H.ptr(scaled), H.ptr(off32), H.ptr(counts), H.i64(T2), H.i64(P_ub), ctypes.c_float(full), H.ptr(vc),
                                    H.ptr(vseg), H.ptr(pid), H.ptr(order), H.ptr(vstart), H.ptr(ws), H.szt(ws.numel()),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.gpn_last_error()
    d = int(full) + 1
    zero, top = torch.zeros(3, device=cuda), torch.full((3,), full, device=cuda)
    _, rvc, rvseg, rpid, rorder, rvstart = H.voxelize(scaled, scaled, torch.from_numpy(off).to(cuda), zero, top, (1.0, 1.0, 1.0),
                                                      (d, d, d), want_csr=True)
    V = rvc.shape[0]
    assert int(counts[3]) == V and V > 100
    assert int((rpid < 0).sum()) > T2 - T, "the case needs points outside their grid inside proposals"
    assert torch.equal(vc[:V], rvc) and torch.equal(vseg[:V], rvseg)
    assert torch.equal(pid, rpid)
    assert torch.equal(order, rorder)
    assert torch.equal(vstart[:V + 1], rvstart)


def _eval_model(cuda, seed_offset=0):
    model = make_model((0, 0))
    model.load_state_dict(recipe.name_keyed_state(model))  # non-trivial BatchNorm statistics: several classes, real proposals
    model = model.to(cuda).eval()
    model.revoxelize_jitter = tuple(j.to(cuda) for j in JITTER)
    model._log_sink = lambda name, value, bs, sync: None
    return model


KEPT_FIELDS = ("score_preds", "pt_sem_classes", "batch_indices", "instance_sem_labels", "ious", "proposal_offsets", "valid_mask")


@pytest.mark.parametrize("n_scenes,n_points", [(2, 5000), (4, 20000), (8, 20000)])
@pytest.mark.parametrize("thresholds", [(0.09, 3, 0.3), (0.5, 20, 0.05), (0.0, 0, 0.9)])
def test_fused_post_processing_equals_the_torch_formulation(cuda, n_scenes, n_points, thresholds):
    """gpn_proposals_postprocess (csrc/postprocess.hip: score filter, sparse intersections through member_slot, NMS in rounds, one
    compaction) against the torch formulation of filter_invalid_proposals + apply_nms (network/grouping_utils.py - itself
    pinned to the reference by tests/test_golden_pipeline.py) on the proposals of a validation step: every field validation_step
    keeps must be EQUAL (integers and floats alike: scores and IoUs are selected, never recomputed)."""
    model = _eval_model(cuda)
    model.sync_free_proposals = False
    model.val_score_threshold, model.val_min_num_points_per_proposal, model.val_nms_iou_threshold = thresholds
    batch = [pc.to(cuda) for pc in make_batch(n_scenes, n_points, seed0=1300 + n_points)]
    with torch.no_grad():
        _, _, proposals, _ = model._training_or_validation_step(batch, 0, "val", want_npcs_preds=False)
        assert proposals is not None and proposals.score_preds is not None
        fused = model._post_process_kept(proposals)
        assert fused is not None, "the fused form must apply to the fused proposal stage's output"
        ref = model._post_process(proposals)
    assert ref.score_preds.shape[0] < proposals.score_preds.shape[0] or thresholds[0] == 0.0
    for name in KEPT_FIELDS:
        a, b = getattr(ref, name), getattr(fused, name)
        assert a.dtype == b.dtype and a.shape == b.shape, (name, a.dtype, b.dtype, tuple(a.shape), tuple(b.shape))
        assert torch.equal(a, b), name


def test_validation_step_without_a_host_read_equals_the_blocking_step(cuda):
    """round 5: validation steps run the device-counted proposal stage too (row counts as device counters, one host read at the
    end of the step: the post-processing's two counts) - what the step keeps for the epoch-end AP must equal the blocking
    step's, the stale plan of the first device-counted step included"""
    batches = [[pc.to(cuda) for pc in make_batch(4, 20000, seed0=1500 + 10 * j)] for j in range(3)]
    kept = {}
    for sync_free, defer in ((False, False), (True, False), (True, True)):
        model = _eval_model(cuda)
        model.sync_free_proposals = sync_free
        model.defer_validation_outputs = defer  # (what the Trainer's evaluation loop sets: a step's read a step later)
        outs = []
        with torch.no_grad():
            for i, batch in enumerate(batches):
                out = model.validation_step(batch, i, 0)
                outs.append(out)
                assert defer == (type(out[2]).__name__ == "_PendingKept")
        if sync_free:
            assert model._prop_pending, "validation steps after the first one ran without reading the proposal counts"
        if defer:
            model._resolve_pending_outputs()
            outs = list(model.validation_step_outputs[0])
            assert len(outs) == len(batches) and all(type(o[2]).__name__ == "Instances" for o in outs)
        kept[(sync_free, defer)] = outs
    pairs = list(zip(kept[(False, False)], kept[(True, False)])) + list(zip(kept[(False, False)], kept[(True, True)]))
    for (ids_a, seg_a, a), (ids_b, seg_b, b) in pairs:
        assert ids_a == ids_b and torch.equal(seg_a.sem_preds, seg_b.sem_preds)
        assert (a is None) == (b is None)
        for name in KEPT_FIELDS:
            x, y = getattr(a, name), getattr(b, name)
            assert x.dtype == y.dtype and x.shape == y.shape, (name, x.dtype, y.dtype, tuple(x.shape), tuple(y.shape))
            if x.dtype.is_floating_point:
                assert torch.allclose(x, y, rtol=1e-5, atol=1e-6), name  # (kernel variants follow the plan: last bits)
            else:
                assert torch.equal(x, y), name


def test_post_processing_with_the_status_bytes_in_the_workspace(cuda):
    """round 6: the NMS kernel's status table is sized by the LIVE proposal count - LDS up to 131072, the
    workspace beyond - and any bound is accepted.  Here: the workspace form (threshold lowered to 0) equals the LDS form field
    by field, and a validation-sized bound above 131072 (32 scenes x 20k points: 256 001) with a few hundred live proposals no
    longer returns GPN_ERR_ARG."""
    import ctypes
    from gapartnet_amd import _C, hip_ops as H
    L = _C.lib()
    L.gpn_proposals_postprocess_lds_proposals.argtypes = [ctypes.c_int64]
    L.gpn_proposals_postprocess_lds_proposals.restype = ctypes.c_int64
    model = _eval_model(cuda)
    model.sync_free_proposals = False
    model.val_score_threshold, model.val_min_num_points_per_proposal, model.val_nms_iou_threshold = 0.0, 0, 0.3
    batch = [pc.to(cuda) for pc in make_batch(4, 20000, seed0=21300)]
    with torch.no_grad():
        _, _, proposals, _ = model._training_or_validation_step(batch, 0, "val", want_npcs_preds=False)
        ref = model._post_process_kept(proposals)
        prev = L.gpn_proposals_postprocess_lds_proposals(0)
        try:
            ws_form = model._post_process_kept(proposals)
        finally:
            L.gpn_proposals_postprocess_lds_proposals(prev)
    assert prev == 131072
    for name in KEPT_FIELDS:
        assert torch.equal(getattr(ref, name), getattr(ws_form, name)), name
    # the same proposals inside buffers of a 256 001-proposal bound, the live count on the device
    P = int(proposals.score_preds.shape[0])
    bound = 256001
    pad = lambda t, n, fill=0: torch.cat([t, torch.full((n - t.shape[0],) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)])
    live = torch.tensor([P], dtype=torch.int64, device=cuda)
    out = H.proposals_postprocess(pad(proposals.score_preds, bound), pad(proposals.num_points_per_proposal, bound),
                                  pad(proposals.proposal_offsets, bound + 1), proposals.point_indices, proposals.proposal_indices,
                                  proposals.member_slot, 0.0, 0, 0.3, rows=H.DevCount(live, P))
    assert out is not None
    ids, new_offsets, src_row = out
    assert torch.equal(new_offsets, ref.proposal_offsets) and ids.shape[0] == ref.score_preds.shape[0]
    assert torch.equal(proposals.score_preds.index_select(0, ids), ref.score_preds)


@pytest.mark.parametrize("defer", [False, True])
def test_validation_step_falls_back_when_a_neighbour_table_overflows(cuda, monkeypatch, defer):
    """round 6: a proposal that shares points with more proposals than the post-processing kernel's tables
    hold makes the fused call report an overflow; a device-counted validation step then cuts its tensors to their live sizes and
    runs the torch formulation (round 5 raised).  The overflow is forced here (the call's result replaced by the overflow answer);
    what the step keeps must equal the fused call's."""
    from gapartnet_amd import hip_ops as H
    batches = [[pc.to(cuda) for pc in make_batch(4, 20000, seed0=1500 + 10 * j)] for j in range(2)]
    kept = {}
    for forced in (False, True):
        model = _eval_model(cuda)
        model.sync_free_proposals = True
        model.defer_validation_outputs = defer
        if forced:
            real = H.proposals_postprocess

            def overflowing(*a, defer=False, **kw):
                out = real(*a, defer=defer, **kw)
                if defer:
                    out.result = lambda: None
                    return out
                return None
            monkeypatch.setattr(H, "proposals_postprocess", overflowing)
        with torch.no_grad():
            for i, batch in enumerate(batches):
                model.validation_step(batch, i, 0)
            model._resolve_pending_outputs()
        if forced:
            monkeypatch.undo()
        assert model._prop_pending, "the second step ran device-counted"
        kept[forced] = list(model.validation_step_outputs[0])
    for (_, _, a), (_, _, b) in zip(kept[False], kept[True]):
        for name in KEPT_FIELDS:
            x, y = getattr(a, name), getattr(b, name)
            assert x.shape == y.shape and torch.equal(x, y), name


def test_validation_step_with_batchnorm_in_the_conv_epilogues_equals_batchnorm_launches(cuda):
    """round 6, the whole model at config 2's size (4 x 20k points, eval mode): the inference pass (every BatchNorm applied by the conv
    launch that produces its input, GPN_NET_INFERENCE) against the same validation steps with the BatchNorm launches
    (gpn_net_bn_fusion(0)): semantic predictions and every field the step keeps are EQUAL - the fold is the stand-alone pass's
    arithmetic per element"""
    from gapartnet_amd import _C
    lib = _C.lib()
    batches = [[pc.to(cuda) for pc in make_batch(4, 20000, seed0=1700 + 10 * j)] for j in range(2)]
    kept = {}
    for fused in (1, 0):
        prev = lib.gpn_net_bn_fusion(fused)
        try:
            model = _eval_model(cuda)
            outs = []
            with torch.no_grad():
                for i, batch in enumerate(batches):
                    outs.append(model.validation_step(batch, i, 0))
            kept[fused] = outs
        finally:
            lib.gpn_net_bn_fusion(prev)
    for (ids_a, seg_a, a), (ids_b, seg_b, b) in zip(kept[1], kept[0]):
        assert ids_a == ids_b and torch.equal(seg_a.sem_preds, seg_b.sem_preds)
        assert a is not None and b is not None
        for name in KEPT_FIELDS:
            x, y = getattr(a, name), getattr(b, name)
            assert x.shape == y.shape and torch.equal(x, y), name
