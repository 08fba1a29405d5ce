"""Drop-in check against the reference's OWN glue code (only where /root/reference exists, i.e. the build container;
skipped on the GPU box): the reference's `network/backbone.py` and `network/grouping_utils.py` are imported unmodified
with this repo's mirrors registered under the names `spconv.pytorch` / `epic_ops.*`, and must give the same results as
this repo's re-implementation of that glue (same weights, same inputs, oracle raw-op backend)."""
import functools
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

REF = "/root/reference/gapartnet"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


@pytest.fixture()
def reference_modules():
    import gapartnet_amd.epic_ops as eo
    import gapartnet_amd.spconv
    import gapartnet_amd.spconv.pytorch
    from gapartnet_amd import backend
    from oracle import torch_ops
    saved = dict(sys.modules)
    sys.modules["spconv"] = gapartnet_amd.spconv
    sys.modules["spconv.pytorch"] = gapartnet_amd.spconv.pytorch
    sys.modules["epic_ops"] = eo
    for sub in ("voxelize", "ball_query", "ccl", "reduce", "iou", "nms"):
        sys.modules[f"epic_ops.{sub}"] = getattr(eo, sub)
    sys.path.insert(0, REF)
    try:
        with backend.using(torch_ops):
            import network.backbone as ref_backbone
            import network.grouping_utils as ref_grouping
            yield ref_backbone, ref_grouping
    finally:
        sys.path.remove(REF)
        for name in list(sys.modules):
            if name not in saved:
                del sys.modules[name]
        sys.modules.update(saved)


def test_reference_backbone_runs_on_the_mirrors_and_matches(reference_modules):
    ref_backbone, _ = reference_modules
    from gapartnet_amd.network.backbone import SparseUNet
    from gapartnet_amd.spconv import pytorch as spconv
    from tests import synth
    norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
    torch.manual_seed(0)
    mine = SparseUNet.build(6, [16, 32, 48], 2, norm_fn)
    theirs = ref_backbone.SparseUNet.build(6, [16, 32, 48], 2, norm_fn)
    assert list(theirs.state_dict().keys()) == list(mine.state_dict().keys())
    theirs.load_state_dict(mine.state_dict())
    rng = np.random.default_rng(0)
    idx = synth.surface_indices(rng, 2, [40, 40, 40], 900)
    feats = torch.from_numpy(rng.normal(size=(idx.shape[0], 6)).astype(np.float32))
    outs = []
    for net in (mine, theirs):
        x = spconv.SparseConvTensor(feats, torch.from_numpy(idx), [40, 40, 40], 2)
        outs.append(net(x).features)
    # same module tree and weights; this repo fuses BatchNorm + residual + ReLU into one op, so equality is to rounding
    assert torch.allclose(outs[0], outs[1], atol=1e-5, rtol=1e-5)


def test_reference_grouping_glue_matches(reference_modules):
    _, ref_grouping = reference_modules
    from gapartnet_amd.network import grouping_utils as G
    from tests import synth
    rng = np.random.default_rng(1)
    pts, batch = synth.clustered_points(rng, 2, 700, n_clusters=5)
    offs = torch.tensor([0, 700, 1400], dtype=torch.int32)
    sem = torch.from_numpy(rng.integers(1, 3, 1400).astype(np.int32))
    mine = G.cluster_proposals(torch.from_numpy(pts), torch.from_numpy(batch), offs, sem, 0.05, 30)
    theirs = ref_grouping.cluster_proposals(torch.from_numpy(pts), torch.from_numpy(batch), offs, sem, 0.05, 30)
    assert torch.equal(mine[0], theirs[0])  # labels; the reference's sort is unstable, so compare membership only:
    assert torch.equal(torch.sort(mine[1])[0], torch.sort(theirs[1])[0])
    # re-voxelisation with a pinned RNG: the reference draws torch.rand(3) twice
    labels, order = mine
    _, prop_idx, sizes = torch.unique_consecutive(labels, return_inverse=True, return_counts=True)
    offsets = G.offsets_from_counts(sizes)
    xyz = torch.from_numpy(pts)[order]
    feats = torch.from_numpy(rng.normal(size=(1400, 4)).astype(np.float32))
    torch.manual_seed(5)
    a = ref_grouping.segmented_voxelize(xyz, feats, offsets, prop_idx, sizes, 28, 50)
    torch.manual_seed(5)
    b = G.segmented_voxelize(xyz, feats, offsets, prop_idx, sizes, 28, 50)
    for t_ref, t_mine in zip(a, b):
        assert torch.equal(t_ref, t_mine)
    ious = torch.tensor([0.1, 0.3, 0.6, 0.8])
    assert torch.equal(ref_grouping.get_gt_scores(ious), G.get_gt_scores(ious))
