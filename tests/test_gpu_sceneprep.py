"""gpn_scene_prepare (csrc/sceneprep.hip, include/gpn.h section SP): label compaction, augmentation and the per-instance statistics
of a batch of RAW scenes in three launches, against the torch formulation of dataset/device_pipeline.py (which the CPU tests pin
against the per-scene functions of dataset/gapartnet.py = the reference's loader, and tests/test_golden_loader.py against the
reference's own outputs - that golden test runs through this call on the GPU).  Integer outputs equal; coordinates and means
within an ulp (a float64 dot product / an order-independent fixed-point sum against float64 atomics)."""
import numpy as np
import pytest
import torch

from gapartnet_amd.dataset import device_pipeline as dp
from gapartnet_amd.structure.point_cloud import PointCloud

pytestmark = pytest.mark.gpu
VOXEL = (0.01, 0.01, 0.01)


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _scenes(rng, sizes, ids_per_scene, dev, sem_dtype=torch.int64, id_scale=1):
    out = []
    for s, (n, k) in enumerate(zip(sizes, ids_per_scene)):
        pts = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        pool = np.sort(rng.choice(np.arange(0, max(4 * k, 8)) * id_scale + s, size=k, replace=False))
        ins = pool[rng.integers(0, k, n)].astype(np.int32)
        ins[rng.random(n) < 0.3] = -100  # points on no instance
        ins[:k] = pool  # every id present
        sem = rng.integers(0, 10, n)
        out.append(PointCloud(pc_id=f"c_{s}_00_000", obj_cat=s, points=torch.from_numpy(pts).to(dev),
                              sem_labels=torch.from_numpy(sem).to(sem_dtype).to(dev), instance_labels=torch.from_numpy(ins).to(dev),
                              gt_npcs=torch.from_numpy(pts[:, :3].copy()).to(dev)))
    return out


def _both(raw, aug):
    res = []
    for fused in (True, False):
        dp.FUSED = fused
        np.random.seed(7)
        try:
            res.append(dp.prepare_batch(raw, VOXEL, dict(aug) if aug else None, voxels=False))
        finally:
            dp.FUSED = True
    return res


def _ulp_close(a, b, ulps=1):
    a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)))


CASES = {
    "bench batch": ([20000] * 8, [12, 7, 30, 1, 18, 9, 25, 4], None, torch.int64, 1),
    "ragged scenes, augmented": ([5000, 123, 20000, 1, 777], [5, 3, 40, 1, 11],
                                 dict(pos_jitter=0.1, color_jitter=0.3, flip_prob=0.3, rotate_prob=0.3), torch.int64, 1),
    "sparse large ids, int16 labels": ([3000, 4000], [200, 256], dict(pos_jitter=0.05, color_jitter=0., flip_prob=0., rotate_prob=0.),
                                       torch.int16, 2_000_000),
}


@pytest.mark.parametrize("name", list(CASES))
def test_fused_scene_preparation_equals_the_torch_formulation(cuda, name):
    sizes, ks, aug, sem_dtype, id_scale = CASES[name]
    raw = _scenes(np.random.default_rng(len(name)), sizes, ks, cuda, sem_dtype, id_scale)
    if sem_dtype != torch.int64:  # (the packed cache hands int16 labels to the library; the torch formulation takes int64)
        from gapartnet_amd import hip_ops as H
        pts = torch.cat([p.points for p in raw]); sem = torch.cat([p.sem_labels for p in raw]); ins = torch.cat([p.instance_labels for p in raw])
        off = dp.scene_offsets(sizes, cuda)
        a = H.scene_prepare(pts, sem, ins, off)
        b = H.scene_prepare(pts, sem.long(), ins, off)
        for k in ("instance_sem_labels", "num_points_per_instance", "instance_labels"):
            assert torch.equal(a[k], b[k]), k
        raw = [PointCloud(pc_id=p.pc_id, obj_cat=p.obj_cat, points=p.points, sem_labels=p.sem_labels.long(),
                          instance_labels=p.instance_labels, gt_npcs=p.gt_npcs) for p in raw]
    fused, plain = _both(raw, aug)
    assert fused.num_instances == plain.num_instances == ks
    for f in ("batch_indices", "instance_labels", "sem_labels", "num_points_per_instance", "instance_sem_labels"):
        a, b = getattr(fused, f), getattr(plain, f)
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), f
    assert _ulp_close(fused.points, plain.points), "augmented points"
    if not aug:
        assert torch.equal(fused.points, plain.points)
    ra, rb = fused.instance_regions, plain.instance_regions
    assert _ulp_close(ra[:, :3], rb[:, :3], 2), "instance means"
    if not aug:  # (same points -> the same minima / maxima exactly)
        assert torch.equal(ra[:, 3:], rb[:, 3:]), "instance min / max"
    else:
        assert _ulp_close(ra[:, 3:], rb[:, 3:]), "instance min / max"
    # twice the same: the fixed-point sums do not depend on the order of the atomics
    dp.FUSED = True
    np.random.seed(7)
    again = dp.prepare_batch(raw, VOXEL, dict(aug) if aug else None, voxels=False)
    assert torch.equal(again.instance_regions, ra) and torch.equal(again.points, fused.points)


def test_more_instances_than_the_tables_hold_take_the_torch_formulation(cuda):
    from gapartnet_amd import hip_ops as H
    raw = _scenes(np.random.default_rng(3), [4000, 6000], [10, 300], cuda)
    pts = torch.cat([p.points for p in raw]); sem = torch.cat([p.sem_labels for p in raw]); ins = torch.cat([p.instance_labels for p in raw])
    assert H.scene_prepare(pts, sem, ins, dp.scene_offsets([4000, 6000], cuda)) is None
    fused, plain = _both(raw, None)
    assert fused.num_instances == plain.num_instances == [10, 300]
    assert torch.equal(fused.instance_labels, plain.instance_labels)
    assert torch.equal(fused.instance_regions, plain.instance_regions)


def test_a_scene_without_instances_is_reported(cuda):
    raw = _scenes(np.random.default_rng(5), [500, 600], [3, 2], cuda)
    raw[1].instance_labels = torch.full_like(raw[1].instance_labels, -100)
    with pytest.raises(ValueError, match="without a labelled instance"):
        dp.prepare_batch(raw, VOXEL, None, voxels=False)
