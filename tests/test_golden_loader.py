"""The loader side of the hot path against the reference's OWN loader functions.

tests/golden/loader.npz holds what /root/reference/gapartnet/dataset/gapartnet.py:85-229 (load_data,
compact_instance_labels, apply_augmentations, generate_inst_info, apply_voxelization — imported unmodified by
tests/golden/make_golden_pipeline.py in the build container) produce for seeded ``.pth`` scenes, plain and augmented
with gapartnet.yaml's augmentation settings.  Compared here:
  * the per-scene functions of gapartnet_amd.dataset.gapartnet (CPU tensors, oracle voxeliser)            [not gpu]
  * the per-batch device pipeline (dataset/device_pipeline.py + batched voxelisation), CPU tensors        [not gpu]
  * the same on the GPU through libgpn_hip.so                                                             [gpu]
Integers (instance ids, counts, labels, voxel coordinates, point->voxel map) must be equal; the augmentation replays
the reference's draw order from the same numpy seed.
"""
import os

import numpy as np
import pytest
import torch

from gapartnet_amd import backend
from gapartnet_amd.dataset import device_pipeline as dp
from gapartnet_amd.dataset import gapartnet as ds
from gapartnet_amd.structure.point_cloud import PointCloud
from tests.golden import recipe

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VOXEL = (0.01, 0.01, 0.01)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "loader.npz"))


def _scene_file(tmp_path, seed):
    path = os.path.join(str(tmp_path), f"Box_{seed}_00_000.pth")
    torch.save(tuple(recipe.scene_arrays(seed, recipe.LOADER_POINTS)), path)
    return path


def _check_scene(gold, pre, pc, vox_exact=True):
    g = lambda k: gold[pre + k]  # noqa: E731
    assert np.array_equal(pc.instance_labels.cpu().numpy(), g("instance_labels"))
    assert int(pc.num_instances) == int(g("num_instances"))
    assert np.array_equal(pc.num_points_per_instance.cpu().numpy(), g("num_points_per_instance"))
    assert np.array_equal(pc.instance_sem_labels.cpu().numpy(), g("instance_sem_labels"))
    assert np.allclose(pc.points.cpu().numpy(), g("points"), rtol=0, atol=2e-6)
    reg = pc.instance_regions.cpu().numpy()
    assert np.allclose(reg[:, :3], g("instance_regions")[:, :3], rtol=0, atol=3e-6)       # mean: summation order
    assert np.allclose(reg[:, 3:], g("instance_regions")[:, 3:], rtol=0, atol=2e-6)       # min | max (quirk: this order)


def _check_voxels(gold, pre, coords, feats, pid, rng):
    assert np.array_equal(coords, gold[pre + "voxel_coords"]), "voxel coordinates: bit-exact, same (x,y,z) order"
    assert np.array_equal(pid, gold[pre + "pc_voxel_id"])
    assert list(rng) == gold[pre + "voxel_coords_range"].tolist()
    assert np.allclose(feats, gold[pre + "voxel_features"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("seed", recipe.LOADER_SEEDS)
def test_per_scene_loader_functions_match_the_reference(gold, tmp_path, seed):
    from oracle import torch_ops
    pc = ds.load_data(_scene_file(tmp_path, seed))
    assert pc.obj_cat == int(gold[f"s{seed}_obj_cat"])
    pc = ds.compact_instance_labels(ds.downsample(pc, max_points=20000))
    np.random.seed(seed)
    aug = ds.apply_augmentations(pc, **recipe.AUG)
    for tag, scene in (("plain", pc), ("aug", aug)):
        full = ds.generate_inst_info(scene).to_tensor()
        pre = f"s{seed}_{tag}_"
        if tag == "plain":
            assert np.array_equal(full.points.numpy(), gold[pre + "points"])
        _check_scene(gold, pre, full)
        if tag == "aug":  # voxelise exactly the reference's augmented points, so that coordinates must be equal
            full.points = torch.from_numpy(gold[pre + "points"])
        with backend.using(torch_ops):
            vox = ds.apply_voxelization(full, voxel_size=VOXEL)
        _check_voxels(gold, pre, vox.voxel_coords.numpy(), vox.voxel_features.numpy(), vox.pc_voxel_id.numpy(),
                      vox.voxel_coords_range)


def test_augmentation_branches_are_covered(gold):
    b = gold["branches"]
    assert b[:, 0].any() and not b[:, 0].all() and b[:, 1].any() and not b[:, 1].all()
    # the flip_prob-gates-rotate quirk (dataset/gapartnet.py:103-104): a seed whose second uniform draw lies between
    # rotate_prob and flip_prob would tell the two gates apart only if they differed; with yaml's 0.3 / 0.3 they coincide,
    # so the quirk is pinned by a direct call with different probabilities instead:
    pc = PointCloud(pc_id="x", points=np.random.default_rng(0).normal(size=(50, 6)).astype(np.float32))
    np.random.seed(3)
    out = ds.apply_augmentations(pc, pos_jitter=0.0, flip_prob=1e-9, rotate_prob=1.0)
    assert np.array_equal(out.points, pc.points), "rotation must stay gated by flip_prob"


def _raw_batch(seeds, device):
    raw = []
    for seed in seeds:
        xyz, rgb, sem, ins, npcs, _ = recipe.scene_arrays(seed, recipe.LOADER_POINTS)
        raw.append(PointCloud(pc_id=f"Box_{seed}_00_000", obj_cat=0,
                              points=torch.from_numpy(np.concatenate([xyz, rgb], 1)).to(device),
                              sem_labels=torch.from_numpy(sem.astype(np.int64)).to(device),
                              instance_labels=torch.from_numpy(ins).to(device),
                              gt_npcs=torch.from_numpy(npcs).to(device)))
    return raw


def _check_batch(gold, batch, seeds, tag):
    n = recipe.LOADER_POINTS
    vox_start = 0
    idx = batch.voxel_tensor.indices.cpu().numpy()
    for row, seed in enumerate(seeds):
        pre = f"s{seed}_{tag}_"
        sl = slice(row * n, (row + 1) * n)
        k = int(gold[pre + "num_instances"])
        assert np.array_equal(batch.instance_labels[sl].cpu().numpy(), gold[pre + "instance_labels"])
        assert batch.num_instances[row] == k
        assert np.array_equal(batch.num_points_per_instance[row, :k].cpu().numpy(), gold[pre + "num_points_per_instance"])
        assert np.array_equal(batch.instance_sem_labels[row, :k].cpu().numpy(), gold[pre + "instance_sem_labels"])
        assert bool((batch.instance_sem_labels[row, k:] == -1).all()) and bool((batch.num_points_per_instance[row, k:] == 0).all())
        assert np.allclose(batch.points[sl].cpu().numpy(), gold[pre + "points"], rtol=0, atol=2e-6)
        assert np.allclose(batch.instance_regions[sl].cpu().numpy(), gold[pre + "instance_regions"], rtol=0, atol=3e-6)
        if tag == "plain":  # identical points -> identical voxels (augmented points differ in the last bit, see above)
            mine = idx[idx[:, 0] == row]
            assert np.array_equal(mine[:, 1:], gold[pre + "voxel_coords"])
            assert np.array_equal(batch.pc_voxel_id[sl].cpu().numpy() - vox_start, gold[pre + "pc_voxel_id"])
            feats = batch.voxel_tensor.features[vox_start:vox_start + mine.shape[0]].cpu().numpy()
            assert np.allclose(feats, gold[pre + "voxel_features"], rtol=0, atol=1e-6)
            vox_start += mine.shape[0]
    if tag == "plain":
        want_shape = np.max([gold[f"s{seed}_plain_voxel_coords_range"] for seed in seeds], axis=0).tolist()
        assert list(batch.voxel_tensor.spatial_shape) == want_shape


@pytest.mark.parametrize("tag", ["plain", "aug"])
def test_device_pipeline_on_cpu_tensors_matches_the_reference_loader(gold, tag):
    from oracle import torch_ops
    seeds = recipe.LOADER_SEEDS[:4]
    raw = _raw_batch(seeds, "cpu")
    with backend.using(torch_ops):
        if tag == "aug":
            batch = _prepare_augmented(raw, seeds)
        else:
            batch = dp.prepare_batch(raw, VOXEL, None)
    _check_batch(gold, batch, seeds, tag)


def _prepare_augmented(raw, seeds):
    """the reference seeds are per scene (np.random.seed(seed) before each scene's apply_augmentations), so the batch is
    assembled from per-scene draws in that order"""
    mats, shifts = [], []
    for seed in seeds:
        np.random.seed(seed)
        m, s = dp.draw_augmentation(1, color_channels=3, **recipe.AUG)
        mats.append(m[0]); shifts.append(s[0])
    orig = dp.draw_augmentation
    dp.draw_augmentation = lambda n, **kw: (np.stack(mats), np.stack(shifts))
    try:
        return dp.prepare_batch(raw, VOXEL, dict(recipe.AUG))
    finally:
        dp.draw_augmentation = orig


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "aug"])
def test_device_pipeline_on_the_gpu_matches_the_reference_loader(cuda, gold, tag):
    seeds = recipe.LOADER_SEEDS[:4]
    raw = _raw_batch(seeds, cuda)
    batch = _prepare_augmented(raw, seeds) if tag == "aug" else dp.prepare_batch(raw, VOXEL, None)
    _check_batch(gold, batch, seeds, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", recipe.LOADER_SEEDS[:3])
def test_scene_voxelisation_on_the_gpu_is_bit_exact_vs_the_reference_call_site(cuda, gold, seed):
    """apply_voxelization (dataset/gapartnet.py:179-205) on the reference's own augmented points, HIP kernel V"""
    for tag in ("plain", "aug"):
        pre = f"s{seed}_{tag}_"
        pc = PointCloud(pc_id="x", points=torch.from_numpy(gold[pre + "points"]).to(cuda))
        vox = ds.apply_voxelization(pc, voxel_size=VOXEL)
        _check_voxels(gold, pre, vox.voxel_coords.cpu().numpy(), vox.voxel_features.cpu().numpy(),
                      vox.pc_voxel_id.cpu().numpy(), vox.voxel_coords_range)
