"""dataset/packed_cache.py: the ``.pth`` scenes of a split as one memory-mapped array file + whole batches through staging blocks.
Reference behaviour = dataset/gapartnet.py:208-229 (``load_data``: what a scene file contains) and the raw hand-over of
``GAPartNetDataset(device_pipeline=True)``: every scene must come out of the cache bit-identical to what the per-file loader
returns, batches must partition an epoch like the DataLoader's, and a batch prepared from the cache must equal the batch
prepared from the files."""
import os

import numpy as np
import pytest
import torch

from gapartnet_amd.dataset.gapartnet import GAPartNetDataset, GAPartNetInst
from gapartnet_amd.dataset.packed_cache import PackedSceneLoader, PackedScenes, StagedBatch
from gapartnet_amd.structure.point_cloud import PointCloud
from tests.golden.recipe import scene_arrays


@pytest.fixture(scope="module")
def pth_root(tmp_path_factory):
    root = tmp_path_factory.mktemp("gpn_pth")
    rng = np.random.default_rng(0)
    for split, n, seed0 in (("train", 11, 100), ("val", 3, 500), ("test_intra", 3, 600), ("test_inter", 2, 700)):
        d = root / split / "pth"
        os.makedirs(d)
        for i in range(n):
            n_points = 600 + int(rng.integers(0, 200))  # ragged scenes
            torch.save(scene_arrays(seed0 + i, n_points), str(d / f"StorageFurniture_{seed0 + i:05d}_00_{i:03d}.pth"))
    return str(root)


def _same(a: PointCloud, b: PointCloud):
    assert a.pc_id == b.pc_id and a.obj_cat == b.obj_cat
    for f in ("points", "sem_labels", "instance_labels", "gt_npcs"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape, (f, x.dtype, y.dtype)
        assert torch.equal(x, y), f


def test_every_scene_comes_out_of_the_cache_as_the_file_loader_returns_it(pth_root, tmp_path):
    ds = GAPartNetDataset(os.path.join(pth_root, "train", "pth"), device_pipeline=True)
    scenes = PackedScenes.open(ds.all_paths, str(tmp_path / "cache"), "train")
    assert len(scenes) == len(ds) == 11
    for i in range(len(ds)):
        _same(scenes.scene(i), ds[i])
    # a second open finds the files (no rebuild: same directory, same bytes)
    stamp = os.path.getmtime(os.path.join(scenes.directory, "points.f32"))
    again = PackedScenes.open(ds.all_paths, str(tmp_path / "cache"), "train")
    assert again.directory == scenes.directory and os.path.getmtime(os.path.join(again.directory, "points.f32")) == stamp
    # a changed file list is another cache
    other = PackedScenes.open(ds.all_paths[:-1], str(tmp_path / "cache"), "train")
    assert other.directory != scenes.directory and len(other) == 10
    # the dataset's max_points bound holds for the cache too (the per-file loader raises in downsample(); round 5's cache did not look)
    assert PackedScenes.open(ds.all_paths, str(tmp_path / "cache"), "train", max_points=800).directory == scenes.directory
    with pytest.raises(AssertionError):
        PackedScenes.open(ds.all_paths, str(tmp_path / "cache"), "train", max_points=599)
    # a file rewritten in place (same name, same size, a later nanosecond) is another cache
    st = os.stat(ds.all_paths[0])
    os.utime(ds.all_paths[0], ns=(st.st_atime_ns, st.st_mtime_ns + 1000))
    try:
        assert PackedScenes.open(ds.all_paths, str(tmp_path / "cache"), "train").directory != scenes.directory
    finally:
        os.utime(ds.all_paths[0], ns=(st.st_atime_ns, st.st_mtime_ns))


@pytest.mark.parametrize("drop_last", [True, False])
def test_loader_partitions_an_epoch_and_stages_whole_batches(pth_root, tmp_path, drop_last):
    ds = GAPartNetDataset(os.path.join(pth_root, "train", "pth"), device_pipeline=True)
    scenes = PackedScenes.open(ds.all_paths, str(tmp_path / "cache"), "train")
    torch.manual_seed(3)
    loader = PackedSceneLoader(scenes, 4, shuffle=True, drop_last=drop_last, pin=False)
    assert len(loader) == (2 if drop_last else 3)
    for epoch in range(2):
        seen = []
        for batch in loader:
            assert isinstance(batch, StagedBatch)
            pcs = batch.scenes(torch.device("cpu"))
            assert len(pcs) == len(batch) <= 4
            for pc, i in zip(pcs, batch.ids):
                _same(pc, ds[i])
            seen += batch.ids
        assert len(set(seen)) == len(seen) == (8 if drop_last else 11)
    # an explicit sampler (DistributedSampler in the Trainer) decides the order
    loader = PackedSceneLoader(scenes, 2, shuffle=False, drop_last=False, sampler=[5, 1, 7], pin=False)
    assert [b.ids for b in loader] == [[5, 1], [7]]
    # leaving an epoch early stops the staging thread
    it = iter(PackedSceneLoader(scenes, 2, shuffle=False, drop_last=False, pin=False))
    next(it)
    it.close()


def test_data_module_with_the_cache_prepares_the_same_batches_as_from_the_files(pth_root, tmp_path):
    """GAPartNetInst(packed_cache=True) against GAPartNetInst(device_pipeline=True): the same scenes in an evaluation loader's
    (unshuffled) order give the same collated batch - points, labels, per-instance statistics, voxels"""
    kw = dict(max_points=1000, train_batch_size=4, val_batch_size=3, test_batch_size=3, num_workers=0)
    torch.manual_seed(0); np.random.seed(0)
    files = GAPartNetInst(pth_root, device_pipeline=True, **kw)
    files.setup("fit")
    torch.manual_seed(0); np.random.seed(0)  # (the datasets shuffle their file lists with the global generator)
    cached = GAPartNetInst(pth_root, packed_cache=True, cache_dir=str(tmp_path / "cache"), **kw)
    cached.setup("fit")
    assert cached.device_pipeline
    from gapartnet_amd import backend
    from oracle import torch_ops  # (the operators behind collate's voxelisation, on CPU: test infrastructure)
    for a, b in zip(files.val_dataloader(), cached.val_dataloader()):
        batches_a, batches_b = list(a), list(b)
        assert len(batches_a) == len(batches_b) >= 1
        for raw_a, raw_b in zip(batches_a, batches_b):
            with backend.using(torch_ops):
                ba = PointCloud.collate(raw_a, voxel_size=(0.01,) * 3)
                bb = PointCloud.collate(raw_b.to("cpu"), voxel_size=(0.01,) * 3)
            assert ba.pc_ids == bb.pc_ids and ba.num_instances == bb.num_instances
            for f in ("points", "batch_indices", "sem_labels", "instance_labels", "instance_regions", "num_points_per_instance",
                      "instance_sem_labels", "gt_npcs", "pc_voxel_id"):
                assert torch.equal(getattr(ba, f), getattr(bb, f)), f
            assert torch.equal(ba.voxel_tensor.indices, bb.voxel_tensor.indices)
            assert torch.equal(ba.voxel_tensor.features, bb.voxel_tensor.features)
    n = sum(len(b) for b in cached.train_dataloader())
    assert n == 8  # 11 scenes, batches of 4, drop_last
