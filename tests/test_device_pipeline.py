"""Per-batch scene preparation (dataset/device_pipeline.py) against the per-scene CPU functions of
dataset/gapartnet.py, which restate the reference loader (dataset/gapartnet.py:85-176 there).  Runs on CPU tensors here;
tests/test_gpu_model.py runs the same comparison with the batch prepared on the GPU by the prefetcher."""
import numpy as np
import pytest
import torch

from gapartnet_amd.dataset import device_pipeline as dp
from gapartnet_amd.dataset import gapartnet as ds
from gapartnet_amd.dataset import synthetic

AUG = dict(pos_jitter=0.1, color_jitter=0.3, flip_prob=0.5, rotate_prob=0.5)


def _scenes(n_points=(1500, 1500, 1500)):
    out = []
    for i, n in enumerate(n_points):
        pc = synthetic.make_scene(4200 + i, n)
        if i == 1:  # sparse, shuffled instance ids: compaction must rank them per scene
            lab = pc.instance_labels.copy()
            lab[lab >= 0] = lab[lab >= 0] * 7 + 3
            pc.instance_labels = lab
        out.append(pc)
    return out


def _per_scene_reference(scenes, aug):
    np.random.seed(77)
    done = []
    for pc in scenes:
        pc = ds.compact_instance_labels(pc)
        if aug:
            pc = ds.apply_augmentations(pc, **aug)
        done.append(ds.generate_inst_info(pc).to_tensor())
    return done


@pytest.mark.parametrize("aug", [None, AUG])
@pytest.mark.parametrize("ragged", [False, True])
def test_batch_preparation_matches_the_per_scene_loader(aug, ragged):
    scenes = _scenes((1500, 900, 1200) if ragged else (1500, 1500, 1500))
    want = _per_scene_reference(scenes, aug)
    raw = [pc.to_tensor() for pc in scenes]
    counts = [pc.points.shape[0] for pc in raw]
    batch_indices = torch.repeat_interleave(torch.arange(len(raw), dtype=torch.int32), torch.tensor(counts))
    points = torch.cat([pc.points for pc in raw])
    ins = torch.cat([pc.instance_labels for pc in raw])
    sem = torch.cat([pc.sem_labels for pc in raw])

    ins_c, k = dp.compact_instance_labels_batch(ins, batch_indices, len(raw))
    assert torch.equal(ins_c, torch.cat([pc.instance_labels for pc in want]))
    assert k.tolist() == [pc.num_instances for pc in want]

    if aug:
        np.random.seed(77)
        mats, shifts = dp.draw_augmentation(len(raw), color_channels=3, **aug)
        points = dp.augment_points(points, batch_indices, mats, shifts)
        assert torch.allclose(points, torch.cat([pc.points for pc in want]), rtol=0, atol=2e-6)  # 3-term fp64 dot, rounded once

    info = dp.inst_info_batch(points, ins_c, sem, batch_indices, k.tolist())
    want_regions = torch.cat([pc.instance_regions for pc in want])
    assert torch.allclose(info["instance_regions"], want_regions, rtol=0, atol=3e-6)
    if not aug:  # min / max are selections: exact when the coordinates are exact
        assert torch.equal(info["instance_regions"][:, 3:], want_regions[:, 3:])
    width = max(pc.num_instances for pc in want)
    for s, pc in enumerate(want):
        assert torch.equal(info["num_points_per_instance"][s, :pc.num_instances], pc.num_points_per_instance)
        assert torch.equal(info["instance_sem_labels"][s, :pc.num_instances], pc.instance_sem_labels)
        assert bool((info["num_points_per_instance"][s, pc.num_instances:] == 0).all())
        assert bool((info["instance_sem_labels"][s, pc.num_instances:] == -1).all())
    assert info["num_points_per_instance"].shape == (len(raw), width)


def test_raw_scenes_collate_like_prepared_scenes():
    """PointCloud.collate on raw scenes (what a device_pipeline dataset yields) == collate of the per-scene-prepared ones"""
    from gapartnet_amd import backend
    from gapartnet_amd.structure.point_cloud import PointCloud
    from oracle import torch_ops
    scenes = _scenes()
    with backend.using(torch_ops):
        a = PointCloud.collate([pc.to_tensor() for pc in scenes], voxel_size=(0.01, 0.01, 0.01))
        b = PointCloud.collate(_per_scene_reference(scenes, None), voxel_size=(0.01, 0.01, 0.01))
    assert a.num_instances == b.num_instances
    for name in ("points", "batch_indices", "sem_labels", "instance_labels", "pc_voxel_id", "num_points_per_instance",
                 "instance_sem_labels", "gt_npcs"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.allclose(a.instance_regions, b.instance_regions, rtol=0, atol=3e-6)
    assert torch.equal(a.voxel_tensor.indices, b.voxel_tensor.indices)
    assert torch.equal(a.voxel_tensor.features, b.voxel_tensor.features)


def test_dataset_hands_over_raw_scenes_when_asked():
    d = ds.SyntheticGAPartNetDataset(2, 1200, augmentation=True, device_pipeline=True, **AUG)
    pc = d[0]
    assert pc.num_instances is None and pc.instance_regions is None and isinstance(pc.points, torch.Tensor)
    want = synthetic.make_scene(1000, 1200)
    assert np.array_equal(pc.points.numpy(), want.points) and np.array_equal(pc.instance_labels.numpy(), want.instance_labels)
