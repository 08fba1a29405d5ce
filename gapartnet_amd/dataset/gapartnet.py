"""Scene loading / augmentation / instance bookkeeping / voxelisation
(reference: gapartnet/dataset/gapartnet.py:22-532) — same function and class names and the same data contract.

MI355X-first difference: the reference voxelises every scene on CPU tensors inside 16 DataLoader workers
(dataset/gapartnet.py:188).  Here scenes stay un-voxelised in the loader and are voxelised on the GPU, batched, by
``PointCloud.collate(..., voxel_size=...)`` (one kernel-V call per batch).  ``apply_voxelization`` is still provided
with the reference's signature for per-scene use on device tensors.
"""
import copy
import os
from glob import glob
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ..epic_ops.voxelize import voxelize
from ..lightning_lite import LightningDataModule
from ..misc.info import OBJECT_NAME2ID
from ..structure.point_cloud import PointCloud
from . import synthetic


def trivial_batch_collator(batch):
    """the reference's collate_fn: hand the list of scenes to the model untouched (dataset/data_utils.py:8-12)."""
    return batch


# ------------------------------------------------------------------------------------------------- per-scene steps
def load_data(file_path: str, no_label: bool = False) -> PointCloud:
    """``.pth`` 6-tuple (xyz, rgb, sem, ins, npcs, pixel idx) -> numpy PointCloud (dataset/gapartnet.py:208-229)."""
    if no_label:
        raise NotImplementedError("unlabelled scenes are not supported by the reference either (dataset/gapartnet.py:211-213)")
    pc_data = torch.load(file_path, weights_only=False)
    pc_id = os.path.basename(file_path).split(".")[0]
    return PointCloud(pc_id=pc_id, obj_cat=OBJECT_NAME2ID.get(pc_id.split("_")[0], -1),
                      points=np.concatenate([pc_data[0], pc_data[1]], axis=-1, dtype=np.float32),
                      sem_labels=np.asarray(pc_data[2]).astype(np.int64),
                      instance_labels=np.asarray(pc_data[3]).astype(np.int32),
                      gt_npcs=np.asarray(pc_data[4]).astype(np.float32))


def downsample(pc: PointCloud, *, max_points: int = 20000) -> PointCloud:
    """the reference only asserts the bound (dataset/gapartnet.py:123-131)."""
    if pc.points.shape[0] > max_points:
        raise AssertionError((pc.points.shape[0], max_points))
    return copy.copy(pc)


def compact_instance_labels(pc: PointCloud) -> PointCloud:
    """relabel the non-negative instance ids to 0..K-1 in ascending order (dataset/gapartnet.py:134-142)."""
    pc = copy.copy(pc)
    labels = pc.instance_labels.copy()
    valid = labels >= 0
    labels[valid] = np.unique(labels[valid], return_inverse=True)[1]
    pc.instance_labels = labels
    return pc


def apply_augmentations(pc: PointCloud, *, pos_jitter: float = 0., color_jitter: float = 0., flip_prob: float = 0.,
                        rotate_prob: float = 0.) -> PointCloud:
    """random linear jitter / x-flip / z-rotation of xyz and a global colour shift, drawing from the global numpy
    generator in the reference's order (dataset/gapartnet.py:85-120).  Quirk kept: the rotation is gated by
    ``flip_prob``, not ``rotate_prob`` (dataset/gapartnet.py:103-104)."""
    pc = copy.copy(pc)
    m = np.eye(3)
    if pos_jitter > 0:
        m = m + np.random.randn(3, 3) * pos_jitter
    if flip_prob > 0 and np.random.rand() < flip_prob:
        m[0, 0] = -m[0, 0]
    if rotate_prob > 0 and np.random.rand() < flip_prob:
        theta = np.random.rand() * np.pi * 2
        c, s = np.cos(theta), np.sin(theta)
        m = m @ np.asarray([[c, s, 0], [-s, c, 0], [0, 0, 1]])
    pts = pc.points.copy()
    pts[:, :3] = pts[:, :3] @ m
    if color_jitter > 0:
        pts[:, 3:] += np.random.randn(1, pts.shape[1] - 3) * color_jitter
    pc.points = pts
    return pc


def generate_inst_info(pc: PointCloud) -> PointCloud:
    """per-point instance region (mean | min | max xyz of the point's instance, zeros off-instance), per-instance
    point counts and semantic labels (dataset/gapartnet.py:145-176) — vectorised instead of a Python loop."""
    pc = copy.copy(pc)
    labels = pc.instance_labels
    n = pc.points.shape[0]
    k = int(labels.max()) + 1
    assert k > 0, "scene without a labelled instance"
    member = labels >= 0
    idx = labels[member].astype(np.int64)
    xyz = pc.points[member, :3].astype(np.float32)
    counts = np.bincount(idx, minlength=k)
    lo = np.full((k, 3), np.inf, np.float32)
    hi = np.full((k, 3), -np.inf, np.float32)
    np.minimum.at(lo, idx, xyz)
    np.maximum.at(hi, idx, xyz)
    mean = np.zeros((k, 3), np.float32)
    for inst in range(k):  # np.mean per instance: same pairwise float32 summation as the reference's xyz_i.mean(0)
        sel = xyz[idx == inst]
        if sel.shape[0]:
            mean[inst] = sel.mean(0)
    regions = np.zeros((n, 9), np.float32)
    regions[member, 0:3], regions[member, 3:6], regions[member, 6:9] = mean[idx], lo[idx], hi[idx]
    first = np.full(k, n, np.int64)
    np.minimum.at(first, idx, np.nonzero(member)[0])
    pc.num_instances = k
    pc.instance_regions = regions
    pc.num_points_per_instance = counts.astype(np.int32)
    pc.instance_sem_labels = pc.sem_labels[np.minimum(first, n - 1)].astype(np.int32)
    return pc


def apply_voxelization(pc: PointCloud, *, voxel_size: Tuple[float, float, float]) -> PointCloud:
    """voxelise one scene (device tensors) with the reference's conventions (dataset/gapartnet.py:179-205):
    range = [min - 1e-4, max + 1e-4], features = mean of the 6-channel points, extent = (max coord + 1).clamp(min=128)."""
    pc = copy.copy(pc)
    n = pc.points.shape[0]
    xyz = pc.points[:, :3]
    dev = xyz.device
    rng_min, rng_max = xyz.min(0)[0] - 1e-4, xyz.max(0)[0] + 1e-4
    feats, coords, _, pc_voxel_id = voxelize(
        xyz, pc.points, batch_offsets=torch.as_tensor([0, n], dtype=torch.int64, device=dev),
        voxel_size=torch.as_tensor(voxel_size, device=dev), points_range_min=rng_min, points_range_max=rng_max,
        reduction="mean")
    assert bool((pc_voxel_id >= 0).all())
    pc.voxel_features, pc.voxel_coords, pc.pc_voxel_id = feats, coords, pc_voxel_id
    pc.voxel_coords_range = (coords.max(0)[0] + 1).clamp(min=128, max=None).tolist()
    return pc


# ------------------------------------------------------------------------------------------------- datasets
class GAPartNetDataset(torch.utils.data.Dataset):
    """``*.pth`` scenes under ``root_dir`` (reference class of the same name, dataset/gapartnet.py:22-82).
    ``voxelize_on_load=False`` (default) leaves voxelisation to the batched device path in ``PointCloud.collate``."""

    def __init__(self, root_dir: Union[str, Sequence[str]] = "", shuffle: bool = False, max_points: int = 20000,
                 augmentation: bool = False,
                 voxel_size: Tuple[float, float, float] = (1 / 100, 1 / 100, 1 / 100), few_shot: bool = False,
                 few_shot_num: int = 512, pos_jitter: float = 0., color_jitter: float = 0., flip_prob: float = 0.,
                 rotate_prob: float = 0., nopart_path: str = "data/nopart.txt", no_label: bool = False,
                 voxelize_on_load: bool = False, device: Optional[torch.device] = None, device_pipeline: bool = False):
        # a list of directories is accepted like in the reference (dataset/gapartnet.py:39-44: train_with_all passes four)
        roots = list(root_dir) if isinstance(root_dir, (list, tuple)) else [root_dir]
        paths = sorted(p for rt in roots for p in glob(os.path.join(str(rt), "*.pth")))
        self.nopart_files = []
        if os.path.exists(nopart_path):
            with open(nopart_path) as fh:
                lines = fh.readlines()
            self.nopart_files = lines[0].split(" ") if lines else []
        skip = {os.path.basename(p).split(".")[0] for p in self.nopart_files}
        self.pc_paths = [p for p in paths if os.path.basename(p).split(".")[0] not in skip]
        self.all_paths = list(self.pc_paths)  # sorted, before the per-run shuffle / few-shot cut (dataset/packed_cache.py keys on it)
        if shuffle:
            np.random.shuffle(self.pc_paths)
        if few_shot:
            self.pc_paths = self.pc_paths[:few_shot_num]
        self.max_points, self.augmentation, self.voxel_size = max_points, augmentation, tuple(voxel_size)
        self.aug = dict(pos_jitter=pos_jitter, color_jitter=color_jitter, flip_prob=flip_prob, rotate_prob=rotate_prob)
        self.no_label, self.voxelize_on_load, self.device = no_label, voxelize_on_load, device
        # device_pipeline: hand over RAW scenes; compaction / augmentation / instance statistics / voxelisation then run
        # per batch on the GPU (dataset/device_pipeline.py) instead of per scene in this worker
        self.device_pipeline = device_pipeline

    def __len__(self):
        return len(self.pc_paths)

    def _prepare(self, pc: PointCloud) -> PointCloud:
        if not bool((pc.instance_labels != -100).any()):
            raise ValueError(f"scene {pc.pc_id} has no labelled instance (the reference stops in ipdb, dataset/gapartnet.py:69-70)")
        if self.device_pipeline:
            return downsample(pc, max_points=self.max_points).to_tensor()
        pc = compact_instance_labels(downsample(pc, max_points=self.max_points))
        if self.augmentation:
            pc = apply_augmentations(pc, **self.aug)
        pc = generate_inst_info(pc).to_tensor()
        if self.voxelize_on_load:
            pc = apply_voxelization(pc.to(self.device or "cuda"), voxel_size=self.voxel_size)
        return pc

    def __getitem__(self, idx):
        return self._prepare(load_data(self.pc_paths[idx], no_label=self.no_label))


class SyntheticGAPartNetDataset(GAPartNetDataset):
    """seeded synthetic scenes (dataset.synthetic) run through the same per-scene pipeline."""

    def __init__(self, num_scenes: int, n_points: int = 20000, seed0: int = 1000, **kw):
        kw.setdefault("max_points", max(n_points, 20000))
        super().__init__(root_dir="/nonexistent", **kw)
        self.num_scenes, self.n_points, self.seed0 = num_scenes, n_points, seed0

    def __len__(self):
        return self.num_scenes

    def __getitem__(self, idx):
        return self._prepare(synthetic.make_scene(self.seed0 + idx, self.n_points))


class GAPartNetInst(LightningDataModule):
    """data module with one training loader and three evaluation loaders (val / test_intra / test_inter), keyword
    arguments as in gapartnet.yaml ``data.init_args`` (dataset/gapartnet.py:288-532).  ``root_dir="synthetic"`` (or
    any ``synthetic:<n>``) serves seeded synthetic scenes instead of files."""

    def __init__(self, root_dir: str, max_points: int = 20000, voxel_size: Tuple[float, float, float] = (1 / 100, 1 / 100, 1 / 100),
                 train_batch_size: int = 32, val_batch_size: int = 32, test_batch_size: int = 32, num_workers: int = 16,
                 pos_jitter: float = 0., color_jitter: float = 0., flip_prob: float = 0., rotate_prob: float = 0.,
                 train_few_shot: bool = False, val_few_shot: bool = False, intra_few_shot: bool = False,
                 inter_few_shot: bool = False, few_shot_num: int = 256, train_with_all: bool = False,
                 device_pipeline: bool = False, packed_cache: bool = False, cache_dir: Optional[str] = None):
        super().__init__()
        self.save_hyperparameters()
        self.root_dir, self.max_points, self.voxel_size = root_dir, max_points, tuple(voxel_size)
        self.train_batch_size, self.val_batch_size, self.test_batch_size = train_batch_size, val_batch_size, test_batch_size
        self.num_workers = num_workers
        self.aug = dict(pos_jitter=pos_jitter, color_jitter=color_jitter, flip_prob=flip_prob, rotate_prob=rotate_prob)
        self.few = dict(train=train_few_shot, val=val_few_shot, intra=intra_few_shot, inter=inter_few_shot)
        self.few_shot_num, self.train_with_all = few_shot_num, train_with_all
        self.device_pipeline = device_pipeline  # not a reference kwarg: per-batch scene preparation on the GPU
        # not reference kwargs: the scenes of a split from ONE memory-mapped array file (written on first use under cache_dir,
        # default <root_dir>/.gpn_cache) and whole batches through pinned staging blocks instead of worker processes unpickling
        # .pth files (dataset/packed_cache.py); implies the raw hand-over of device_pipeline
        self.packed_cache = packed_cache
        self.cache_dir = cache_dir
        if packed_cache:
            self.device_pipeline = True

    def _dataset(self, split: str, sub: str, augmentation: bool, shuffle: bool):
        few = self.few[split]
        if str(self.root_dir).startswith("synthetic"):
            n = int(str(self.root_dir).split(":")[1]) if ":" in str(self.root_dir) else 64
            n = min(n, self.few_shot_num) if few else n
            seed0 = {"train": 1000, "val": 2000, "intra": 3000, "inter": 4000}[split]
            return SyntheticGAPartNetDataset(n, self.max_points, seed0, augmentation=augmentation, voxel_size=self.voxel_size,
                                             device_pipeline=self.device_pipeline, **(self.aug if augmentation else {}))
        if split == "train" and self.train_with_all:
            # the reference trains on train + val + test_intra + test_inter when train_with_all is set
            # (dataset/gapartnet.py:339-355 there)
            root = [os.path.join(self.root_dir, d, "pth") for d in ("train", "val", "test_intra", "test_inter")]
        else:
            root = os.path.join(self.root_dir, sub, "pth")
        return GAPartNetDataset(root, shuffle=shuffle, max_points=self.max_points,
                                augmentation=augmentation, voxel_size=self.voxel_size, few_shot=few,
                                few_shot_num=self.few_shot_num, device_pipeline=self.device_pipeline,
                                **(self.aug if augmentation else {}))

    def setup(self, stage: Optional[str] = None):
        if stage in (None, "fit", "validate"):
            self.train_data_files = self._dataset("train", "train", True, True)
            self.val_data_files = self._dataset("val", "val", False, True)
        self.intra_data_files = self._dataset("intra", "test_intra", False, True)
        self.inter_data_files = self._dataset("inter", "test_inter", False, True)

    def _loader(self, dataset, batch_size, shuffle, drop_last, sampler=None):
        if self.packed_cache and not isinstance(dataset, SyntheticGAPartNetDataset):
            from .packed_cache import PackedSceneLoader, PackedScenes
            split = os.path.basename(os.path.dirname(os.path.dirname(dataset.all_paths[0]))) if dataset.all_paths else "empty"
            cache_dir = self.cache_dir or os.path.join(str(self.root_dir), ".gpn_cache")
            scenes = PackedScenes.open(dataset.all_paths, cache_dir, split, num_workers=self.num_workers, max_points=dataset.max_points)
            where = {p: i for i, p in enumerate(dataset.all_paths)}
            return PackedSceneLoader(scenes, batch_size, shuffle and sampler is None, drop_last, sampler=sampler,
                                     index_map=[where[p] for p in dataset.pc_paths])
        return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and sampler is None,
                                           num_workers=self.num_workers, collate_fn=trivial_batch_collator,
                                           pin_memory=True, drop_last=drop_last, sampler=sampler)

    def train_dataloader(self, sampler=None):
        return self._loader(self.train_data_files, self.train_batch_size, True, True, sampler)

    def val_dataloader(self):
        return [self._loader(self.val_data_files, self.val_batch_size, False, False),
                self._loader(self.intra_data_files, self.val_batch_size, False, False),
                self._loader(self.inter_data_files, self.val_batch_size, False, False)]

    def test_dataloader(self):
        return [self._loader(self.val_data_files, self.test_batch_size, False, False),
                self._loader(self.intra_data_files, self.test_batch_size, False, False),
                self._loader(self.inter_data_files, self.test_batch_size, False, False)]
