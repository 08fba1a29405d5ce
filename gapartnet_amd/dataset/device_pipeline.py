"""Per-batch scene preparation on the device (the loader side of the hot path).

The reference prepares every scene in a CPU DataLoader worker (dataset/gapartnet.py:55-82): compact the instance ids
(:134-142), augment (:85-120), per-instance statistics with a Python loop over instances (:145-176), voxelise (:179-205).
Here the loader hands over the RAW scene (the ``.pth`` 6-tuple as tensors) and the same steps run once per BATCH on the
GPU, on the prefetch stream, one step ahead of training (dataset/prefetch.py):

* ``draw_augmentation``   - the random draws (a 3x3 matrix and a colour shift per scene) stay on the host, in the
                            reference's per-scene order from numpy's global generator; 12 floats per scene go to the device;
* ``augment_points``      - xyz @ M[scene] in float64 like numpy's float32 @ float64, colour shift;
* ``compact_instance_labels_batch`` - ascending 0..K-1 relabelling per scene from ONE sort of (scene, label) keys;
* ``inst_info_batch``     - mean | min | max region per point, point count and semantic label per instance, for all
                            scenes at once (segmented reductions keyed by (scene, instance));
* voxelisation            - ``PointCloud.collate`` (batched kernel V), unchanged.

Integer outputs equal the per-scene CPU functions exactly; the float32 instance means differ from numpy's pairwise
float32 summation by at most an ulp (summed in float64 here); augmented coordinates are equal up to the rounding of a
3-term float64 dot product.  Every function is plain torch and also runs on CPU tensors (that is how the CPU tests pin
it against the per-scene functions in dataset/gapartnet.py).
"""
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from ..structure.point_cloud import PointCloud, PointCloudBatch


def _upload(host: torch.Tensor, dev: torch.device) -> torch.Tensor:
    """a small host tensor -> device WITHOUT stalling: a pageable host-to-device copy is a synchronising call (it waits for the
    stream's queued work - here the scenes' own upload and the kernels before it: ~0.5 ms of main-thread time per table and
    batch, round 5); through a pinned staging tensor the copy is queued like a kernel (the caching host allocator keeps the
    block until the copy has run)"""
    if dev.type != "cuda":
        return host.to(dev)
    return host.pin_memory().to(dev, non_blocking=True)


_KEY_STRIDE = 1 << 32  # instance ids are int32: (scene << 32 | id) orders by scene, then id


def draw_augmentation(n_scenes: int, *, pos_jitter: float = 0., color_jitter: float = 0., flip_prob: float = 0.,
                      rotate_prob: float = 0., color_channels: int = 3) -> Tuple[np.ndarray, np.ndarray]:
    """-> (M [B,3,3] float64, colour shift [B,C] float64), drawn scene by scene in the order of
    ``apply_augmentations`` (dataset/gapartnet.py:85-120), including its quirk: the rotation is gated by ``flip_prob``."""
    mats = np.empty((n_scenes, 3, 3), np.float64)
    shifts = np.zeros((n_scenes, color_channels), np.float64)
    for s in range(n_scenes):
        m = np.eye(3)
        if pos_jitter > 0:
            m = m + np.random.randn(3, 3) * pos_jitter
        if flip_prob > 0 and np.random.rand() < flip_prob:
            m[0, 0] = -m[0, 0]
        if rotate_prob > 0 and np.random.rand() < flip_prob:
            theta = np.random.rand() * np.pi * 2
            c, sn = np.cos(theta), np.sin(theta)
            m = m @ np.asarray([[c, sn, 0], [-sn, c, 0], [0, 0, 1]])
        mats[s] = m
        if color_jitter > 0:
            shifts[s] = np.random.randn(1, color_channels)[0] * color_jitter
    return mats, shifts


def augment_points(points: torch.Tensor, batch_indices: torch.Tensor, mats: np.ndarray, shifts: np.ndarray) -> torch.Tensor:
    """points [N, 3 + C] float32 -> augmented copy: xyz @ M[scene] (float64 product, rounded once), colours + shift[scene]"""
    dev = points.device
    scene = batch_indices.long()
    m = _upload(torch.from_numpy(np.ascontiguousarray(mats)), dev)[scene]                 # [N,3,3] f64
    xyz = torch.einsum("ni,nij->nj", points[:, :3].double(), m).float()
    shift = _upload(torch.from_numpy(np.ascontiguousarray(shifts)), dev)
    if bool((shifts != 0).any()):
        rgb = (points[:, 3:].double() + shift[scene]).float()  # numpy adds a float64 row to the float32 colours in place
    else:
        rgb = points[:, 3:]
    return torch.cat([xyz, rgb], dim=1)


def compact_instance_labels_batch(instance_labels: torch.Tensor, batch_indices: torch.Tensor, n_scenes: int):
    """-> (labels with the non-negative ids of every scene renumbered 0..K_s-1 in ascending order, K [B] int64 on the
    host).  One ``unique`` over (scene, id) keys; one host read (B counts)."""
    valid = instance_labels >= 0
    scene = batch_indices.long()
    keys = scene[valid] * _KEY_STRIDE + instance_labels[valid].long()
    uniq, inverse = torch.unique(keys, sorted=True, return_inverse=True)
    per_scene = torch.bincount(torch.div(uniq, _KEY_STRIDE, rounding_mode="floor"), minlength=n_scenes)
    first_rank = torch.cumsum(per_scene, 0) - per_scene
    out = instance_labels.clone()
    out[valid] = (inverse - first_rank[scene[valid]]).to(instance_labels.dtype)
    return out, per_scene.cpu()


def inst_info_batch(points: torch.Tensor, instance_labels: torch.Tensor, sem_labels: torch.Tensor,
                    batch_indices: torch.Tensor, num_instances: Sequence[int]) -> Dict[str, torch.Tensor]:
    """``generate_inst_info`` (dataset/gapartnet.py:145-176) for a whole batch.  ``instance_labels`` are compacted,
    ``num_instances[s]`` = K_s.  -> instance_regions [N,9] (mean | min | max xyz of the point's instance, zeros off-instance),
    num_points_per_instance / instance_sem_labels [B, max K] (padding 0 / -1 as ``PointCloud.collate`` pads them)."""
    dev = points.device
    n, n_scenes = points.shape[0], len(num_instances)
    width = int(max(num_instances))
    k_host = torch.as_tensor(list(num_instances), dtype=torch.int64)
    base = _upload(torch.cumsum(k_host, 0) - k_host, dev)      # first global instance slot of each scene
    total = int(k_host.sum())
    member = instance_labels >= 0
    rows = torch.nonzero(member).squeeze(1)
    scene = batch_indices.long()[rows]
    slot = base[scene] + instance_labels[rows].long()           # global instance slot of every member point
    xyz = points[rows, :3]
    counts = torch.bincount(slot, minlength=total)
    idx3 = slot[:, None].expand(-1, 3)
    lo = torch.full((total, 3), float("inf"), dtype=torch.float32, device=dev).scatter_reduce_(0, idx3, xyz, "amin")
    hi = torch.full((total, 3), float("-inf"), dtype=torch.float32, device=dev).scatter_reduce_(0, idx3, xyz, "amax")
    mean = (torch.zeros((total, 3), dtype=torch.float64, device=dev).index_add_(0, slot, xyz.double())
            / counts.clamp(min=1)[:, None]).float()
    regions = torch.zeros((n, 9), dtype=torch.float32, device=dev)
    regions[rows] = torch.cat([mean[slot], lo[slot], hi[slot]], dim=1)
    first = torch.full((total,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, slot, rows, "amin")
    sem_of_slot = sem_labels[first.clamp(max=n - 1)].to(torch.int32)
    # ragged [total] -> padded [B, width]
    slot_scene = torch.repeat_interleave(torch.arange(n_scenes, device=dev), _upload(k_host, dev), output_size=total)
    slot_col = torch.arange(total, device=dev) - base[slot_scene]
    npi = torch.zeros((n_scenes, width), dtype=torch.int32, device=dev)
    isl = torch.full((n_scenes, width), -1, dtype=torch.int32, device=dev)
    npi[slot_scene, slot_col] = counts.to(torch.int32)
    isl[slot_scene, slot_col] = sem_of_slot
    return {"instance_regions": regions, "num_points_per_instance": npi, "instance_sem_labels": isl}


# the fused preparation (csrc/sceneprep.hip) for scenes on the GPU; False: the torch formulation below everywhere (what
# tests/test_gpu_sceneprep.py compares it with)
FUSED = True


def scene_offsets(counts: Sequence[int], dev: torch.device) -> torch.Tensor:
    """[B + 1] int64 row offsets of the scenes on ``dev`` (no host-to-device copy when the scenes have one size)"""
    if len(set(counts)) == 1:
        return torch.arange(len(counts) + 1, dtype=torch.int64, device=dev) * int(counts[0])
    return _upload(torch.as_tensor([0] + list(np.cumsum(counts)), dtype=torch.int64), dev)


def _fused_scene_prepare(points, sem, ins, counts, mats, shifts):
    """-> the outputs of hip_ops.scene_prepare (plus the padded tables under the names inst_info_batch uses), or None when the
    library's tables do not hold a scene's instances (the caller then runs the torch formulation)"""
    from .. import hip_ops
    dev = points.device
    m = s = None
    if mats is not None:
        aug = np.concatenate([np.ascontiguousarray(mats).reshape(len(counts), 9), np.ascontiguousarray(shifts)], axis=1)
        aug_dev = _upload(torch.from_numpy(aug), dev)  # one small table: 9 + C doubles per scene
        m = aug_dev[:, :9].contiguous()
        s = aug_dev[:, 9:].contiguous() if bool((shifts != 0).any()) else None
    return hip_ops.scene_prepare(points, sem, ins, scene_offsets(counts, dev), m, s)


@torch.no_grad()
def prepare_batch(raw: Sequence[PointCloud], voxel_size: Sequence[float], augmentation: Optional[Dict[str, float]] = None,
                  pyramid_levels: int = 0, voxels: bool = True) -> PointCloudBatch:
    """raw scenes (tensors on one device: points, sem_labels, instance_labels, gt_npcs; nothing derived) -> the
    ``PointCloudBatch`` the model trains on: what ``GAPartNetDataset._prepare`` + ``PointCloud.collate`` produce."""
    from ..structure.point_cloud import spconv, voxelize_scenes
    n_scenes = len(raw)
    dev = raw[0].points.device
    counts = [int(pc.points.shape[0]) for pc in raw]
    packed = getattr(raw, "packed", None)  # (dataset.packed_cache: the scenes are slices of these four tensors already)
    if packed is not None:
        points, sem, ins, npcs = packed
    else:
        points = torch.cat([pc.points for pc in raw], dim=0)
        sem = torch.cat([pc.sem_labels for pc in raw], dim=0)
        ins = torch.cat([pc.instance_labels for pc in raw], dim=0)
        npcs = torch.cat([pc.gt_npcs for pc in raw], dim=0) if raw[0].gt_npcs is not None else None
    mats = shifts = None
    if augmentation:
        mats, shifts = draw_augmentation(n_scenes, color_channels=points.shape[1] - 3, **augmentation)
    fused = _fused_scene_prepare(points, sem, ins, counts, mats, shifts) if dev.type == "cuda" and FUSED else None
    if fused is not None:  # (csrc/sceneprep.hip: three launches and one host read instead of ~80 launches and four reads)
        points, batch_indices, ins, info = fused["points"], fused["batch_indices"], fused["instance_labels"], fused
        num_instances = fused["num_instances"]
    else:
        if len(set(counts)) == 1:
            batch_indices = torch.arange(n_scenes, dtype=torch.int32, device=dev).repeat_interleave(counts[0])
        else:
            batch_indices = torch.repeat_interleave(torch.arange(n_scenes, dtype=torch.int32, device=dev),
                                                    _upload(torch.as_tensor(counts, dtype=torch.int64), dev),
                                                    output_size=sum(counts))
        ins, k = compact_instance_labels_batch(ins, batch_indices, n_scenes)
        num_instances = [int(v) for v in k.tolist()]
    empty = [raw[s].pc_id for s in range(n_scenes) if num_instances[s] == 0]
    if empty:
        raise ValueError(f"scenes without a labelled instance: {empty} (the reference stops in ipdb, dataset/gapartnet.py:69-70)")
    if fused is None:
        if augmentation:
            points = augment_points(points, batch_indices, mats, shifts)
        info = inst_info_batch(points, ins, sem, batch_indices, num_instances)
    level_counts = voxel_tensor = pc_voxel_id = csr = None
    if voxels:
        vox = voxelize_scenes(points[:, :3], points, counts, voxel_size, pyramid_levels)
        if pyramid_levels:
            indices, voxel_features, spatial_shape, pc_voxel_id, csr, level_counts = vox
        else:
            indices, voxel_features, spatial_shape, pc_voxel_id, csr = vox
        voxel_tensor = spconv.SparseConvTensor(voxel_features, indices, spatial_shape, n_scenes)
        if level_counts:
            voxel_tensor.level_counts = list(level_counts)
    return PointCloudBatch(
        pc_ids=[pc.pc_id for pc in raw], points=points, batch_indices=batch_indices, batch_size=n_scenes, device=dev,
        voxel_tensor=voxel_tensor, pc_voxel_id=pc_voxel_id, pc_voxel_csr=csr, sem_labels=sem,
        obj_cls_labels=torch.tensor([pc.obj_cat for pc in raw]), instance_labels=ins, num_instances=num_instances,
        instance_regions=info["instance_regions"], num_points_per_instance=info["num_points_per_instance"],
        instance_sem_labels=info["instance_sem_labels"], gt_npcs=npcs, scene_counts=counts)
