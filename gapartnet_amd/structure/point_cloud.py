"""Scene containers and the batch collate that builds the SparseConvTensor
(reference: gapartnet/structure/point_cloud.py:9-189) — same class / field names.

Two ways to reach a batch:
  * reference contract: every PointCloud already carries ``voxel_features / voxel_coords / voxel_coords_range /
    pc_voxel_id`` (produced per scene by dataset.apply_voxelization); collate concatenates them.
  * MI355X path: scenes arrive un-voxelised on the device and ``collate(..., voxel_size=...)`` voxelises the WHOLE
    batch with one kernel-V call (per-scene ranges, scene id = key segment).  Because voxels are ordered by
    (scene, x, y, z) the result is identical to voxelising each scene and concatenating.
"""
from dataclasses import dataclass, fields
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .. import backend
from ..spconv import pytorch as spconv


@dataclass
class PointCloudBatch:
    pc_ids: List[str]
    points: torch.Tensor
    batch_indices: torch.Tensor
    batch_size: int
    device: Any = None
    # voxels
    voxel_tensor: Any = None
    pc_voxel_id: Any = None
    pc_voxel_csr: Any = None  # (order, starts): points grouped by voxel, for the deterministic gather backward
    scene_counts: Any = None  # points per scene (not a reference field)
    # semantics
    sem_labels: Optional[torch.Tensor] = None
    obj_cls_labels: Optional[torch.Tensor] = None
    # instances
    instance_labels: Optional[torch.Tensor] = None
    num_instances: Optional[List[int]] = None
    instance_regions: Optional[torch.Tensor] = None
    num_points_per_instance: Optional[torch.Tensor] = None
    instance_sem_labels: Optional[torch.Tensor] = None
    # npcs
    gt_npcs: Optional[torch.Tensor] = None


@dataclass
class PointCloud:
    pc_id: str
    points: Union[torch.Tensor, np.ndarray]
    obj_cat: int = -1
    sem_labels: Optional[Union[torch.Tensor, np.ndarray]] = None
    instance_labels: Optional[Union[torch.Tensor, np.ndarray]] = None
    gt_npcs: Optional[Union[torch.Tensor, np.ndarray]] = None
    num_instances: Optional[int] = None
    # per point, for the instance the point belongs to: [0:3] mean xyz, [3:6] min xyz, [6:9] max xyz
    # (column order as written by dataset.generate_inst_info; only [:, :3] is consumed, model.py:520)
    instance_regions: Optional[Union[torch.Tensor, np.ndarray]] = None
    num_points_per_instance: Optional[Union[torch.Tensor, np.ndarray]] = None
    instance_sem_labels: Optional[Union[torch.Tensor, np.ndarray]] = None
    voxel_features: Optional[torch.Tensor] = None
    voxel_coords: Optional[torch.Tensor] = None
    voxel_coords_range: Optional[List[int]] = None
    pc_voxel_id: Optional[torch.Tensor] = None

    def to_dict(self) -> Dict[str, Any]:
        return {f.name: getattr(self, f.name) for f in fields(self)}

    def to_tensor(self) -> "PointCloud":
        return PointCloud(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v)
                             for k, v in self.to_dict().items()})

    def to(self, device) -> "PointCloud":
        return PointCloud(**{k: (v.to(device) if isinstance(v, torch.Tensor) else v)
                             for k, v in self.to_dict().items()})

    # -------------------------------------------------------------------------------------------------
    @staticmethod
    def collate(point_clouds: Sequence["PointCloud"], voxel_size: Optional[Sequence[float]] = None,
                augmentation: Optional[Dict[str, float]] = None, pyramid_levels: int = 0, voxels: bool = True) -> PointCloudBatch:
        """``pyramid_levels`` (the caller's U-Net depth - 1; the device prefetcher passes it): the row counts of that many
        stride-2 levels come back with the voxelisation's single host read and ride on ``voxel_tensor.level_counts``.
        ``voxels=False``: un-voxelised scenes stay un-voxelised (``voxel_tensor`` None, ``scene_counts`` set): the caller does
        that part itself (dataset/prefetch.py: voxelisation + the backbone's rulebooks as one native call on a worker thread)"""
        n_scenes = len(point_clouds)
        first = point_clouds[0]
        if first.num_instances is None and first.instance_labels is not None and first.voxel_coords is None:
            # raw scenes (GAPartNetDataset(device_pipeline=True)): label compaction, augmentation and the per-instance
            # statistics run here, per batch, on the scenes' device
            from ..dataset.device_pipeline import prepare_batch
            assert voxel_size is not None, "un-voxelised scenes need voxel_size"
            return prepare_batch(point_clouds, voxel_size, augmentation, pyramid_levels=pyramid_levels, voxels=voxels)
        assert not augmentation, "augmentation at collate time needs raw scenes (GAPartNetDataset(device_pipeline=True))"
        device = first.points.device
        counts = [int(pc.points.shape[0]) for pc in point_clouds]

        def cat(attr):
            if getattr(first, attr) is None:
                return None
            return torch.cat([getattr(pc, attr) for pc in point_clouds], dim=0)

        points = cat("points")
        if len(set(counts)) == 1:  # equal-size scenes: no count tensor, so nothing to copy to / read back from the device
            batch_indices = torch.arange(n_scenes, dtype=torch.int32, device=device).repeat_interleave(counts[0])
        else:
            batch_indices = torch.repeat_interleave(
                torch.arange(n_scenes, dtype=torch.int32, device=device),
                torch.as_tensor(counts, dtype=torch.int64, device=device), output_size=sum(counts))

        num_instances = num_points_per_instance = instance_sem_labels = None
        if first.num_instances is not None:
            num_instances = [int(pc.num_instances) for pc in point_clouds]
            width = max(num_instances)
            # [scenes, width] tables from the per-scene vectors: one scatter through host-computed flat positions per table
            # (a row-slice copy per scene was 2 launches per scene)
            num_points_per_instance = torch.zeros((n_scenes, width), dtype=torch.int32, device=device)
            instance_sem_labels = torch.full((n_scenes, width), -1, dtype=torch.int32, device=device)
            if sum(num_instances) > 0:
                flat = np.concatenate([row * width + np.arange(n, dtype=np.int64) for row, n in enumerate(num_instances)])
                flat = torch.from_numpy(flat).to(device, non_blocking=True)
                num_points_per_instance.view(-1)[flat] = torch.cat([pc.num_points_per_instance.to(torch.int32).reshape(-1)
                                                                    for pc in point_clouds])
                instance_sem_labels.view(-1)[flat] = torch.cat([pc.instance_sem_labels.to(torch.int32).reshape(-1)
                                                                for pc in point_clouds])

        csr = None
        if first.voxel_coords is not None:
            # reference contract: concatenate per-scene voxelisations (structure/point_cloud.py:139-170)
            n_vox = [int(pc.voxel_coords.shape[0]) for pc in point_clouds]
            scene_of_voxel = torch.repeat_interleave(
                torch.arange(n_scenes, dtype=torch.int32, device=device),
                torch.as_tensor(n_vox, dtype=torch.int64, device=device))
            coords = torch.cat([pc.voxel_coords for pc in point_clouds], dim=0).to(torch.int32)
            indices = torch.cat([scene_of_voxel[:, None], coords], dim=1).contiguous()
            voxel_features = torch.cat([pc.voxel_features for pc in point_clouds], dim=0)
            spatial_shape = np.max([pc.voxel_coords_range for pc in point_clouds], axis=0).tolist()
            shifted, start = [], 0
            for pc, nv in zip(point_clouds, n_vox):
                ids = pc.pc_voxel_id
                shifted.append(torch.where(ids >= 0, ids + start, ids))  # out of place (the reference mutates the scene)
                start += nv
            pc_voxel_id = torch.cat(shifted, dim=0)
        elif not voxels:
            indices = voxel_features = spatial_shape = pc_voxel_id = None
        else:
            assert voxel_size is not None, "un-voxelised scenes need voxel_size"
            level_counts = None
            vox = voxelize_scenes(points[:, :3], points, counts, voxel_size, pyramid_levels)
            if pyramid_levels:
                indices, voxel_features, spatial_shape, pc_voxel_id, csr, level_counts = vox
            else:
                indices, voxel_features, spatial_shape, pc_voxel_id, csr = vox

        voxel_tensor = None
        if indices is not None:
            voxel_tensor = spconv.SparseConvTensor(voxel_features, indices, spatial_shape, n_scenes)
            if first.voxel_coords is None and level_counts:
                voxel_tensor.level_counts = list(level_counts)  # rows of the backbone's coarse levels: no read when they are built
        return PointCloudBatch(
            pc_ids=[pc.pc_id for pc in point_clouds], points=points, batch_indices=batch_indices,
            batch_size=n_scenes, device=device, voxel_tensor=voxel_tensor, pc_voxel_id=pc_voxel_id,
            pc_voxel_csr=csr, sem_labels=cat("sem_labels"),
            obj_cls_labels=torch.tensor([pc.obj_cat for pc in point_clouds]),
            instance_labels=cat("instance_labels"), num_instances=num_instances,
            instance_regions=cat("instance_regions"), num_points_per_instance=num_points_per_instance,
            instance_sem_labels=instance_sem_labels, gt_npcs=cat("gt_npcs"), scene_counts=counts)


_VOXEL_SIZE_CACHE = {}


def _voxel_size_on(device, voxel_size):
    """the voxel size as a device tensor, uploaded once per (device, size): a pageable host->device copy per batch is a
    synchronising call"""
    key = (str(device), tuple(float(v) for v in voxel_size))
    t = _VOXEL_SIZE_CACHE.get(key)
    if t is None:
        t = _VOXEL_SIZE_CACHE[key] = torch.as_tensor(list(key[1]), dtype=torch.float32, device=device)
    return t


@torch.no_grad()
def voxelize_scenes(xyz: torch.Tensor, feats: torch.Tensor, counts: Sequence[int], voxel_size: Sequence[float],
                    pyramid_levels: int = 0, _packed: bool = True):
    """Batched scene voxelisation with the reference's per-scene conventions (dataset/gapartnet.py:179-205):
    range = [min - 1e-4, max + 1e-4] per scene, spatial extent per scene = (max coord + 1).clamp(min=128), batch
    extent = elementwise max over scenes.  On the HIP backend the whole preparation is one library call and ONE host read
    (gpn_voxelize_scenes: per-scene ranges reduced on the device, keys that need no grid extent, the voxel count, the
    extent and - with ``pyramid_levels`` - the row counts of the backbone's coarse levels in the same read); returns
    (indices, features, spatial_shape, pc_voxel_id, csr[, level_counts])."""
    device = xyz.device
    n_scenes = len(counts)
    ops = backend.raw()
    if xyz.is_cuda and hasattr(ops, "voxelize_scenes") and _packed:
        if len(set(counts)) == 1:
            offsets_dev = torch.arange(n_scenes + 1, dtype=torch.int64, device=device) * int(counts[0])
        else:
            offsets_dev = torch.as_tensor([0] + list(np.cumsum(counts)), dtype=torch.int64).to(device, non_blocking=True)
        got = ops.voxelize_scenes(xyz, feats, offsets_dev, [float(v) for v in voxel_size], pyramid_levels)
        if got is not None:
            vf, indices, pid, order, starts, max_coord, _dropped, level_counts = got
            spatial_shape = [max(int(m) + 1, 128) for m in max_coord] if indices.shape[0] > 0 else [128] * 3
            out = (indices, vf, spatial_shape, pid, (order, starts))
            return out + (level_counts,) if pyramid_levels else out
    offsets = torch.zeros((n_scenes + 1,), dtype=torch.int64)
    offsets[1:] = torch.as_tensor(counts, dtype=torch.int64).cumsum(0)
    if len(set(counts)) == 1:  # equal-size scenes (the 20k-point contract): one strided reduction, offsets made on the device
        offsets_dev = torch.arange(n_scenes + 1, dtype=torch.int64, device=device) * int(counts[0])
        per_scene = xyz.reshape(n_scenes, counts[0], 3)
        lo, hi = per_scene.amin(1), per_scene.amax(1)
    else:
        offsets_dev = offsets.to(device)
        bounds = offsets.tolist()
        lo = torch.stack([xyz[bounds[s]:bounds[s + 1]].amin(0) for s in range(n_scenes)])
        hi = torch.stack([xyz[bounds[s]:bounds[s + 1]].amax(0) for s in range(n_scenes)])
    rmin, rmax = lo - 1e-4, hi + 1e-4
    vs = _voxel_size_on(device, voxel_size)
    cells = (torch.floor((rmax - rmin) / vs).max(0)[0].to(torch.int64) + 2).tolist()  # host sync #1 (3 ints)
    out = backend.raw().voxelize(xyz, feats, offsets_dev, rmin, rmax, [float(v) for v in voxel_size], cells,
                                 want_csr=True, want_stats=True)
    vf, vc, vseg, pid, order, starts, stats = out
    indices = torch.cat([vseg[:, None], vc], dim=1).contiguous()
    # (max coord + 1).clamp(min=128) per scene then max over scenes == max over the batch, clamped; the maxima come
    # back with the voxel count in the kernel wrapper's single host read
    spatial_shape = [max(int(m) + 1, 128) for m in stats["max_coord"]] if vc.shape[0] > 0 else [128] * 3
    out = (indices, vf, spatial_shape, pid, (order, starts))
    return out + (None,) if pyramid_levels else out
