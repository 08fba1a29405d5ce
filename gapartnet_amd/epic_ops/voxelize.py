"""epic_ops.voxelize.voxelize (call sites: dataset/gapartnet.py:188-195, network/grouping_utils.py:93-101)."""
from typing import Tuple

import torch

from .. import backend


def _host3(v):
    if isinstance(v, torch.Tensor):
        return [float(x) for x in v.detach().cpu().tolist()]
    return [float(x) for x in v]


def voxelize(points: torch.Tensor, pt_features: torch.Tensor, batch_offsets: torch.Tensor,
             voxel_size: torch.Tensor, points_range_min: torch.Tensor, points_range_max: torch.Tensor,
             reduction: str = "mean") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (voxel_features [V,C], voxel_coords [V,3] int32, voxel_batch_indices [V] int32, pc_voxel_id [M] int32).

    coord = floor((p - range_min) / voxel_size); voxels ordered by ascending (segment,x,y,z); features are the
    mean of member points; points outside [range_min, range_max) get pc_voxel_id = -1.
    The range / voxel-size tensors are read on the host (they are 3 floats each; the reference builds them from
    Python lists), which costs one sync when they live on the device."""
    if reduction != "mean":
        raise NotImplementedError("GAPartNet only uses reduction='mean'")
    vs, mn, mx = _host3(voxel_size), _host3(points_range_min), _host3(points_range_max)
    # cells per axis used only to linearise keys: any bound >= max coord + 1 gives the same voxel order
    grid = [int((mx[a] - mn[a]) / vs[a]) + 2 for a in range(3)]
    dev = points.device
    rmin = torch.tensor([mn], dtype=torch.float32, device=dev)
    rmax = torch.tensor([mx], dtype=torch.float32, device=dev)
    if pt_features.requires_grad and torch.is_grad_enabled():
        from .. import functional as GF  # mean reduction back-propagates to the point features (GF._VoxelMeanFn)
        vf, vc, vseg, pid = GF.voxelize_mean(points.detach(), pt_features, batch_offsets.to(torch.int64), rmin, rmax, vs,
                                             grid)[:4]
        return vf, vc, vseg, pid
    with torch.no_grad():
        vf, vc, vseg, pid = backend.raw().voxelize(points, pt_features, batch_offsets.to(torch.int64), rmin, rmax, vs,
                                                   grid)
    return vf, vc, vseg, pid
