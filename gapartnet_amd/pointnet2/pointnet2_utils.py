"""PointNet++ point operators with the reference's names and tensor conventions
(reference: dataset/process_tools/utils/pointnet_lib/pointnet2_utils.py:10-332, the Python side of the vendored
``pointnet2_cuda`` extension; ``pointnet2_ops.furthest_point_sample`` at structure/utils.py:360).

Each autograd Function calls one libgpn_hip.so kernel (family F of SURVEY.md §8a) through ``backend.raw()``; layouts
are the reference's: xyz (B, N, 3), features (B, C, N), indices int32.
"""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import backend


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B, N, 3) -> indices (B, npoint) int32; starts at index 0 (sampling_gpu.cu:113-115)."""
        idx = backend.raw().pn2_furthest_point_sampling(xyz.contiguous(), int(npoint))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B, C, N), idx (B, M) -> (B, C, M)."""
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return backend.raw().pn2_gather_points(features.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return backend.raw().pn2_gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class KNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """k nearest known points of every unknown point -> (dist (B, N, k) = sqrt(d2), idx (B, N, k)); k <= 200."""
        d2, idx = backend.raw().pn2_knn(unknown.contiguous(), known.contiguous(), int(k))
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(d2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """three nearest known points -> (dist (B, N, 3) = sqrt(d2), idx (B, N, 3)) (pointnet2_utils.py:112-135)."""
        d2, idx = backend.raw().pn2_three_nn(unknown.contiguous(), known.contiguous())
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(d2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B, C, M), idx/weight (B, N, 3) -> (B, C, N) weighted sum of the three gathered features."""
        ctx.save_for_backward(idx, weight)
        ctx.m = features.shape[2]
        return backend.raw().pn2_three_interpolate(features.contiguous(), idx.contiguous(), weight.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return backend.raw().pn2_three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B, C, N), idx (B, npoint, nsample) -> (B, C, npoint, nsample)."""
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return backend.raw().pn2_group_points(features.contiguous(), idx.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return backend.raw().pn2_group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """first ``nsample`` points (index order) with d2 < radius^2 of each query; row pre-filled with the first hit;
        rows without a hit stay zero (ball_query_gpu.cu:9-45, pointnet2_utils.py:261)."""
        idx = backend.raw().pn2_ball_query(float(radius), int(nsample), xyz.contiguous(), new_xyz.contiguous())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball query + grouping of (relative xyz, features) (pointnet2_utils.py:274-307)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None) -> torch.Tensor:
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "cannot group neither xyz nor features"
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    """one group holding every point (pointnet2_utils.py:310-332)."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None) -> torch.Tensor:
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
