"""PointNet++ set-abstraction / feature-propagation modules over the HIP point operators.

Same constructor keywords, forward contracts and state_dict keys as the reference's vendored modules
(dataset/process_tools/utils/pointnet_lib/pointnet2_modules.py:10-156 and the SharedMLP naming of pytorch_utils.py:5-100:
``mlps.<k>.layer<i>.conv.weight``, ``mlps.<k>.layer<i>.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}``),
so a checkpoint of the reference's PointNet++ blocks loads unchanged.  The sampling / grouping / interpolation run on
kernel family F (furthest point sampling, ball query, group, 3-NN, 3-interpolate: gapartnet_amd/csrc/pointnet2.hip) through
the autograd Functions of ``pointnet2_utils``; the shared MLPs are 1x1 convolutions on (B, C, npoint, nsample).

One deliberate difference: the reference adds 3 to the caller's ``mlp`` list IN PLACE when ``use_xyz`` (building two modules
from one spec list silently widens the second); the spec is copied here.
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import pointnet2_utils as P


class PointwiseConv2d(nn.Conv2d):
    """1x1 Conv2d (same parameters / state_dict keys as nn.Conv2d) evaluated as a channel contraction (a library GEMM)
    instead of through the convolution library: MIOpen's implicit-GEMM backward-data kernel for this shape
    (igemm_bwd_gtcx35_nhwc_fp32 ...) faulted on the MI355X image depending on which solver its search picked
    (rocgdb: memory violation inside that kernel, in the middle of the GPU test suite)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.einsum("oi,bihw->bohw", self.weight[:, :, 0, 0], x)
        return y if self.bias is None else y + self.bias[None, :, None, None]


def shared_mlp(spec: Sequence[int], bn: bool = True, instance_norm: bool = False) -> nn.Sequential:
    """per-point MLP as a stack of 1x1 Conv2d (+BatchNorm2d | InstanceNorm2d) + ReLU, children named like the reference's
    ``SharedMLP`` (pytorch_utils.py:5-32): layer<i> -> conv, bn -> bn, activation"""
    net = nn.Sequential()
    for i, (cin, cout) in enumerate(zip(spec[:-1], spec[1:])):
        block = nn.Sequential()
        conv = PointwiseConv2d(cin, cout, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        block.add_module("conv", conv)
        if bn:
            holder = nn.Sequential()
            holder.add_module("bn", nn.BatchNorm2d(cout))
            block.add_module("bn", holder)
        block.add_module("activation", nn.ReLU(inplace=True))
        if not bn and instance_norm:
            block.add_module("in", nn.InstanceNorm2d(cout, affine=False, track_running_stats=False))
        net.add_module(f"layer{i}", block)
    return net


class PointnetSAModuleMSG(nn.Module):
    """set abstraction with multi-scale grouping: FPS centres, one ball-query group + shared MLP + pool per scale"""

    def __init__(self, *, npoint: Optional[int], radii: List[Optional[float]], nsamples: List[Optional[int]],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True, pool_method: str = "max_pool",
                 instance_norm: bool = False):
        super().__init__()
        if not len(radii) == len(nsamples) == len(mlps):
            raise ValueError("radii, nsamples and mlps must have one entry per scale")
        if pool_method not in ("max_pool", "avg_pool"):
            raise NotImplementedError(pool_method)
        self.npoint, self.pool_method = npoint, pool_method
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(P.QueryAndGroup(radius, nsample, use_xyz=use_xyz) if npoint is not None
                                 else P.GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(shared_mlp(spec, bn=bn, instance_norm=instance_norm))

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None, new_xyz: Optional[torch.Tensor] = None
                ) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        """xyz (B, N, 3), features (B, C, N) -> (new_xyz (B, npoint, 3) | None, new_features (B, sum_k mlps[k][-1], npoint))"""
        if new_xyz is None and self.npoint is not None:
            centres = P.furthest_point_sample(xyz, self.npoint)
            new_xyz = P.gather_operation(xyz.transpose(1, 2).contiguous(), centres).transpose(1, 2).contiguous()
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped = mlp(grouper(xyz, new_xyz, features))            # (B, C', npoint, nsample)
            pooled.append(grouped.amax(dim=3) if self.pool_method == "max_pool" else grouped.mean(dim=3))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModule(PointnetSAModuleMSG):
    """single-scale set abstraction (``npoint=None``: one group with every point)"""

    def __init__(self, *, mlp: List[int], npoint: Optional[int] = None, radius: Optional[float] = None,
                 nsample: Optional[int] = None, bn: bool = True, use_xyz: bool = True, pool_method: str = "max_pool",
                 instance_norm: bool = False):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """feature propagation: inverse-distance interpolation from the 3 nearest known points, skip concat, shared MLP"""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = shared_mlp(list(mlp), bn=bn)

    def forward(self, unknown: torch.Tensor, known: Optional[torch.Tensor], unknow_feats: Optional[torch.Tensor],
                known_feats: torch.Tensor) -> torch.Tensor:
        """unknown (B, n, 3), known (B, m, 3), unknow_feats (B, C1, n), known_feats (B, C2, m) -> (B, mlp[-1], n)"""
        if known is not None:
            dist, idx = P.three_nn(unknown, known)
            recip = 1.0 / (dist + 1e-8)
            weight = recip / recip.sum(dim=2, keepdim=True)
            spread = P.three_interpolate(known_feats, idx, weight)
        else:
            spread = known_feats.expand(*known_feats.shape[:2], unknown.shape[1])
        stacked = spread if unknow_feats is None else torch.cat([spread, unknow_feats], dim=1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)
