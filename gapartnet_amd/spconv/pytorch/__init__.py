"""``spconv.pytorch`` subset used by the reference network (network/backbone.py:8-165,
structure/point_cloud.py:158-162, network/model.py:323-327), implemented over libgpn_hip.so.

Classes: SparseConvTensor, SparseModule, SparseSequential, SubMConv3d, SparseConv3d, SparseInverseConv3d.
Semantics:
  * indices [N,4] int32 = (batch, x, y, z) in ``spatial_shape`` order; features [N,C] float32.
  * rulebooks are cached per ``indice_key`` in ``indice_dict``, which every ``replace_feature`` copy shares.
  * SubMConv3d(k=3,pad=1): cross-correlation, tap (dx+1)*9+(dy+1)*3+(dz+1); output rows == input rows.
  * SparseConv3d(k=2,s=2): out coord = in//2, out shape = D//2 (= (D-2)//2+1), coarse rows in ascending key order.
  * SparseInverseConv3d(k=2, same indice_key): output rows == the paired down-conv's input rows, same taps.
  * weight parameters use the spconv-2.x layout [Cout, kD, kH, kW, Cin]; a 1.x-layout checkpoint tensor
    [kD, kH, kW, Cin, Cout] is accepted by ``load_state_dict`` and permuted.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from ... import backend
from ... import functional as GF


@dataclass
class _DownRecord:
    in_indices: torch.Tensor
    in_shape: List[int]
    out_indices: torch.Tensor
    out_shape: List[int]
    rb_fwd: object  # dst = coarse rows
    rb_bwd: object  # dst = fine rows


class SparseConvTensor:
    def __init__(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape: Sequence[int],
                 batch_size: int, indice_dict: Optional[dict] = None):
        assert features.dim() == 2 and indices.dim() == 2 and indices.shape[1] == 4, (features.shape, indices.shape)
        assert features.shape[0] == indices.shape[0]
        assert indices.dtype == torch.int32, "indices must be int32 (reference: voxel_coords.int(), model.py:324)"
        self._features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        # not part of spconv's API: a tensor whose live row count (and number of batch entries) is still on the device
        # (hip_ops.DevCount; include/gpn.h section DEV) - features / indices then have the rows of a BOUND, batch_size is a
        # bound, and level_plans carries the host's estimates of the coarse levels' row counts.  None: ordinary tensor.
        self.rows_dev = None
        self.batch_dev = None
        self.level_plans = None

    @property
    def features(self) -> torch.Tensor:
        return self._features

    @features.setter
    def features(self, _value):
        raise ValueError("assign features through replace_feature(), as spconv >= 2.1 requires")

    def replace_feature(self, feature: torch.Tensor) -> "SparseConvTensor":
        out = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.indice_dict)
        out.rows_dev, out.batch_dev, out.level_plans = self.rows_dev, self.batch_dev, self.level_plans
        return out

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def dense(self, channels_first: bool = True) -> torch.Tensor:
        D0, D1, D2 = self.spatial_shape
        C = self._features.shape[1]
        out = torch.zeros((self.batch_size, D0, D1, D2, C), dtype=self._features.dtype, device=self._features.device)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self._features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out

    @property
    def device(self):
        return self._features.device


class SparseModule(nn.Module):
    """marker base class: SparseSequential hands these a SparseConvTensor instead of a feature matrix"""


def _is_sparse(module: nn.Module) -> bool:
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """nn.Sequential over sparse tensors; children are registered under their index ("0", "1", ...), which is
    what fixes the reference's state_dict keys."""

    def __init__(self, *args):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def forward(self, x):
        modules = list(self._modules.values())
        i = 0
        while i < len(modules):
            module = modules[i]
            if _is_sparse(module):
                x = module(x)
            elif isinstance(x, SparseConvTensor):
                if x.features.shape[0] > 0:
                    if isinstance(module, nn.BatchNorm1d):
                        # BatchNorm1d [+ ReLU] on a feature matrix: one fused kernel family (csrc/bn.hip)
                        fuse_relu = i + 1 < len(modules) and isinstance(modules[i + 1], nn.ReLU)
                        x = x.replace_feature(GF.bn_act(x.features, module, relu=fuse_relu))
                        i += 1 if fuse_relu else 0
                    else:
                        x = x.replace_feature(module(x.features))
            else:
                x = module(x)
            i += 1
        return x


def _triple(v) -> List[int]:
    return [int(v)] * 3 if isinstance(v, int) else [int(t) for t in v]


class _SparseConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__()
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        assert _triple(dilation) == [1, 1, 1] and groups == 1, "dilation/groups are not used by GAPartNet"
        self.indice_key = indice_key
        k = self.kernel_size
        self.weight = nn.Parameter(torch.empty(self.out_channels, k[0], k[1], k[2], self.in_channels))
        self.bias = nn.Parameter(torch.empty(self.out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        # same scheme as torch.nn.Conv3d (kaiming_uniform_, a=sqrt(5)), evaluated on the [Cout, fan_in] view
        bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "weight"
        if key in state_dict:
            w = state_dict[key]
            k = self.kernel_size
            v1_shape = (k[0], k[1], k[2], self.in_channels, self.out_channels)
            if tuple(w.shape) == v1_shape and tuple(w.shape) != tuple(self.weight.shape):
                state_dict[key] = w.permute(4, 0, 1, 2, 3).contiguous()  # spconv 1.x -> 2.x layout
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def canonical_weight(self) -> torch.Tensor:
        """[K, Cin, Cout], tap-major, differentiable view of the parameter."""
        K = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        return self.weight.permute(1, 2, 3, 4, 0).reshape(K, self.in_channels, self.out_channels)

    def _conv(self, features, rb, rb_t, reverse_taps):
        """the fused sparse conv on this module's parameter"""
        if self.in_channels % 16 == 0 and self.out_channels % 16 == 0:
            K = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            w = self.weight.view(self.out_channels, K, self.in_channels)  # parameter layout, no copy
            return GF.sparse_conv_param(features, w, rb, rb_t, reverse_taps)
        return GF.sparse_conv(features, self.canonical_weight(), rb, rb_t, reverse_taps)

    def _finish(self, out):
        return out if self.bias is None else out + self.bias

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, bias={self.bias is not None}, indice_key={self.indice_key}")


def _identity_rulebook(n: int, device):
    """K = 1 rulebook whose only tap maps every row to itself (pure torch: it is just arange)."""
    from ...hip_ops import Rulebook, n_tiles
    if torch.device(device).type == "cuda":
        from ... import hip_ops
        return hip_ops.rulebook_identity(n, torch.device(device))
    rows = torch.arange(n, dtype=torch.int32, device=device)
    tile_off = torch.clamp(torch.arange(n_tiles(n) + 1, dtype=torch.int32, device=device) * 32, max=n).reshape(1, -1)
    nbr = torch.cat([rows, torch.full((1,), -1, dtype=torch.int32, device=device)])
    return Rulebook(rows, rows, tile_off.contiguous(), 1, n, n, torch.full((), n, dtype=torch.int64, device=device), nbr)


class SubMConv3d(_SparseConvBase):
    """submanifold conv: k=3/pad=1 (rulebook K1) or k=1 (a per-row dense GEMM)."""

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        k = self.kernel_size
        if k == [1, 1, 1]:
            # a 1x1x1 submanifold conv is the K=1 case of the same fused kernel (identity neighbour table): at these
            # sizes ([1e5, 32] x [32, 16]) it is several times faster than a library GEMM launch
            ident_key = f"__identity_{x.features.shape[0]}__"
            rb = x.indice_dict.get(ident_key)
            if rb is None:
                rb = _identity_rulebook(x.features.shape[0], x.features.device)
                x.indice_dict[ident_key] = rb
            out = self._conv(x.features, rb, rb, False)
            return x.replace_feature(self._finish(out))
        assert k == [3, 3, 3] and self.padding == [1, 1, 1] and self.stride == [1, 1, 1], \
            "GAPartNet uses SubMConv3d with kernel 3 / padding 1 or kernel 1 only"
        rb = x.find_indice_pair(self.indice_key)
        if rb is None:
            rb = backend.raw().rulebook_subm3(x.indices, x.spatial_shape)
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = rb
        out = self._conv(x.features, rb, rb, True)
        return x.replace_feature(self._finish(out))


class SparseConv3d(_SparseConvBase):
    """strided sparse conv; GAPartNet uses kernel 2 / stride 2 / padding 0 (rulebook K2)."""

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        assert self.kernel_size == [2, 2, 2] and self.stride == [2, 2, 2] and self.padding == [0, 0, 0], \
            "GAPartNet uses SparseConv3d with kernel 2 / stride 2 only"
        rec = x.find_indice_pair(self.indice_key)
        if rec is None:
            out_idx, out_shape, rb_fwd, rb_bwd = backend.raw().rulebook_down(x.indices, x.spatial_shape, x.batch_size)
            rec = _DownRecord(x.indices, list(x.spatial_shape), out_idx, out_shape, rb_fwd, rb_bwd)
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = rec
        out = self._conv(x.features, rec.rb_fwd, rec.rb_bwd, False)
        return SparseConvTensor(self._finish(out), rec.out_indices, rec.out_shape, x.batch_size, x.indice_dict)


class SparseInverseConv3d(_SparseConvBase):
    """inverse of the SparseConv3d that registered ``indice_key``: restores its input active set and order."""

    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        rec = x.find_indice_pair(self.indice_key)
        assert isinstance(rec, _DownRecord), f"SparseInverseConv3d: no SparseConv3d registered '{self.indice_key}'"
        assert x.features.shape[0] == rec.out_indices.shape[0]
        out = self._conv(x.features, rec.rb_bwd, rec.rb_fwd, False)
        return SparseConvTensor(self._finish(out), rec.in_indices, rec.in_shape, x.batch_size, x.indice_dict)
