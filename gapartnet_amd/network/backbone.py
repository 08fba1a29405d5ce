"""Sparse residual U-Net (reference: gapartnet/network/backbone.py:8-165).

Same module tree — and therefore the same state_dict keys — as the reference:
  SparseUNet.stem            = [SubMConv3d(k3, "subm1"), norm, ReLU]      (or [norm, ReLU] when without_stem)
  SparseUNet.ublock          = UBlock(channels, ...)
  UBlock.encoder_blocks      = block_repeat x ResBlock(c0, c0)            key subm{level}
  UBlock.downsample          = [SparseConv3d(k2 s2, "spconv{level}"), norm, ReLU]
  UBlock.ublock              = UBlock(channels[1:], level + 1)
  UBlock.upsample            = [SparseInverseConv3d(k2, "spconv{level}"), norm, ReLU]
  UBlock.decoder_blocks      = ResBlock(2 c0, c0) + (block_repeat - 1) x ResBlock(c0, c0)
  ResBlock.shortcut          = Identity | [SubMConv3d(k1), norm];  conv1 / conv2 = [SubMConv3d(k3), norm]
Every conv is a libgpn_hip.so sparse convolution (gapartnet_amd.spconv.pytorch).
"""
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from .. import backend
from .. import functional as GF
from ..spconv import pytorch as spconv

NormFn = Callable[[int], nn.Module]


def _conv_norm(cin: int, cout: int, kernel: int, norm_fn: NormFn, indice_key: Optional[str]) -> spconv.SparseSequential:
    pad = 1 if kernel == 3 else 0
    return spconv.SparseSequential(
        spconv.SubMConv3d(cin, cout, kernel_size=kernel, padding=pad, bias=False, indice_key=indice_key),
        norm_fn(cout))


class ResBlock(spconv.SparseModule):
    """two 3x3x3 submanifold convs with a residual connection (1x1x1 projection when the width changes)."""

    def __init__(self, in_channels: int, out_channels: int, norm_fn: NormFn, indice_key=None):
        super().__init__()
        if in_channels == out_channels:
            self.shortcut = nn.Identity()
        else:
            self.shortcut = _conv_norm(in_channels, out_channels, 1, norm_fn, None)
        self.conv1 = _conv_norm(in_channels, out_channels, 3, norm_fn, indice_key)
        self.conv2 = _conv_norm(out_channels, out_channels, 3, norm_fn, indice_key)

    def forward(self, x: spconv.SparseConvTensor) -> spconv.SparseConvTensor:
        skip = self.shortcut(x)
        # conv -> BatchNorm -> ReLU, then conv -> BatchNorm -> (+ skip) -> ReLU: BN, residual add and ReLU run as one
        # fused op per layer (gapartnet_amd.functional.bn_act) on the parameters of the BatchNorm1d sub-modules
        y = self.conv1[0](x)
        y = y.replace_feature(GF.bn_act(y.features, self.conv1[1], relu=True))
        y = self.conv2[0](y)
        return y.replace_feature(GF.bn_act(y.features, self.conv2[1], relu=True, residual=skip.features))


class UBlock(nn.Module):
    """one resolution level: encoder blocks, then (if deeper levels exist) down -> child level -> up -> concat skip
    -> decoder blocks."""

    def __init__(self, channels: List[int], block_fn, block_repeat: int, norm_fn: NormFn, indice_key_id: int = 1):
        super().__init__()
        self.channels = channels
        c0 = channels[0]
        subm_key, down_key = f"subm{indice_key_id}", f"spconv{indice_key_id}"
        self.encoder_blocks = spconv.SparseSequential(
            *[block_fn(c0, c0, norm_fn, indice_key=subm_key) for _ in range(block_repeat)])
        if len(channels) > 1:
            c1 = channels[1]
            self.downsample = spconv.SparseSequential(
                spconv.SparseConv3d(c0, c1, kernel_size=2, stride=2, bias=False, indice_key=down_key),
                norm_fn(c1), nn.ReLU())
            self.ublock = UBlock(channels[1:], block_fn, block_repeat, norm_fn, indice_key_id + 1)
            self.upsample = spconv.SparseSequential(
                spconv.SparseInverseConv3d(c1, c0, kernel_size=2, bias=False, indice_key=down_key),
                norm_fn(c0), nn.ReLU())
            widths = [2 * c0] + [c0] * (block_repeat - 1)
            self.decoder_blocks = spconv.SparseSequential(
                *[block_fn(w, c0, norm_fn, indice_key=subm_key) for w in widths])

    def forward(self, x: spconv.SparseConvTensor) -> spconv.SparseConvTensor:
        x = self.encoder_blocks(x)
        if len(self.channels) == 1:
            return x
        skip = x
        x = self.upsample(self.ublock(self.downsample(x)))
        x = x.replace_feature(torch.cat([x.features, skip.features], dim=-1))
        return self.decoder_blocks(x)


class SparseUNet(nn.Module):
    # run forward / backward of the whole network through the native layer-program executor (network/net_exec.py,
    # csrc/net.hip) instead of module by module; same kernels, same results, one library call per direction
    use_native_executor = True

    def __init__(self, stem: Optional[nn.Module], ublock: UBlock):
        super().__init__()
        self.stem = stem
        self.ublock = ublock

    def forward(self, x: spconv.SparseConvTensor) -> spconv.SparseConvTensor:
        if self.use_native_executor and backend.raw().name == "hip":
            from . import net_exec
            out = net_exec.run(self, x)
            if out is not None:
                return out
        if getattr(x, "rows_dev", None) is not None:
            # a tensor whose buffers are sized for a BOUND (live row count on the device): only the native executor reads that
            # counter; the module-by-module path below would run BatchNorm statistics, convs and weight gradients over the
            # unwritten rows behind it without any error
            raise RuntimeError("SparseUNet.forward: the input's row count is a device counter (sync-free proposal stage); that "
                               "form runs on the native layer-program executor only (use_native_executor, HIP backend, uniform "
                               "BatchNorm training flags) - set model.sync_free_proposals = False")
        if self.stem is not None:
            x = self.stem(x)
        return self.ublock(x)

    @classmethod
    def build(cls, in_channels: int, channels: List[int], block_repeat: int, norm_fn: NormFn,
              without_stem: bool = False) -> "SparseUNet":
        if without_stem:
            stem = spconv.SparseSequential(norm_fn(channels[0]), nn.ReLU())
        else:
            stem = spconv.SparseSequential(
                spconv.SubMConv3d(in_channels, channels[0], kernel_size=3, padding=1, bias=False, indice_key="subm1"),
                norm_fn(channels[0]), nn.ReLU())
        return cls(stem, UBlock(channels, ResBlock, block_repeat, norm_fn, indice_key_id=1))
