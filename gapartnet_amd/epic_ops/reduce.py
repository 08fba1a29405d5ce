"""epic_ops.reduce.{segmented_reduce, segmented_maxpool} (call sites: network/grouping_utils.py:59-70,
network/model.py:360-362)."""
from typing import Tuple

import torch

from .. import backend
from .. import functional as GF


@torch.no_grad()
def segmented_reduce(values: torch.Tensor, segment_offsets_begin: torch.Tensor, segment_offsets_end: torch.Tensor,
                     mode: str = "sum") -> torch.Tensor:
    """values [M,C]; per-segment sum / min / max -> [P,C]."""
    return backend.raw().segmented_reduce(values, segment_offsets_begin, segment_offsets_end, mode)


def segmented_maxpool(values: torch.Tensor, segment_offsets_begin: torch.Tensor,
                      segment_offsets_end: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """differentiable per-segment max-pool -> (pooled [P,C], argmax row [P,C] int32); ties -> lowest row."""
    return GF.segmented_maxpool(values, segment_offsets_begin, segment_offsets_end)
