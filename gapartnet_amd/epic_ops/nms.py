"""epic_ops.nms.nms (call site: network/grouping_utils.py:244)."""
import torch

from .. import backend


@torch.no_grad()
def nms(ious: torch.Tensor, scores: torch.Tensor, threshold: float) -> torch.Tensor:
    """greedy NMS over a precomputed IoU matrix; visits by descending score (ties -> lower index); returns kept
    indices (int64) in visiting order."""
    return backend.raw().nms(ious, scores, float(threshold))
