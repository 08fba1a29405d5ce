"""epic_ops.iou.batch_instance_seg_iou (call site: network/model.py:373-378)."""
import torch

from .. import backend


@torch.no_grad()
def batch_instance_seg_iou(proposal_offsets: torch.Tensor, instance_labels: torch.Tensor,
                           batch_indices: torch.Tensor, num_points_per_instance: torch.Tensor) -> torch.Tensor:
    """-> ious [P, max_inst] float32: inter / (|proposal| + npi[b, k] - inter)."""
    return backend.raw().instance_iou(proposal_offsets, instance_labels, batch_indices, num_points_per_instance)
