"""Proposal container (reference: gapartnet/structure/instances.py:7-36) — same field names."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Instances:
    # selection of the points that entered clustering, and their order inside proposals
    valid_mask: Optional[torch.Tensor] = None
    valid_indices: Optional[torch.Tensor] = None  # nonzero(valid_mask), computed once (not a reference field)
    sorted_indices: Optional[torch.Tensor] = None
    point_indices: Optional[torch.Tensor] = None  # row of every proposal point in the batch's point matrix (not a reference field)
    pt_xyz: Optional[torch.Tensor] = None
    # CSR over proposals
    batch_indices: Optional[torch.Tensor] = None
    proposal_offsets: Optional[torch.Tensor] = None
    proposal_indices: Optional[torch.Tensor] = None
    num_points_per_proposal: Optional[torch.Tensor] = None
    # predictions
    sem_preds: Optional[torch.Tensor] = None
    pt_sem_classes: Optional[torch.Tensor] = None
    score_preds: Optional[torch.Tensor] = None
    npcs_preds: Optional[torch.Tensor] = None
    # ground truth carried along for losses / AP
    sem_labels: Optional[torch.Tensor] = None
    instance_labels: Optional[torch.Tensor] = None
    instance_sem_labels: Optional[torch.Tensor] = None
    num_points_per_instance: Optional[torch.Tensor] = None
    gt_npcs: Optional[torch.Tensor] = None
    npcs_valid_mask: Optional[torch.Tensor] = None
    ious: Optional[torch.Tensor] = None
    cls_preds: Optional[torch.Tensor] = None
    cls_labels: Optional[torch.Tensor] = None
    name: Optional[str] = None
    # not a reference field: {"M", "P", "V"} -> hip_ops.DevCount when the proposal stage ran without a host read - every tensor
    # above then has the rows of a bound and only the first *count rows are defined (training steps only)
    dev_counts: Optional[dict] = None


@dataclass
class Result:
    xyz: torch.Tensor
    rgb: torch.Tensor
    sem_preds: torch.Tensor
    ins_preds: torch.Tensor
    npcs_preds: torch.Tensor
