"""Segmentation losses and metrics (reference: gapartnet/network/losses.py:8-158) — value-identical
re-implementation in plain torch (these are elementwise / small reductions; the hot path is elsewhere).
tests/test_golden.py pins focal_loss / dice_loss against values captured from the reference functions."""
from typing import Optional

import torch
import torch.nn.functional as F


@torch.no_grad()
def pixel_accuracy(pred_mask: torch.Tensor, gt_mask: torch.Tensor) -> float:
    """fraction of equal entries (0.0 for an empty mask); returns a Python float like the reference (losses.py:8-19)."""
    if gt_mask.numel() == 0:
        return 0.0
    return ((pred_mask == gt_mask).sum() / gt_mask.numel()).item()


@torch.no_grad()
def mean_iou(pred_mask: torch.Tensor, gt_mask: torch.Tensor, num_classes: int) -> torch.Tensor:
    """mean over classes of (tp + eps) / (pred + gt - tp + eps), eps = 1e-6, on entries with gt >= 0: the value
    kornia.metrics.mean_iou(...).mean() gives for one batch row (losses.py:22-32)."""
    keep = gt_mask >= 0
    pred, gt = pred_mask[keep].long(), gt_mask[keep].long()
    conf = torch.bincount(gt * num_classes + pred, minlength=num_classes * num_classes)
    conf = conf.reshape(num_classes, num_classes).to(torch.float32)
    tp = conf.diagonal()
    denom = conf.sum(0) + conf.sum(1) - tp
    return ((tp + 1e-6) / (denom + 1e-6)).mean()


def focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: Optional[torch.Tensor] = None, gamma: float = 2.0,
               reduction: str = "mean", ignore_index: int = -100) -> torch.Tensor:
    """multi-class focal loss: -(1 - p_t)^gamma * log p_t (optionally class-weighted by alpha); rows whose target is
    ``ignore_index`` are removed first; an all-ignored batch gives 0 (losses.py:35-64)."""
    if ignore_index is not None and reduction in ("mean", "sum"):
        # masked form of "drop the ignored rows, then reduce": no boolean-mask selection, so no host sync
        keep = targets != ignore_index
        safe = torch.where(keep, targets, torch.zeros_like(targets))
        log_p = F.log_softmax(inputs, dim=-1)
        log_pt = log_p.gather(1, safe[:, None]).squeeze(-1)
        ce = -log_pt if alpha is None else -log_pt * alpha[safe]
        loss = torch.where(keep, ce * (1 - log_pt.exp()) ** gamma, torch.zeros_like(ce))
        if reduction == "sum":
            return loss.sum()
        count = keep.sum()
        return torch.where(count > 0, loss.sum() / count.clamp(min=1), torch.zeros_like(loss.sum()))
    if ignore_index is not None:
        keep = targets != ignore_index
        targets = targets[keep]
        if targets.shape[0] == 0:
            return torch.zeros((), dtype=inputs.dtype, device=inputs.device)
        inputs = inputs[keep]
    log_p = F.log_softmax(inputs, dim=-1)
    log_pt = log_p.gather(1, targets[:, None]).squeeze(-1)
    ce = -log_pt if alpha is None else -log_pt * alpha[targets]
    loss = ce * (1 - log_pt.exp()) ** gamma
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def sigmoid_focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: float = 0.25, gamma: float = 2,
                       reduction: str = "none") -> torch.Tensor:
    """binary (RetinaNet) focal loss (losses.py:67-108); unused by the default config, kept for API parity."""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def one_hot(labels: torch.Tensor, num_classes: int, device=None, dtype=None, eps: float = 1e-6) -> torch.Tensor:
    """[B, ...] int64 -> [B, num_classes, ...] one-hot plus eps everywhere (losses.py:111-129)."""
    if labels.dtype != torch.int64:
        raise ValueError(f"labels must be int64, got {labels.dtype}")
    if num_classes < 1:
        raise ValueError(f"num_classes must be >= 1, got {num_classes}")
    shape = labels.shape
    out = torch.zeros((shape[0], num_classes) + tuple(shape[1:]), device=device, dtype=dtype)
    return out.scatter_(1, labels.unsqueeze(1), 1.0) + eps


def dice_loss(input: torch.Tensor, target: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """soft dice over a [B, C, H, W] logit map and [B, H, W] labels: mean_b(1 - 2 * sum(p * y) / (sum(p + y) + eps))
    with y the eps-smoothed one-hot target (losses.py:132-158)."""
    if input.dim() != 4:
        raise ValueError(f"expected BxCxHxW logits, got {tuple(input.shape)}")
    if input.shape[-2:] != target.shape[-2:]:
        raise ValueError(f"logits {tuple(input.shape)} and target {tuple(target.shape)} disagree")
    if input.device != target.device:
        raise ValueError("logits and target must be on the same device")
    prob = F.softmax(input, dim=1)
    y = one_hot(target, num_classes=input.shape[1], device=input.device, dtype=input.dtype)
    inter = (prob * y).sum(dim=(1, 2, 3))
    card = (prob + y).sum(dim=(1, 2, 3))
    return (1.0 - 2.0 * inter / (card + eps)).mean()
