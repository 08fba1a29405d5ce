"""Semantic-segmentation result container (reference: gapartnet/structure/segmentation.py:7-14)."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Segmentation:
    batch_size: int
    sem_preds: torch.Tensor
    sem_labels: Optional[torch.Tensor] = None
    all_accu: Optional[torch.Tensor] = None
    pixel_accu: Optional[float] = None
