"""epic_ops.ball_query.ball_query (call site: network/grouping_utils.py:119-128)."""
from typing import Optional, Tuple

import torch

from .. import backend


@torch.no_grad()
def ball_query(points: torch.Tensor, query: torch.Tensor, batch_indices: torch.Tensor,
               batch_offsets: torch.Tensor, radius: float, num_samples: int,
               point_labels: Optional[torch.Tensor] = None,
               query_labels: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (indices [Q, num_samples] int32, -1 padded; num_points_per_query [Q] int32).

    Neighbours of query i: points j of the same batch segment with equal label and ||p_j - q_i||^2 < radius^2
    (strict), ascending j, truncated to num_samples."""
    return backend.raw().ball_query(points, query, batch_indices, batch_offsets, float(radius), int(num_samples),
                                    point_labels, query_labels)
