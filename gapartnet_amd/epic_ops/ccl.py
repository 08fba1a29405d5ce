"""epic_ops.ccl.connected_components_labeling (call site: network/grouping_utils.py:135-137)."""
import torch

from .. import backend


@torch.no_grad()
def connected_components_labeling(indices: torch.Tensor, edges: torch.Tensor, compacted: bool = True) -> torch.Tensor:
    """indices [2Q] = interleaved (begin, end) into edges; edges undirected; label = minimum vertex index of the
    component (compacted=False) or its rank among components (compacted=True). SURVEY.md Appendix A.4."""
    return backend.raw().ccl(indices, edges, bool(compacted))
