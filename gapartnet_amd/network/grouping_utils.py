"""Proposal clustering / re-voxelisation / filtering / NMS / AP glue
(reference: gapartnet/network/grouping_utils.py:14-454) — same function names and results, written over the
HIP operators (epic_ops mirrors) and restructured to stay on the device.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..epic_ops.ball_query import ball_query
from ..epic_ops.ccl import connected_components_labeling
from ..epic_ops.nms import nms
from ..epic_ops.reduce import segmented_reduce
from .. import functional as GF
from ..structure.instances import Instances


def offsets_from_counts(counts: torch.Tensor, dtype=torch.int32) -> torch.Tensor:
    """CSR offsets [n+1] from per-segment counts."""
    out = torch.zeros((counts.shape[0] + 1,), dtype=dtype, device=counts.device)
    out[1:] = counts.cumsum(0)
    return out


# ------------------------------------------------------------------------------------------------- NPCS loss
def compute_npcs_loss(npcs_preds: torch.Tensor, gt_npcs: torch.Tensor, proposal_indices: torch.Tensor,
                      symmetry_matrix: torch.Tensor) -> torch.Tensor:
    """symmetry-aware smooth-L1-like NPCS loss (grouping_utils.py:14-43): per point and symmetry m the target is
    gt @ S_m, residual r = pred - target - 0.5, cost = 5 r^2 if r^2 <= 0.01 else |r| - 0.05; mean per proposal,
    min over symmetries, mean over proposals.  ``proposal_indices`` must be grouped (non-decreasing runs)."""
    _, lengths = torch.unique_consecutive(proposal_indices, return_counts=True)
    targets = torch.matmul(gt_npcs[:, None, None, :], symmetry_matrix).squeeze(2)          # [n, m, 3]
    dist2 = ((npcs_preds[:, None, :] - targets - 0.5) ** 2).sum(dim=-1)                  # [n, m]
    cost = torch.where(dist2 <= 0.01, 5 * dist2, torch.sqrt(dist2) - 0.05)
    per_proposal = torch.segment_reduce(cost, "mean", lengths=lengths)                    # [P, m]
    return per_proposal.min(dim=-1)[0].mean()


def compute_npcs_loss_masked(npcs_preds: torch.Tensor, gt_npcs: torch.Tensor, proposal_indices: torch.Tensor,
                             symmetry_matrix: torch.Tensor, member: torch.Tensor, num_proposals: int) -> torch.Tensor:
    """``compute_npcs_loss`` restricted to the points where ``member`` is set, WITHOUT selecting them: same value as
    compute_npcs_loss(npcs_preds[member], gt_npcs[member], proposal_indices[member], symmetry_matrix[member]) (0 when
    no point is a member), but no data-dependent shapes and therefore no host sync.  ``proposal_indices`` must be
    non-decreasing with values in [0, num_proposals)."""
    targets = torch.matmul(gt_npcs[:, None, None, :], symmetry_matrix).squeeze(2)          # [n, m, 3]
    dist2 = ((npcs_preds[:, None, :] - targets - 0.5) ** 2).sum(dim=-1)                  # [n, m]
    cost = torch.where(dist2 <= 0.01, 5 * dist2, torch.sqrt(dist2) - 0.05)
    cost = torch.where(member[:, None], cost, torch.zeros_like(cost))
    # per-proposal sums over the (contiguous) runs of proposal_indices: ordered segment sums (deterministic,
    # differentiable); `unsafe=True` skips the lengths-vs-size validation, which would be a host sync
    edges = torch.searchsorted(proposal_indices.contiguous(),
                               torch.arange(num_proposals + 1, dtype=proposal_indices.dtype, device=cost.device))
    lengths = edges[1:] - edges[:-1]
    seg_sum = torch.segment_reduce(cost, "sum", lengths=lengths, unsafe=True)              # [P, m]
    seg_cnt = torch.segment_reduce(member.to(cost.dtype), "sum", lengths=lengths, unsafe=True)  # [P]
    has = seg_cnt > 0
    per_proposal = seg_sum / seg_cnt.clamp(min=1)[:, None]
    best = per_proposal.min(dim=-1)[0]
    total = torch.where(has, best, torch.zeros_like(best)).sum()
    return total / has.sum().clamp(min=1)


class SymmetryTables:
    """the three symmetry tables of the NPCS loss ([3,2,3,3] for symmetry types 0-2, [1,12,3,3] for type 3, [1,24,3,3]
    for type 4; misc/info.py:338-346) flattened for ``compute_npcs_loss_grouped``:
      flat  [3, 3 * n_matrices]   every candidate rotation side by side (one matmul gives every candidate target)
      cols  [n_types, m_max]      per symmetry type, the columns of its candidates (short lists padded by repeating
                                  their last candidate: a repeated candidate cannot change a minimum)
      group [n_types]             which of the three tables (= loss terms) a type belongs to"""

    def __init__(self, tables, device):
        mats, cols, group = [], [], []
        m_max = max(t.shape[1] for t in tables)
        for g, table in enumerate(tables):
            for t in range(table.shape[0]):
                start = len(mats)
                mats.extend(table[t, m] for m in range(table.shape[1]))
                ids = list(range(start, start + table.shape[1]))
                cols.append(ids + [ids[-1]] * (m_max - len(ids)))
                group.append(g)
        stacked = torch.stack(mats).to(device=device, dtype=torch.float32)            # [n_matrices, 3, 3]
        self.flat = stacked.permute(1, 0, 2).reshape(3, -1).contiguous()             # [3, n_matrices * 3]
        self.cols = torch.tensor(cols, dtype=torch.int64, device=device)
        self.group = torch.tensor(group, dtype=torch.int64, device=device)
        self.n_groups = len(tables)


def compute_npcs_loss_grouped(npcs_preds: torch.Tensor, gt_npcs: torch.Tensor, proposal_indices: torch.Tensor,
                              sym: torch.Tensor, tables: SymmetryTables, num_proposals: int) -> torch.Tensor:
    """sum over the symmetry groups of ``compute_npcs_loss`` on that group's points (model.py:446-462 calls it once per
    group on boolean-mask selections), evaluated for all groups at once and without selecting anything:
      * every candidate target of every point comes from ONE [n,3] x [3, 3*38] matmul instead of a per-point gather of
        3x3 matrices (which moved 136 MB per call);
      * per (proposal, group) means are ordered segment sums over the runs of ``proposal_indices`` (non-decreasing, values
        in [0, num_proposals)) of the member-masked costs; a group without points contributes 0, as the reference's
        ``if`` does.
    ``sym`` = symmetry type of every point.  No data-dependent shapes, so no host sync."""
    n = npcs_preds.shape[0]
    targets = (gt_npcs @ tables.flat).view(n, tables.flat.shape[1] // 3, 3)             # [n, n_matrices, 3]
    dist2 = ((npcs_preds[:, None, :] - targets - 0.5) ** 2).sum(dim=-1)                  # [n, n_matrices]
    cost = torch.where(dist2 <= 0.01, 5 * dist2, torch.sqrt(dist2) - 0.05)
    cost = cost.gather(1, tables.cols[sym])                                              # [n, m_max] own candidates
    member = tables.group[sym][:, None] == torch.arange(tables.n_groups, device=sym.device)[None, :]   # [n, G]
    m_max = tables.cols.shape[1]
    masked = (cost[:, None, :] * member[:, :, None].to(cost.dtype)).reshape(n, tables.n_groups * m_max)
    edges = torch.searchsorted(proposal_indices.contiguous(),
                               torch.arange(num_proposals + 1, dtype=proposal_indices.dtype, device=cost.device))
    lengths = edges[1:] - edges[:-1]
    seg_sum = torch.segment_reduce(masked, "sum", lengths=lengths, unsafe=True).view(num_proposals, tables.n_groups, m_max)
    seg_cnt = torch.segment_reduce(member.to(cost.dtype), "sum", lengths=lengths, unsafe=True)   # [P, G]
    has = seg_cnt > 0
    best = (seg_sum / seg_cnt.clamp(min=1)[:, :, None]).min(dim=-1)[0]                   # [P, G]
    per_group = torch.where(has, best, torch.zeros_like(best)).sum(0) / has.sum(0).clamp(min=1)
    return per_group.sum()


# ------------------------------------------------------------------------------------------------- re-voxelise
def segmented_voxelize(pt_xyz: torch.Tensor, pt_features: torch.Tensor, segment_offsets: torch.Tensor,
                       segment_indices: torch.Tensor, num_points_per_segment: torch.Tensor, score_fullscale: float,
                       score_scale: float, jitter: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                       with_extras: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Scale every proposal into a ``score_fullscale``^3 grid and voxelise it as its own batch element
    (grouping_utils.py:47-104).  -> (voxel_features, voxel_coords [V,4] = (proposal, x, y, z), pc_voxel_id).

    ``jitter`` = the two uniform 3-vectors the reference draws with torch.rand(3) (grouping_utils.py:86-90, one pair
    shared by all proposals, drawn in eval too); pass them to make a run reproducible, else they are drawn here in
    the same order from the device generator.  ``with_extras`` appends {"csr": points grouped by voxel (for the
    gather's backward), "dropped": number of points that fell outside the grid}, both from the voxeliser's own pass."""
    begin, end = segment_offsets[:-1], segment_offsets[1:]
    mean = segmented_reduce(pt_xyz, begin, end, mode="sum") / num_points_per_segment[:, None]
    centered = pt_xyz - mean[segment_indices]
    lo = segmented_reduce(centered, begin, end, mode="min")
    hi = segmented_reduce(centered, begin, end, mode="max")

    scale = 1.0 / ((hi - lo) / score_fullscale).max(-1)[0] - 0.01
    scale = torch.clamp(scale, min=None, max=score_scale)
    lo_s, hi_s = lo * scale[:, None], hi * scale[:, None]
    extent = hi_s - lo_s
    if jitter is None:
        r_a = torch.rand(3, dtype=lo.dtype, device=lo.device)
        r_b = torch.rand(3, dtype=lo.dtype, device=lo.device)
    else:
        r_a, r_b = jitter
    shift = (-lo_s + torch.clamp(score_fullscale - extent - 0.001, min=0) * r_a
             + torch.clamp(score_fullscale - extent + 0.001, max=0) * r_b)
    scaled = centered * scale[segment_indices][:, None] + shift[segment_indices]

    full = float(score_fullscale)
    n_seg = segment_offsets.shape[0] - 1
    dev = pt_xyz.device
    rmin = torch.zeros((1, 3), dtype=torch.float32, device=dev)
    rmax = torch.full((1, 3), full, dtype=torch.float32, device=dev)
    # direct kernel-V call with host-known grid (no sync for the range tensors, unlike the generic wrapper)
    # kernel V through its differentiable wrapper: the proposal features carry the backbone's graph
    vf, vc, vseg, pid, order, starts, stats = GF.voxelize_mean(
        scaled, pt_features, segment_offsets.to(torch.int64), rmin, rmax, [1.0, 1.0, 1.0], [int(full) + 1] * 3,
        want_stats=with_extras)
    voxel_coords = torch.cat([vseg[:, None], vc], dim=1)
    if with_extras:
        return vf, voxel_coords, pid, {"csr": (order, starts), "dropped": stats["dropped"]}
    return vf, voxel_coords, pid


# ------------------------------------------------------------------------------------------------- clustering
_SIZE_LADDER_WARMED = set()


def warm_size_dependent_kernels(device: torch.device) -> None:
    """torch's sort / scan / select pick a different rocPRIM kernel family per input-size class, and HIP loads a kernel's code
    the first time it is launched: the first step whose foreground-point count falls into a new class stalled 40-50 ms in
    ``aten::sort`` (tools/hiccup_probe.py: step 23 of a cold run, when the half-trained network's foreground set shrinks).
    Touch every size class once, up front, per device."""
    if device.type != "cuda" or device.index in _SIZE_LADDER_WARMED:
        return
    _SIZE_LADDER_WARMED.add(device.index)
    for p in range(6, 22):
        n = (1 << p) - 3
        for dtype in (torch.int32, torch.int64):
            keys = torch.arange(n, device=device, dtype=dtype).flip(0) // 3
            torch.sort(keys, stable=True)
            torch.cumsum(keys, 0)
            torch.unique_consecutive(keys, return_inverse=True, return_counts=True)
        torch.nonzero(keys > (n // 6))


def cluster_proposals(pt_xyz: torch.Tensor, batch_indices: torch.Tensor, batch_offsets: torch.Tensor,
                      sem_preds: torch.Tensor, ball_query_radius: float, max_num_points_per_query: int
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """label-aware ball query -> connected components -> points sorted by component
    (grouping_utils.py:108-140).  Components are labelled by their minimum point index and the sort is stable, so
    proposals come out ordered by first member and members in ascending point order."""
    K = int(max_num_points_per_query)
    warm_size_dependent_kernels(pt_xyz.device)
    neighbours, counts = ball_query(pt_xyz, pt_xyz, batch_indices, batch_offsets, ball_query_radius, K,
                                    point_labels=sem_preds, query_labels=sem_preds)
    begin = torch.arange(pt_xyz.shape[0], dtype=torch.int32, device=pt_xyz.device) * K
    begin_end = torch.stack([begin, begin + counts.to(torch.int32)], dim=1).view(-1)
    cc_labels = connected_components_labeling(begin_end, neighbours.view(-1), compacted=False)
    sorted_cc_labels, sorted_indices = torch.sort(cc_labels, stable=True)
    return sorted_cc_labels, sorted_indices


def get_gt_scores(ious: torch.Tensor, fg_thresh: float = 0.75, bg_thresh: float = 0.25) -> torch.Tensor:
    """soft score target: 0 below bg_thresh, 1 above fg_thresh, linear in between (grouping_utils.py:144-156)."""
    k = 1 / (fg_thresh - bg_thresh)
    b = bg_thresh / (bg_thresh - fg_thresh)
    fg = ious > fg_thresh
    mid = ~(fg | (ious < bg_thresh))
    return torch.where(mid, ious * k + b, fg.to(ious.dtype))


# ------------------------------------------------------------------------------------------------- filtering
def _keep_proposals(proposals: Instances, keep_proposal: torch.Tensor) -> Instances:
    """restrict every per-point / per-proposal field to the proposals selected by the boolean mask."""
    keep_point = keep_proposal[proposals.proposal_indices]
    _, new_indices, counts = torch.unique_consecutive(proposals.proposal_indices[keep_point], return_inverse=True,
                                                      return_counts=True)
    npcs_keep = keep_point[proposals.npcs_valid_mask] if proposals.npcs_valid_mask is not None else keep_point

    def pts(t, mask=keep_point):
        return None if t is None else t[mask]

    return Instances(
        valid_mask=proposals.valid_mask, valid_indices=proposals.valid_indices,
        sorted_indices=pts(proposals.sorted_indices), point_indices=pts(proposals.point_indices),
        pt_xyz=pts(proposals.pt_xyz),
        batch_indices=pts(proposals.batch_indices), proposal_offsets=offsets_from_counts(counts),
        proposal_indices=new_indices, num_points_per_proposal=counts, sem_preds=pts(proposals.sem_preds),
        score_preds=proposals.score_preds[keep_proposal], npcs_preds=pts(proposals.npcs_preds, npcs_keep),
        sem_labels=pts(proposals.sem_labels), instance_labels=pts(proposals.instance_labels),
        instance_sem_labels=proposals.instance_sem_labels, num_points_per_instance=proposals.num_points_per_instance,
        gt_npcs=pts(proposals.gt_npcs, npcs_keep), npcs_valid_mask=pts(proposals.npcs_valid_mask),
        ious=None if proposals.ious is None else proposals.ious[keep_proposal])


def filter_invalid_proposals(proposals: Instances, score_threshold: float, min_num_points_per_proposal: int) -> Instances:
    """drop proposals with score <= threshold or size <= min points (strict, grouping_utils.py:159-218)."""
    keep = (proposals.score_preds > score_threshold) & (proposals.num_points_per_proposal > min_num_points_per_proposal)
    return _keep_proposals(proposals, keep)


@torch.no_grad()
def proposal_intersections(sorted_indices: torch.Tensor, proposal_indices: torch.Tensor, num_proposals: int,
                           max_sets: Optional[int] = None) -> torch.Tensor:
    """[P,P] float32 number of shared points between proposals (diagonal = sizes).  Equivalent to the reference's
    dense ``csr @ csr.T`` (grouping_utils.py:231-239) but built from the sorted (point, proposal) incidence list, so
    memory is O(P^2 + M) instead of O(P * M).  ``max_sets``: an upper bound on the proposals a point can be in, when the
    caller knows one (dual-set clustering: 2) - the walk over neighbour distances then has a fixed length and no host read;
    None: walk until no point repeats at that distance (one read per distance)."""
    dev = sorted_indices.device
    inter = torch.zeros((num_proposals, num_proposals), dtype=torch.float32, device=dev)
    sizes = torch.bincount(proposal_indices, minlength=num_proposals).to(torch.float32)
    inter.diagonal().copy_(sizes)
    if sorted_indices.shape[0] < 2:
        return inter
    pt, order = torch.sort(sorted_indices.to(torch.int64), stable=True)
    prop = proposal_indices[order]
    if max_sets is not None:
        for d in range(1, min(int(max_sets), pt.shape[0])):
            w = (pt[d:] == pt[:-d]).to(torch.float32)  # weight 0 where the two entries are different points: nothing selected
            a, b = prop[:-d], prop[d:]
            inter.index_put_((a, b), w, accumulate=True)
            inter.index_put_((b, a), w, accumulate=True)
        return inter
    d = 1
    while d < pt.shape[0]:
        same = pt[d:] == pt[:-d]
        if not bool(same.any()):
            break
        a, b = prop[:-d][same], prop[d:][same]
        ones = torch.ones(a.shape[0], dtype=torch.float32, device=dev)
        inter.index_put_((a, b), ones, accumulate=True)
        inter.index_put_((b, a), ones, accumulate=True)
        d += 1
    return inter


def apply_nms(proposals: Instances, iou_threshold: float = 0.3, max_sets: Optional[int] = None) -> Instances:
    """greedy NMS on point-set IoU between proposals (grouping_utils.py:221-298).  ``max_sets``: see proposal_intersections."""
    P = proposals.score_preds.shape[0]
    inter = proposal_intersections(proposals.sorted_indices, proposals.proposal_indices, P, max_sets)
    sizes = proposals.num_points_per_proposal.to(torch.float32)
    union = sizes[:, None] + sizes[None, :] - inter
    ious = inter / (union + 1e-8)
    keep = nms(ious, proposals.score_preds, iou_threshold)
    mask = torch.zeros(P, dtype=torch.bool, device=proposals.score_preds.device)
    mask[keep] = True
    return _keep_proposals(proposals, mask)


# ------------------------------------------------------------------------------------------------- AP
def voc_ap(rec: torch.Tensor, prec: torch.Tensor, use_07_metric: bool = False) -> float:
    """VOC average precision: 11-point (2007) or area under the monotone precision envelope (grouping_utils.py:302-342)."""
    rec = rec.detach().cpu().numpy()
    prec = prec.detach().cpu().numpy().astype(rec.dtype)
    dt = rec.dtype.type
    if use_07_metric:
        ap = dt(0)
        for t in np.arange(0, 11) / 10.0:
            sel = rec >= t
            ap = ap + (prec[sel].max() if sel.any() else dt(0)) / dt(11.0)
        return float(ap)
    mrec = np.concatenate([[0.0], rec, [1.0]]).astype(rec.dtype)
    mpre = np.concatenate([[0.0], prec, [0.0]]).astype(rec.dtype)
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]  # precision envelope
    i = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(((mrec[i + 1] - mrec[i]) * mpre[i + 1]).sum(dtype=rec.dtype))


def _compute_ap_per_class(tp: torch.Tensor, fp: torch.Tensor, num_gt_instances) -> float:
    if tp.shape[0] == 0:
        return 0.0
    tp, fp = tp.cumsum(0), fp.cumsum(0)
    return voc_ap(tp / num_gt_instances, tp / (tp + fp + 1e-8))


def compute_ap_multi(proposals: List[Instances], num_classes: int, iou_thresholds: Sequence[float]) -> torch.Tensor:
    """per-class AP at several IoU thresholds over a list of per-batch proposal sets -> [T, num_classes - 1] float32 tensor
    on the proposals' device (grouping_utils.py:360-454 is the per-threshold form; model.py:734-745 calls it ten times).

    The reference walks the proposals in descending confidence, one Python iteration (and several device round trips) each:
    a proposal is a true positive if its best-IoU ground-truth instance of the same class exceeds the threshold and is
    still unmatched.  A proposal's best instance does not depend on the matching state, and an instance is matched
    exactly when the FIRST proposal (in confidence order) that clears the threshold on it is reached - so the walk is
    equivalent to: candidates = proposals whose best same-class IoU > threshold; true positives = the first candidate of
    every (set, scene, instance) key in confidence order.  Everything here is array operations ON THE DEVICE that holds the
    proposals: one sort, one min-scatter per threshold, and the per-class precision / recall curves
    of all classes as [classes, proposals] prefix sums and a reversed running maximum (the VOC precision envelope) - no
    host read until the caller takes the result.  oracle/eval_ap.py keeps the sequential walk as the checker and
    tests/golden/eval_ap.npz holds the reference implementation's own results."""
    thresholds = [float(t) for t in iou_thresholds]
    n_cls = num_classes - 1
    dev = proposals[0].score_preds.device if proposals else torch.device("cpu")
    n_total = sum(int(p.score_preds.shape[0]) for p in proposals)
    if n_total == 0:
        return torch.zeros((len(thresholds), n_cls), dtype=torch.float32, device=dev)
    conf = torch.cat([p.score_preds for p in proposals]).detach().float()
    classes = torch.cat([p.pt_sem_classes for p in proposals]).detach().long()
    order = torch.argsort(conf, descending=True)

    best_iou, key, gt = [], [], []
    key_base = 0
    for p in proposals:
        n = int(p.score_preds.shape[0])
        labels = p.instance_sem_labels.to(dev)                                  # [scenes, W]
        width = int(labels.shape[1]) if labels.dim() == 2 else 0
        gt.append(labels.reshape(-1).long())
        if n and width:
            sample = p.batch_indices[p.proposal_offsets[:-1].long()].long()
            cls = p.pt_sem_classes.long()
            row = torch.where(labels[sample].long() == cls[:, None], p.ious.detach().float(), p.ious.new_zeros(()).float())
            top, arg = row.max(dim=1)                                           # first maximum, like the sequential walk
            best_iou.append(top)
            key.append(key_base + sample * width + arg)
        else:
            best_iou.append(torch.zeros(n, dtype=torch.float32, device=dev))
            key.append(torch.full((n,), key_base, dtype=torch.int64, device=dev))
        key_base += int(labels.shape[0]) * max(width, 1)
    ranked_iou = torch.cat(best_iou)[order]
    ranked_key = torch.cat(key)[order]
    ranked_cls = classes[order]
    gt_classes = torch.cat(gt) if gt else torch.zeros(0, dtype=torch.int64, device=dev)

    cls_ids = torch.arange(1, num_classes, device=dev)
    of_class = ranked_cls[None, :] == cls_ids[:, None]                          # [C, n]
    num_gt = (gt_classes[None, :] == cls_ids[:, None]).sum(1)                   # [C] int64, as the reference divides
    has_proposals = of_class.any(1)
    rank = torch.arange(n_total, device=dev)
    big = torch.full((max(key_base, 1),), n_total, dtype=torch.int64, device=dev)
    out = []
    ranked_iou64 = ranked_iou.double()  # the reference compares numpy float IoUs with Python-float thresholds in double:
    for thr in thresholds:               # in float32 an IoU on a threshold boundary (0.5, 0.55, ...) can flip TP / FP
        cand = ranked_iou64 > float(thr)
        first = big.scatter_reduce(0, ranked_key, torch.where(cand, rank, rank.new_full((), n_total)), "amin")
        tp = cand & (first[ranked_key] == rank)
        tp_c = (of_class & tp[None, :]).float().cumsum(1)
        fp_c = (of_class & ~tp[None, :]).float().cumsum(1)
        rec = tp_c / num_gt[:, None]
        prec = tp_c / (tp_c + fp_c + 1e-8)
        env = torch.flip(torch.cummax(torch.flip(prec, dims=[1]), dim=1)[0], dims=[1])     # precision envelope
        step = rec - torch.cat([torch.zeros_like(rec[:, :1]), rec[:, :-1]], dim=1)          # recall increments
        ap = (step * env).sum(1)
        out.append(torch.where(has_proposals, ap, torch.zeros_like(ap)))        # a class without proposals scores 0
    return torch.stack(out)


def compute_ap(proposals: List[Instances], num_classes: int = 9, iou_threshold: float = 0.5, device="cpu") -> List[float]:
    """per-class AP at one IoU threshold (grouping_utils.py:420-454) - ``compute_ap_multi`` and one host read."""
    return [float(v) for v in compute_ap_multi(proposals, num_classes, [iou_threshold])[0].cpu()]
