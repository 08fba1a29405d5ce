"""TEST INFRASTRUCTURE - sequential restatement of the reference's AP matching walk (network/grouping_utils.py:360-454 of
the reference: proposals visited one by one in descending confidence, greedy matching against still-unmatched ground-truth
instances).  ``gapartnet_amd.network.grouping_utils.compute_ap`` evaluates the same rule with array operations; this is
its checker (tests/test_eval_ap.py).  Pinned by tests/golden/voc_ap.npz for the VOC integration it shares with the product.
Only tests may import this module."""
from typing import List

import numpy as np
import torch


def compute_tp_fp_sequential(proposals, iou_threshold: float = 0.5):
    """-> (tp, fp, classes) in descending-confidence order, by the reference's per-proposal loop"""
    conf = torch.cat([p.score_preds for p in proposals]).detach().cpu()
    classes = torch.cat([p.pt_sem_classes for p in proposals]).detach().cpu().numpy()
    order = torch.argsort(conf, descending=True).numpy()
    n_total = conf.shape[0]
    set_of = np.concatenate([np.full(p.score_preds.shape[0], i, np.int64) for i, p in enumerate(proposals)])
    sample_of = np.concatenate([p.batch_indices[p.proposal_offsets[:-1].long()].long().cpu().numpy() for p in proposals])
    local_of = np.concatenate([np.arange(p.score_preds.shape[0]) for p in proposals])
    inst_labels = [p.instance_sem_labels.detach().cpu().numpy() for p in proposals]
    ious = [p.ious.detach().cpu().numpy() for p in proposals]
    matched = [np.zeros(l.shape, dtype=bool) for l in inst_labels]
    tp = np.zeros(n_total, np.float32)
    fp = np.zeros(n_total, np.float32)
    for rank, idx in enumerate(order):
        s, smp, loc, cls = set_of[idx], sample_of[idx], local_of[idx], classes[idx]
        row = np.where(inst_labels[s][smp] == cls, ious[s][loc], 0.0)
        best = int(row.argmax()) if row.shape[0] else 0
        best_iou = float(row[best]) if row.shape[0] else 0.0
        if best_iou > iou_threshold and not matched[s][smp, best]:
            tp[rank] = 1.0
            matched[s][smp, best] = True
        else:
            fp[rank] = 1.0
    return tp, fp, classes[order]


def compute_ap_sequential(proposals, num_classes: int = 9, iou_threshold: float = 0.5) -> List[float]:
    from gapartnet_amd.network.grouping_utils import _compute_ap_per_class
    tp, fp, sorted_classes = compute_tp_fp_sequential(proposals, iou_threshold)
    inst_labels = [p.instance_sem_labels.detach().cpu().numpy() for p in proposals]
    gt_classes = np.concatenate([l.reshape(-1) for l in inst_labels])
    tp_t, fp_t = torch.from_numpy(tp), torch.from_numpy(fp)
    return [_compute_ap_per_class(tp_t[torch.from_numpy(sorted_classes == c)], fp_t[torch.from_numpy(sorted_classes == c)],
                                  int((gt_classes == c).sum())) for c in range(1, num_classes)]
