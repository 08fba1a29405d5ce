"""Generates tests/golden/*.npz by IMPORTING the reference's Python (only possible in the build container, where
/root/reference exists; the fixtures — inputs and expected outputs, no reference source — are what is committed).

    python tests/golden/make_golden.py

Stubs are injected into sys.modules for the third-party packages the reference imports but that are absent here
(kornia, epic_ops); no reference file is modified.  Functions captured:
  network/losses.py         focal_loss, dice_loss
  network/grouping_utils.py get_gt_scores, compute_npcs_loss, voc_ap, _compute_ap_per_class
  misc/info.py              get_symmetry_matrix
  misc/pose_fitting.py      estimate_similarity_umeyama
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/gapartnet"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    assert os.path.isdir(REF), "reference tree not present: fixtures can only be regenerated in the build container"
    sys.path.insert(0, REF)
    _stub("kornia")
    _stub("kornia.metrics", mean_iou=lambda *a, **k: None)
    _stub("epic_ops")
    for sub, fn in (("ball_query", "ball_query"), ("ccl", "connected_components_labeling"), ("nms", "nms"),
                    ("reduce", "segmented_reduce"), ("voxelize", "voxelize")):
        _stub(f"epic_ops.{sub}", **{fn: None})
    from misc.info import get_symmetry_matrix
    from misc.pose_fitting import estimate_similarity_umeyama
    from network import grouping_utils as G
    from network import losses as L

    g = torch.Generator().manual_seed(1234)
    # ---- losses
    logits = torch.randn(257, 10, generator=g) * 2
    labels = torch.randint(0, 10, (257,), generator=g)
    labels_ign = labels.clone()
    labels_ign[::7] = -100
    np.savez(os.path.join(OUT, "losses.npz"), logits=logits.numpy(), labels=labels.numpy(), labels_ign=labels_ign.numpy(),
             focal=L.focal_loss(logits, labels_ign, gamma=2.0, ignore_index=-100).item(),
             focal_sum=L.focal_loss(logits, labels_ign, gamma=2.0, ignore_index=-100, reduction="sum").item(),
             dice=L.dice_loss(logits[:, :, None, None], labels[:, None, None]).item())
    # ---- score targets
    ious = torch.tensor([0.0, 0.1, 0.25, 0.3, 0.5, 0.74, 0.75, 0.76, 0.9, 1.0])
    np.savez(os.path.join(OUT, "score_targets.npz"), ious=ious.numpy(), gt=G.get_gt_scores(ious, 0.75, 0.25).numpy())
    # ---- symmetry matrices + npcs loss per group
    sm = get_symmetry_matrix()
    out = dict(sm1=sm[0].numpy(), sm2=sm[1].numpy(), sm3=sm[2].numpy())
    n = 300
    pred = torch.rand(n, 3, generator=g) - 0.5
    gt = torch.rand(n, 3, generator=g) - 0.5
    prop = torch.sort(torch.randint(0, 12, (n,), generator=g))[0]
    out.update(pred=pred.numpy(), gt=gt.numpy(), prop=prop.numpy())
    for name, table, hi in (("g1", sm[0], 3), ("g2", sm[1], 1), ("g3", sm[2], 1)):
        which = torch.randint(0, hi, (n,), generator=g)
        out[f"{name}_which"] = which.numpy()
        out[f"{name}_loss"] = G.compute_npcs_loss(pred, gt, prop, table[which]).item()
    np.savez(os.path.join(OUT, "npcs_loss.npz"), **out)
    # ---- voc_ap / per-class AP
    rec = torch.tensor([0.2, 0.4, 0.4, 0.8])
    prec = torch.tensor([1.0, 1.0, 0.66, 0.75])
    tp = (torch.rand(50, generator=g) > 0.4).float()
    fp = 1 - tp
    np.savez(os.path.join(OUT, "voc_ap.npz"), rec=rec.numpy(), prec=prec.numpy(), ap=G.voc_ap(rec, prec),
             ap07=G.voc_ap(rec, prec, use_07_metric=True), tp=tp.numpy(), fp=fp.numpy(),
             ap_class=G._compute_ap_per_class(tp, fp, 37))
    # ---- umeyama similarity fit
    rng = np.random.default_rng(7)
    src = rng.uniform(-0.5, 0.5, (3, 200))
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    dst = 0.7 * R @ src + np.array([[0.1], [-0.2], [0.3]]) + rng.normal(0, 1e-3, (3, 200))
    src_h = np.vstack([src, np.ones((1, 200))])
    dst_h = np.vstack([dst, np.ones((1, 200))])
    scales, rot, trans, T = estimate_similarity_umeyama(src_h, dst_h)
    np.savez(os.path.join(OUT, "umeyama.npz"), src=src_h, dst=dst_h, scales=scales, rot=rot, trans=trans, T=T)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
