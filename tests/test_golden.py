"""Golden vectors captured from the reference's own Python (tests/golden/make_golden.py imports
/root/reference/gapartnet/{network/losses.py, network/grouping_utils.py, misc/info.py, misc/pose_fitting.py} in the
build container; only inputs/outputs are committed).  They pin the host-side glue this repo re-implements."""
import os

import numpy as np
import torch

from gapartnet_amd.misc import info, pose_fitting
from gapartnet_amd.network import grouping_utils as G
from gapartnet_amd.network import losses as L

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(HERE, name))


def test_losses_match_reference():
    z = load("losses.npz")
    logits, labels, labels_ign = torch.from_numpy(z["logits"]), torch.from_numpy(z["labels"]), torch.from_numpy(z["labels_ign"])
    assert abs(L.focal_loss(logits, labels_ign, gamma=2.0, ignore_index=-100).item() - float(z["focal"])) < 1e-6
    assert abs(L.focal_loss(logits, labels_ign, gamma=2.0, ignore_index=-100, reduction="sum").item() - float(z["focal_sum"])) < 1e-3
    assert abs(L.dice_loss(logits[:, :, None, None], labels[:, None, None]).item() - float(z["dice"])) < 1e-6
    assert L.focal_loss(logits, torch.full_like(labels, -100)).item() == 0.0


def test_score_targets_match_reference():
    z = load("score_targets.npz")
    assert np.allclose(G.get_gt_scores(torch.from_numpy(z["ious"]), 0.75, 0.25).numpy(), z["gt"], atol=1e-7)


def test_symmetry_matrices_match_reference():
    z = load("npcs_loss.npz")
    sm = info.get_symmetry_matrix()
    for mine, name in zip(sm, ("sm1", "sm2", "sm3")):
        assert mine.shape == z[name].shape and np.array_equal(mine.numpy(), z[name])


def test_npcs_loss_matches_reference():
    z = load("npcs_loss.npz")
    sm = info.get_symmetry_matrix()
    pred, gt, prop = torch.from_numpy(z["pred"]), torch.from_numpy(z["gt"]), torch.from_numpy(z["prop"])
    for name, table in (("g1", sm[0]), ("g2", sm[1]), ("g3", sm[2])):
        which = torch.from_numpy(z[f"{name}_which"])
        got = G.compute_npcs_loss(pred, gt, prop, table[which]).item()
        assert abs(got - float(z[f"{name}_loss"])) < 1e-6, name


def test_voc_ap_matches_reference():
    z = load("voc_ap.npz")
    rec, prec = torch.from_numpy(z["rec"]), torch.from_numpy(z["prec"])
    assert abs(G.voc_ap(rec, prec) - float(z["ap"])) < 1e-6
    assert abs(G.voc_ap(rec, prec, use_07_metric=True) - float(z["ap07"])) < 1e-6
    assert abs(G._compute_ap_per_class(torch.from_numpy(z["tp"]), torch.from_numpy(z["fp"]), 37) - float(z["ap_class"])) < 1e-6


def test_umeyama_matches_reference():
    z = load("umeyama.npz")
    scale, rot, trans, T = pose_fitting.estimate_similarity_umeyama(z["src"], z["dst"])
    assert np.allclose(scale, z["scales"], atol=1e-10) and np.allclose(rot, z["rot"], atol=1e-10)
    assert np.allclose(trans, z["trans"], atol=1e-10) and np.allclose(T, z["T"], atol=1e-10)
    assert abs(scale[0] - 0.7) < 1e-2


def test_pose_from_npcs_recovers_box():
    rng = np.random.default_rng(3)
    np.random.seed(0)
    npcs = rng.uniform(-0.5, 0.5, (400, 3)) * np.array([1.0, 0.6, 0.3])
    th = 0.4
    R = np.array([[np.cos(th), np.sin(th), 0], [-np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    xyz = npcs @ (0.8 * R) + np.array([0.2, -0.1, 0.5])
    bbox, scale, rot, trans, T, idx = pose_fitting.estimate_pose_from_npcs(xyz, npcs)
    assert bbox.shape == (8, 3) and abs(scale[0] - 0.8) < 1e-6 and np.allclose(trans, [0.2, -0.1, 0.5], atol=1e-6)
    assert np.allclose(np.abs(bbox - trans).max(), np.abs(xyz - trans).max(), rtol=0.2)


def test_part_tables():
    assert info.PART_ID2NAME[0] == "others" and info.PART_NAME2ID["revolute_handle"] == 9 and len(info.OBJECT_NAME2ID) == 27
    assert info.OBJECT_NAME2ID["StorageFurniture"] == 8 and info.OBJECT_NAME2ID["Suitcase"] == 26
