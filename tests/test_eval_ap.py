"""Array-form AP matching (grouping_utils.compute_ap) against the sequential walk of the reference
(oracle/eval_ap.py restates network/grouping_utils.py:360-454): identical true / false positive flags, identical APs."""
import numpy as np
import pytest
import torch

from gapartnet_amd.network import grouping_utils as gu
from gapartnet_amd.structure.instances import Instances
from oracle import eval_ap


def _random_set(rng, n_prop, n_scenes, width, quantised):
    sizes = rng.integers(1, 6, n_prop)
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    scene_of_prop = np.sort(rng.integers(0, n_scenes, n_prop))
    batch_indices = np.repeat(scene_of_prop, sizes)
    ious = rng.random((n_prop, width)).astype(np.float32)
    if quantised:  # many exact ties: the first maximum must win in both forms
        ious = np.round(ious * 4) / 4
    labels = rng.integers(-1, 9, (n_scenes, width)).astype(np.int32)
    conf = rng.random(n_prop).astype(np.float32)
    if quantised:
        conf = np.round(conf * 8) / 8
    return Instances(score_preds=torch.from_numpy(conf), pt_sem_classes=torch.from_numpy(rng.integers(1, 9, n_prop)),
                     batch_indices=torch.from_numpy(batch_indices), proposal_offsets=torch.from_numpy(offsets),
                     instance_sem_labels=torch.from_numpy(labels), ious=torch.from_numpy(ious))


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("quantised", [False, True])
def test_array_form_equals_the_sequential_walk(seed, quantised):
    rng = np.random.default_rng(seed)
    sets = [_random_set(rng, int(rng.integers(0, 60)), int(rng.integers(1, 5)), int(rng.integers(1, 12)), quantised)
            for _ in range(int(rng.integers(1, 5)))]
    if all(s.score_preds.shape[0] == 0 for s in sets):
        sets.append(_random_set(rng, 10, 2, 5, quantised))
    for thr in (0.25, 0.5, 0.75):
        want = eval_ap.compute_ap_sequential(sets, 9, thr)
        got = gu.compute_ap(sets, 9, thr)
        # a class without ground truth gives 0/0 = nan in the reference as well; the fp32 sum over ALL ranks (zeros where
        # the recall does not move) rounds differently from the reference's sum over the recall steps only: <= 1e-6
        assert np.allclose(np.asarray(got), np.asarray(want), rtol=0, atol=1e-6, equal_nan=True), (thr, got, want)


def test_no_proposals_at_all():
    empty = Instances(score_preds=torch.zeros(0), pt_sem_classes=torch.zeros(0, dtype=torch.int64),
                      batch_indices=torch.zeros(0, dtype=torch.int64), proposal_offsets=torch.zeros(1, dtype=torch.int64),
                      instance_sem_labels=torch.full((2, 3), 1, dtype=torch.int32), ious=torch.zeros((0, 3)))
    assert gu.compute_ap([empty], 9, 0.5) == [0.0] * 8
