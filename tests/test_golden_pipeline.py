"""End-to-end pin against the reference's OWN model code (SURVEY.md App. B ``glue_step.npz``).

tests/golden/glue_step.npz was produced in the build container by running /root/reference/gapartnet/network/model.py
(:466-805: _training_or_validation_step, validation_step, on_validation_epoch_end), backbone.py, grouping_utils.py and
structure/point_cloud.py:84-189 UNMODIFIED (tests/golden/make_golden_pipeline.py explains the stand-ins for the absent
third-party packages).  Inputs: two seeded 4000-point scenes, GAPartNet(**gapartnet.yaml init_args) with name-keyed
seeded weights (tests/golden/recipe.py), the two torch.rand(3) re-voxelisation jitters.  Pinned: backbone features,
sem_logits, offsets, both cluster label sets, the proposal CSR and every per-point field, the proposal voxel grid,
score_logits, npcs_logits, every loss term and logged metric, parameter gradients, BatchNorm running statistics; in eval
mode additionally the filtered + NMS-ed proposals and the epoch-end log.

This repo's GAPartNet must reproduce all of it:
  * over the CPU oracle operators                         [not gpu]   integers equal, fp <= 1e-5
  * over libgpn_hip.so on the MI355X                      [gpu]       integers equal, fp <= 1e-4 (north_star), gradients
                                                                      |d| <= 1e-3 max|g| per tensor
Also here: the yaml contract (tests/golden/yaml_init_args.json = parsed gapartnet/gapartnet.yaml) and compute_ap against
the reference's sequential implementation (tests/golden/eval_ap.npz).
"""
import os

import numpy as np
import pytest
import torch

from gapartnet_amd import backend
from tests import pipeline_runner as R

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

INT_KEYS = ("prop_voxel_indices", "prop_pc_voxel_id", "prop_valid_mask", "prop_sorted_indices", "prop_batch_indices",
            "prop_proposal_offsets", "prop_proposal_indices", "prop_num_points_per_proposal", "prop_sem_preds",
            "prop_instance_labels", "prop_sem_labels", "prop_npcs_valid_mask")
FP_KEYS = ("pc_feature", "sem_logits", "offsets", "prop_voxel_features", "prop_pt_xyz", "prop_ious", "prop_score_preds",
           "prop_npcs_preds", "score_logits", "npcs_logits")


# Gradient bound of the GPU run against the reference model's gradients: max|d| <= bound x max|g| per tensor.  Gradients pass
# through ~70 training-mode BatchNorms whose batch statistics on the few-hundred-row deep levels amplify fp32 summation-order
# differences, so this is looser than north_star's 1e-4 on features; it is set at ~2x the worst error the shipped kernels
# achieve (printed by the test, recorded in profiles/r04_golden_grad_errors.json), not at a round number.
GPU_GRAD_BOUND = 8e-4  # achieved (round 4): 3.8e-4 worst of 228 tensors (a BatchNorm bias of level 3's decoder), 2.2e-4 next


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "glue_step.npz"))


def _check_forward(gold, out, prefix, tol, optional=()):
    for k in INT_KEYS:
        if k in optional and prefix + k not in out:
            continue
        assert np.array_equal(out[prefix + k], gold[prefix + k]), f"{prefix}{k}: integer output differs from the reference"
    worst = {}
    for k in FP_KEYS:
        if k in optional and prefix + k not in out:
            continue
        g, v = gold[prefix + k], out[prefix + k]
        assert g.shape == v.shape, (k, g.shape, v.shape)
        err = float(np.abs(g.astype(np.float64) - v).max()) if g.size else 0.0
        scale = max(1.0, float(np.abs(g).max())) if g.size else 1.0
        worst[k] = err / scale
        assert err <= tol * scale, f"{prefix}{k}: |d| {err:.3e} > {tol} x {scale:.3g}"
    return worst


def _check_logs(gold, out, prefix, tol):
    keys = [k for k in gold.files if k.startswith(prefix)]
    assert keys and sorted(keys) == sorted(k for k in out if k.startswith(prefix)), "logged keys differ from the reference"
    for k in keys:
        g, v = gold[k], np.asarray(out[k])
        assert g.shape == v.shape, k
        assert np.array_equal(np.isnan(g), np.isnan(v)), k
        assert np.allclose(g, v, rtol=tol, atol=tol, equal_nan=True), (k, g, v)


def _check_gradients(gold, grads, rel):
    """per tensor: |d| <= rel * max|g| on the tensors stored in full, L2 norm within rel everywhere; tensors whose
    gradient is structurally zero in the reference (a bias in front of BatchNorm) must be ~zero here as well"""
    names = [str(n) for n in gold["grad_names"]]
    assert names == list(grads.keys()), "parameter names / order differ from the reference model"
    # scale of a 'zero' gradient: relative to the largest gradient entry of the whole model
    top = float(gold["grad_maxabs"].max())
    achieved = {}  # per tensor stored in full: max|d| / max|g| (what the bound is set from: 2 x the worst of these)
    for name, norm, maxabs in zip(names, gold["grad_norms"], gold["grad_maxabs"]):
        mine = grads[name].astype(np.float64)
        if maxabs <= 1e-6 * top:
            assert np.abs(mine).max() <= 1e-5 * top, f"{name}: structurally zero gradient is {np.abs(mine).max():.3e}"
            continue
        assert abs(np.linalg.norm(mine) - norm) <= rel * norm + 1e-7, f"{name}: grad norm {np.linalg.norm(mine)} vs {norm}"
        key = "train_grad/" + name
        if key in gold.files:
            err = np.abs(mine - gold[key]).max()
            achieved[name] = float(err / maxabs)
            assert err <= rel * maxabs + 1e-7, f"{name}: grad |d| {err:.3e} > {rel} x max|g| {maxabs:.3e}"
    return achieved


def _check_buffers(gold, buffers, tol):
    for k in gold.files:
        if k.startswith("train_buffer/"):
            assert np.allclose(buffers[k[len("train_buffer/"):]], gold[k], rtol=tol, atol=tol), k


def _check_kept(gold, out, tol):
    for f in ("pt_sem_classes", "batch_indices", "instance_sem_labels", "proposal_offsets", "valid_mask"):
        assert np.array_equal(out["eval_kept_" + f], gold["eval_kept_" + f]), f"kept proposals: {f}"
    for f in ("score_preds", "ious"):
        assert np.allclose(out["eval_kept_" + f], gold["eval_kept_" + f], rtol=0, atol=tol), f"kept proposals: {f}"
    assert list(out["eval_pc_ids"]) == list(gold["eval_pc_ids"])


# ------------------------------------------------------------------------------------------------ yaml contract
def test_yaml_contract_builds_the_model_and_the_data_module(gold):
    """GAPartNet(**model.init_args) / GAPartNetInst(**data.init_args) straight from the reference's gapartnet.yaml;
    state_dict keys = the reference model's (so release.ckpt loads with strict=False and nothing is missing)"""
    from gapartnet_amd.dataset.gapartnet import GAPartNetInst
    from gapartnet_amd.network.model import GAPartNet
    cfg = R.yaml_args()
    assert cfg["model"]["class_path"] == "network.model.GAPartNet"
    assert cfg["data"]["class_path"] == "dataset.gapartnet.GAPartNetInst"
    model = GAPartNet(**cfg["model"]["init_args"])
    assert list(model.state_dict().keys()) == [str(k) for k in gold["state_keys"]]
    assert (model.start_scorenet, model.start_npcs, model.start_clustering) == (5, 10, 5)
    assert model.symmetry_indices.tolist() == cfg["model"]["init_args"]["symmetry_indices"]
    assert sum(p.numel() for p in model.parameters()) == 7_897_617
    dm = GAPartNetInst(**cfg["data"]["init_args"])
    assert (dm.train_batch_size, dm.val_batch_size, dm.test_batch_size, dm.num_workers) == (64, 32, 32, 16)
    assert dm.aug == dict(pos_jitter=0.1, color_jitter=0.3, flip_prob=0.3, rotate_prob=0.3)
    assert tuple(dm.voxel_size) == (0.01, 0.01, 0.01) and dm.max_points == 20000 and dm.few_shot_num == 640
    opt = model.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-3


# ------------------------------------------------------------------------------------------------ CPU (oracle operators)
def test_training_step_matches_the_reference_model_on_the_oracle(gold):
    from oracle import torch_ops
    with backend.using(torch_ops):
        out = R.run_train_step(gold, "cpu")
    _check_forward(gold, out, "train_", 1e-5)
    _check_logs(gold, out, "train_log/", 1e-5)
    assert abs(out["train_loss"] - float(gold["train_loss"])) <= 1e-5 * float(gold["train_loss"])
    _check_gradients(gold, out["grads"], 1e-4)
    _check_buffers(gold, out["buffers"], 1e-5)


def test_validation_epoch_matches_the_reference_model_on_the_oracle(gold):
    from oracle import torch_ops
    with backend.using(torch_ops):
        out = R.run_validation_epoch(gold, "cpu")
    _check_forward(gold, out, "eval_", 1e-5)
    _check_kept(gold, out, 1e-5)
    _check_logs(gold, out, "eval_log/", 1e-5)


def test_batched_device_voxelisation_gives_the_reference_batch(gold):
    """the product path voxelises the whole batch at collate time instead of per scene in the loader
    (dataset/gapartnet.py:188 there): same voxels, same order, same point->voxel map, same step"""
    from oracle import torch_ops
    with backend.using(torch_ops):
        out = R.run_train_step(gold, "cpu", per_scene_voxelisation=False)
    _check_forward(gold, out, "train_", 1e-5)
    _check_logs(gold, out, "train_log/", 1e-5)


# ------------------------------------------------------------------------------------------------ GPU (HIP operators)
@pytest.mark.gpu
@pytest.mark.parametrize("per_scene", [True, False])
def test_training_step_on_the_gpu_matches_the_reference_model(cuda, gold, per_scene):
    out = R.run_train_step(gold, cuda, per_scene_voxelisation=per_scene)
    worst = _check_forward(gold, out, "train_", 1e-4)
    print("worst fp error / scale per output:", {k: f"{v:.2e}" for k, v in worst.items()})
    _check_logs(gold, out, "train_log/", 1e-4)
    achieved = _check_gradients(gold, out["grads"], GPU_GRAD_BOUND)
    top5 = sorted(achieved.items(), key=lambda kv: -kv[1])[:5]
    print("gradient error / max|g|, worst five tensors (bound %.1e):" % GPU_GRAD_BOUND, {k: f"{v:.2e}" for k, v in top5})
    report = os.environ.get("GPN_GOLDEN_GRAD_REPORT")  # a directory: the achieved errors GPU_GRAD_BOUND is derived from are kept
    if report:                                          # there (tools/final_measure.sh sets it; a plain test run writes nothing)
        import json
        os.makedirs(report, exist_ok=True)
        with open(os.path.join(report, f"golden_grad_errors_per_scene{int(per_scene)}.json"), "w") as fh:
            json.dump(dict(bound=GPU_GRAD_BOUND, worst=dict(top5), n_tensors=len(achieved)), fh, indent=1)
    _check_buffers(gold, out["buffers"], 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("record_npcs_preds", [True, False])
def test_validation_epoch_on_the_gpu_matches_the_reference_model(cuda, gold, record_npcs_preds):
    """record_npcs_preds=False is the product's default validation step (round 5): fused post-processing, and from the second
    step of the epoch on the device-counted proposal stage - the reference's kept proposals and every logged epoch-end metric (AP at
    ten thresholds, mIoU, accuracies over the three loaders) must come out the same; the NPCS predictions a validation step
    never reads are not kept then"""
    out = R.run_validation_epoch(gold, cuda, record_npcs_preds=record_npcs_preds)
    optional = () if record_npcs_preds else ("prop_npcs_preds", "prop_gt_npcs", "prop_npcs_valid_mask")
    worst = _check_forward(gold, out, "eval_", 1e-4, optional)
    print("worst fp error / scale per output:", {k: f"{v:.2e}" for k, v in worst.items()})
    _check_kept(gold, out, 1e-4)
    _check_logs(gold, out, "eval_log/", 1e-4)


# ------------------------------------------------------------------------------------------------ epoch-end AP
def _ap_sets(z, case, device="cpu"):
    from gapartnet_amd.structure.instances import Instances
    sets = []
    for i in range(int(z[f"case{case}_num_sets"])):
        f = {k: torch.from_numpy(z[f"case{case}_set{i}_{k}"]).to(device)
             for k in ("score_preds", "pt_sem_classes", "batch_indices", "proposal_offsets", "instance_sem_labels", "ious")}
        sets.append(Instances(**f))
    return sets


@pytest.mark.parametrize("case", [0, 1, 2])
def test_compute_ap_matches_the_reference_walk(case):
    """grouping_utils.compute_ap (array form) vs the reference's per-proposal Python loop (grouping_utils.py:360-454)
    at the ten thresholds of on_validation_epoch_end, ~10^3 proposals per epoch, IoU ties included"""
    from gapartnet_amd.network import grouping_utils as G
    z = np.load(os.path.join(HERE, "eval_ap.npz"))
    sets = _ap_sets(z, case)
    for t, want in zip(z[f"case{case}_thresholds"], z[f"case{case}_aps"]):
        got = np.asarray(G.compute_ap(sets, 10, float(t)))
        assert np.allclose(got, want, rtol=0, atol=1e-6, equal_nan=True), (t, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 2])
def test_compute_ap_on_the_gpu_matches_the_reference_walk(cuda, case):
    from gapartnet_amd.network import grouping_utils as G
    z = np.load(os.path.join(HERE, "eval_ap.npz"))
    sets = _ap_sets(z, case, cuda)
    for t, want in zip(z[f"case{case}_thresholds"], z[f"case{case}_aps"]):
        got = np.asarray(G.compute_ap(sets, 10, float(t)))
        assert np.allclose(got, want, rtol=0, atol=1e-6, equal_nan=True), (t, got, want)
