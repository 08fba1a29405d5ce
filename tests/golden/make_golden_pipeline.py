"""Generates the fixtures that pin the END-TO-END path to the reference's own Python:

    tests/golden/yaml_init_args.json   parsed gapartnet/gapartnet.yaml  model.init_args + data.init_args
    tests/golden/loader.npz            reference dataset/gapartnet.py:85-229 (load_data, compact_instance_labels,
                                       apply_augmentations, generate_inst_info, apply_voxelization) on seeded scenes
    tests/golden/eval_ap.npz           reference network/grouping_utils.py:360-454 compute_ap at the 10 IoU thresholds of
                                       model.py:734 on seeded proposal sets (~10^3 proposals per epoch)
    tests/golden/glue_step.npz         reference network/model.py:466-805 (+ backbone.py, grouping_utils.py,
                                       structure/point_cloud.py:84-189) run UNMODIFIED: one training step (forward +
                                       backward) and one validation epoch on two seeded 2000-point scenes

    python tests/golden/make_golden_pipeline.py        (build container only: needs /root/reference)

How the reference runs here: its modules are imported from /root/reference/gapartnet as they are; the third-party
packages they import and that are absent from the image are provided through ``sys.modules``:
  spconv.pytorch, epic_ops.*   -> this repo's mirrors (gapartnet_amd.spconv / gapartnet_amd.epic_ops) over the CPU
                                  oracle backend (oracle.torch_ops) — the operator arithmetic is the oracle's, the
                                  glue (everything the fixture pins) is the reference's
  lightning.pytorch            -> a 30-line stand-in: LightningModule = nn.Module + save_hyperparameters / log /
                                  current_epoch / device;  LightningDataModule = object
  torchdata.datapipes          -> torch.utils.data.datapipes (dead datapipe code is only DEFINED on import)
  cv2, kornia.metrics          -> empty stubs (only imported, never called on this path; kornia's mean_iou
                                  is replaced by a confusion-matrix mIoU that is NOT pinned — see ``miou_standin``)
Two library calls behave differently in this process, both documented where they are installed
(install_reference_environment): Tensor.cuda() is the identity (no GPU here) and torch.sort is stable (the reference
leaves the order of tied cluster labels unspecified; the fixture pins ascending point index).
Only inputs and outputs are written to the fixtures; no reference source text goes anywhere.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/gapartnet"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.golden import recipe  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _StandInLightningModule(nn.Module):
    """what model.py touches of lightning.pytorch.LightningModule"""

    def __init__(self):
        super().__init__()
        self.current_epoch = 0
        self.logged = {}

    def save_hyperparameters(self, *a, **k):
        pass

    @property
    def device(self):
        return next(self.parameters()).device if any(True for _ in self.parameters()) else torch.device("cpu")

    def log(self, name, value, **kw):
        self.logged.setdefault(name, []).append(float(value))


def miou_standin(pred, target, num_classes, eps=1e-6):
    """confusion-matrix IoU per class with kornia's signature ([B,N] ints -> [B,C]); stand-in, not pinned."""
    out = []
    for p, t in zip(pred, target):
        conf = torch.zeros(num_classes, num_classes, dtype=torch.float64)
        conf.index_put_((t.long(), p.long()), torch.ones_like(p, dtype=torch.float64), accumulate=True)
        diag = conf.diag()
        out.append((diag + eps) / (conf.sum(0) + conf.sum(1) - diag + eps))
    return torch.stack(out).float()


def install_reference_environment():
    import gapartnet_amd.epic_ops as eo
    import gapartnet_amd.spconv
    import gapartnet_amd.spconv.pytorch
    sys.modules["spconv"] = gapartnet_amd.spconv
    sys.modules["spconv.pytorch"] = gapartnet_amd.spconv.pytorch
    sys.modules["epic_ops"] = eo
    for sub in ("voxelize", "ball_query", "ccl", "reduce", "iou", "nms"):
        sys.modules[f"epic_ops.{sub}"] = getattr(eo, sub)
    lp = _stub("lightning.pytorch", LightningModule=_StandInLightningModule, LightningDataModule=object)
    _stub("lightning", pytorch=lp)
    import torch.utils.data.datapipes as torch_dp   # torchdata's datapipes started life here: same class names
    from torch.utils.data import functional_datapipe
    dp_iter = types.SimpleNamespace(ShardingFilter=torch_dp.iter.ShardingFilter, IterableWrapper=torch_dp.iter.IterableWrapper,
                                    IterDataPipe=torch.utils.data.IterDataPipe)
    dp = _stub("torchdata.datapipes", functional_datapipe=functional_datapipe, iter=dp_iter)
    _stub("torchdata", datapipes=dp)
    cv2 = _stub("cv2")
    cv2.__getattr__ = lambda name: 0   # visu_util.py reads a few cv2 constants at import; nothing of cv2 is called
    _stub("kornia")
    _stub("kornia.metrics", mean_iou=miou_standin)
    # apply_nms hard-codes ``.cuda()`` on its inputs (grouping_utils.py:244); there is no GPU in the build container,
    # so for this process Tensor.cuda is the identity (an environment stand-in like the module stubs above)
    torch.Tensor.cuda = lambda self, *a, **k: self
    # cluster_proposals ends in ``torch.sort(cc_labels)`` (grouping_utils.py:139) — an UNSTABLE sort, so the order of the
    # points inside a cluster is unspecified in the reference, and it matters: the score head is trained on the class of
    # each proposal's FIRST point (model.py:553-561).  torch's CPU sort scrambles ties; a radix sort (what a GPU runs for
    # this call) keeps them in ascending index order.  The fixture pins the latter: for this process torch.sort is stable.
    unstable_sort = torch.sort
    torch.sort = lambda input, *a, **k: unstable_sort(input, *a, **{**k, "stable": True})
    sys.path.insert(0, REF)


def _np(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()   # a copy: later in-place updates of the tensor must not reach the fixture
    return np.array(t)


# ---------------------------------------------------------------------------------------------------- yaml
def write_yaml_fixture():
    with open(os.path.join(REF, "gapartnet.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    out = dict(model=dict(class_path=cfg["model"]["class_path"], init_args=cfg["model"]["init_args"]),
               data=dict(class_path=cfg["data"]["class_path"], init_args=cfg["data"]["init_args"]),
               trainer=dict(max_epochs=cfg["trainer"]["max_epochs"]), seed_everything=cfg["seed_everything"])
    with open(os.path.join(HERE, "yaml_init_args.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    return out


# ---------------------------------------------------------------------------------------------------- loader
def _write_pth(directory, name, arrays):
    path = os.path.join(directory, name + ".pth")
    torch.save(tuple(arrays), path)
    return path


def write_loader_fixture(ref_ds):
    """reference loader functions, scene by scene.  Augmented scenes: np.random.seed(seed) right before
    apply_augmentations, so a re-implementation that draws in the same order from the same seed must give the same
    points (flip / rotate branches are both taken and skipped among the seeds, incl. the flip_prob-gates-rotate quirk)."""
    out = {}
    branches = []
    with tempfile.TemporaryDirectory() as tmp:
        for seed in recipe.LOADER_SEEDS:
            arrays = recipe.scene_arrays(seed, recipe.LOADER_POINTS)
            path = _write_pth(tmp, f"Box_{seed}_00_000", arrays)
            pc = ref_ds.load_data(path)
            pc = ref_ds.downsample(pc, max_points=20000)
            pc = ref_ds.compact_instance_labels(pc)
            np.random.seed(seed)
            aug = ref_ds.apply_augmentations(pc, **recipe.AUG)
            # which branches this seed takes (replay of the draw order; informational, stored in the fixture)
            np.random.seed(seed)
            np.random.randn(3, 3)
            flip = np.random.rand() < recipe.AUG["flip_prob"]
            rot = np.random.rand() < recipe.AUG["flip_prob"]
            branches.append((bool(flip), bool(rot)))
            for tag, scene in (("plain", pc), ("aug", aug)):
                full = ref_ds.generate_inst_info(scene).to_tensor()
                full = ref_ds.apply_voxelization(full, voxel_size=(0.01, 0.01, 0.01))
                pre = f"s{seed}_{tag}_"
                out[pre + "points"] = _np(full.points)
                out[pre + "instance_labels"] = _np(full.instance_labels)
                out[pre + "instance_regions"] = _np(full.instance_regions)
                out[pre + "num_points_per_instance"] = _np(full.num_points_per_instance)
                out[pre + "instance_sem_labels"] = _np(full.instance_sem_labels)
                out[pre + "num_instances"] = np.int64(full.num_instances)
                out[pre + "voxel_features"] = _np(full.voxel_features)
                out[pre + "voxel_coords"] = _np(full.voxel_coords)
                out[pre + "voxel_coords_range"] = np.asarray(full.voxel_coords_range, dtype=np.int64)
                out[pre + "pc_voxel_id"] = _np(full.pc_voxel_id)
            out[f"s{seed}_obj_cat"] = np.int64(pc.obj_cat)
    out["branches"] = np.asarray(branches)
    assert out["branches"][:, 0].any() and not out["branches"][:, 0].all(), "want flipped and unflipped seeds"
    assert out["branches"][:, 1].any() and not out["branches"][:, 1].all(), "want rotated and unrotated seeds"
    np.savez_compressed(os.path.join(HERE, "loader.npz"), **out)
    return out


# ---------------------------------------------------------------------------------------------------- pipeline
def _load_scenes(ref_ds, tmp):
    scenes = []
    for seed, cat in recipe.PIPELINE_SCENES:
        arrays = recipe.scene_arrays(seed, recipe.PIPELINE_POINTS)
        path = _write_pth(tmp, f"{cat}_{seed}_00_000", arrays)
        pc = ref_ds.load_data(path)
        pc = ref_ds.compact_instance_labels(ref_ds.downsample(pc, max_points=20000))
        pc = ref_ds.generate_inst_info(pc).to_tensor()
        scenes.append(ref_ds.apply_voxelization(pc, voxel_size=(0.01, 0.01, 0.01)))
    return scenes


class _Tap:
    """records what goes through the reference model's own sub-forwards (instance-level wrappers; the class and its
    source stay untouched)"""

    def __init__(self, model, ref_model_module):
        self.rec = {}
        self.model = model
        for name in ("forward_backbone", "forward_sem_seg", "forward_offset", "proposal_clustering_and_revoxelize",
                     "forward_proposal_score", "forward_proposal_npcs", "loss_proposal_score", "loss_proposal_npcs"):
            self._wrap(name)
        self.clusters = []
        orig = ref_model_module.cluster_proposals

        def cluster_proposals(*a, **k):
            res = orig(*a, **k)
            self.clusters.append(tuple(_np(t) for t in res))
            return res
        ref_model_module.cluster_proposals = cluster_proposals
        self._restore = lambda: setattr(ref_model_module, "cluster_proposals", orig)

    def _wrap(self, name):
        bound = getattr(self.model, name)

        def wrapper(*a, **k):
            res = bound(*a, **k)
            self.rec[name] = res
            return res
        setattr(self.model, name, wrapper)

    def close(self):
        self._restore()
        for name in list(self.rec):
            delattr(self.model, name) if name in self.model.__dict__ else None


def _proposal_fields(prefix, voxel_tensor, pc_voxel_id, proposals, out):
    out[prefix + "voxel_features"] = _np(voxel_tensor.features)
    out[prefix + "voxel_indices"] = _np(voxel_tensor.indices)
    out[prefix + "pc_voxel_id"] = _np(pc_voxel_id)
    for f in ("valid_mask", "sorted_indices", "pt_xyz", "batch_indices", "proposal_offsets", "proposal_indices",
              "num_points_per_proposal", "sem_preds", "instance_labels", "sem_labels", "ious", "score_preds",
              "npcs_preds", "npcs_valid_mask"):
        v = getattr(proposals, f, None)
        if v is not None:
            out[prefix + f] = _np(v)


def write_pipeline_fixture(cfg, ref_ds):
    import network.model as ref_model_module
    init_args = dict(cfg["model"]["init_args"])
    init_args["visualize_cfg"] = dict(init_args["visualize_cfg"], visualize=False)
    model = ref_model_module.GAPartNet(**init_args)
    model.load_state_dict(recipe.name_keyed_state(model))
    out = {"state_keys": np.asarray(list(model.state_dict().keys()))}

    with tempfile.TemporaryDirectory() as tmp:
        # ------------------------------------------------------------------ training step (epoch 10: every head on)
        model.train()
        model.current_epoch = 10
        tap = _Tap(model, ref_model_module)
        torch.manual_seed(recipe.PIPELINE_TORCH_SEED)
        scenes = _load_scenes(ref_ds, tmp)            # collate mutates pc_voxel_id in place: fresh scenes per step
        loss = model.training_step(scenes, 0)
        loss.backward()
        torch.manual_seed(recipe.PIPELINE_TORCH_SEED)
        out["jitter_a"], out["jitter_b"] = torch.rand(3).numpy(), torch.rand(3).numpy()  # the step's two torch.rand(3)
        out["train_pc_feature"] = _np(tap.rec["forward_backbone"])
        out["train_sem_logits"] = _np(tap.rec["forward_sem_seg"])
        out["train_offsets"] = _np(tap.rec["forward_offset"])
        vt, pid, props = tap.rec["proposal_clustering_and_revoxelize"]
        _proposal_fields("train_prop_", vt, pid, props, out)
        out["train_score_logits"] = _np(tap.rec["forward_proposal_score"])
        out["train_npcs_logits"] = _np(tap.rec["forward_proposal_npcs"])
        for i, (labels, order) in enumerate(tap.clusters):
            out[f"train_cc_labels_{i}"], out[f"train_cc_order_{i}"] = labels, order
        for k, v in model.logged.items():
            out["train_log/" + k] = np.asarray(v)
        out["train_loss"] = np.float64(loss.item())
        # gradients: small tensors in full, every tensor's L2 norm and max-abs
        names, norms, maxabs = [], [], []
        for name, p in model.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            names.append(name); norms.append(float(g.double().norm())); maxabs.append(float(g.abs().max()))
            if p.numel() <= 4096:
                out["train_grad/" + name] = _np(g)
        out["grad_names"], out["grad_norms"], out["grad_maxabs"] = np.asarray(names), np.asarray(norms), np.asarray(maxabs)
        # BatchNorm running statistics after the step (momentum 0.1 update from the batch statistics)
        for key in ("backbone.stem.1.running_mean", "backbone.stem.1.running_var", "offset_head.1.running_mean",
                    "score_unet.stem.0.running_var"):
            out["train_buffer/" + key] = _np(model.state_dict()[key])
        tap.close()

        # ------------------------------------------------------------------ validation epoch (eval mode, fresh weights)
        model.load_state_dict(recipe.name_keyed_state(model))
        model.zero_grad(set_to_none=True)
        model.eval()
        model.logged = {}
        tap = _Tap(model, ref_model_module)
        with torch.no_grad():
            for loader_idx in range(3):
                torch.manual_seed(recipe.PIPELINE_TORCH_SEED)
                scenes = _load_scenes(ref_ds, tmp)
                if loader_idx == 1:
                    scenes = scenes[::-1]
                if loader_idx == 2:
                    scenes = scenes[:1]
                pc_ids, sem_seg, kept = model.validation_step(scenes, 0, loader_idx)
                if loader_idx == 0:
                    out["eval_pc_feature"] = _np(tap.rec["forward_backbone"])
                    out["eval_sem_logits"] = _np(tap.rec["forward_sem_seg"])
                    out["eval_offsets"] = _np(tap.rec["forward_offset"])
                    vt, pid, props = tap.rec["proposal_clustering_and_revoxelize"]
                    _proposal_fields("eval_prop_", vt, pid, props, out)
                    out["eval_score_logits"] = _np(tap.rec["forward_proposal_score"])
                    out["eval_npcs_logits"] = _np(tap.rec["forward_proposal_npcs"])
                    out["eval_all_accu"], out["eval_pixel_accu"] = np.float64(sem_seg.all_accu), np.float64(sem_seg.pixel_accu)
                    for f in ("score_preds", "pt_sem_classes", "batch_indices", "instance_sem_labels", "ious",
                              "proposal_offsets", "valid_mask"):
                        out["eval_kept_" + f] = _np(getattr(kept, f))
                    out["eval_pc_ids"] = np.asarray(pc_ids)
            model.on_validation_epoch_end()
        for k, v in model.logged.items():
            out["eval_log/" + k] = np.asarray(v)
        tap.close()
    np.savez_compressed(os.path.join(HERE, "glue_step.npz"), **out)
    return out


# ---------------------------------------------------------------------------------------------------- eval AP
def ap_case_sets(seed: int):
    """seeded per-batch proposal sets for compute_ap (inputs of the fixture; also what the tests rebuild): distinct
    confidences (argsort of ties is unspecified in the reference), IoUs quantised to 1/8 so that ties between ground-truth
    instances and thresholds hit exactly, padded instance labels (-1)."""
    rng = np.random.default_rng(seed)
    sets = []
    for _ in range(int(rng.integers(2, 5))):
        n_prop, n_scenes, width = int(rng.integers(150, 400)), int(rng.integers(2, 9)), int(rng.integers(3, 13))
        sizes = rng.integers(4, 40, n_prop)
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        scene_of_prop = np.sort(rng.integers(0, n_scenes, n_prop))
        labels = rng.integers(1, 10, (n_scenes, width)).astype(np.int32)
        labels[rng.random((n_scenes, width)) < 0.2] = -1
        classes = rng.integers(1, 10, n_prop)
        ious = (rng.random((n_prop, width)) ** 3).astype(np.float32)
        hit = rng.random(n_prop) < 0.6                      # most proposals overlap one instance well
        tgt = rng.integers(0, width, n_prop)
        ious[hit, tgt[hit]] = 0.4 + 0.6 * rng.random(int(hit.sum())).astype(np.float32)
        same = rng.random(n_prop) < 0.7                     # ... and usually predict that instance's class
        lab_t = labels[scene_of_prop, tgt]
        classes = np.where(hit & same & (lab_t > 0), lab_t, classes)
        ious = (np.round(ious * 8) / 8).astype(np.float32)
        conf = (rng.permutation(n_prop).astype(np.float32) + 0.5) / n_prop
        sets.append(dict(score_preds=conf, pt_sem_classes=classes.astype(np.int64),
                         batch_indices=np.repeat(scene_of_prop, sizes).astype(np.int32), proposal_offsets=offsets,
                         instance_sem_labels=labels, ious=ious))
    return sets


def write_eval_ap_fixture():
    """reference grouping_utils.compute_ap (:360-454, the sequential Python walk) at the ten thresholds of
    on_validation_epoch_end (model.py:734) on three seeded epochs of 10^3 proposals each."""
    import network.grouping_utils as ref_grouping
    from structure.instances import Instances
    out = {}
    for seed in (0, 1, 2):
        sets = ap_case_sets(seed)
        insts = [Instances(**{k: torch.from_numpy(v) for k, v in d.items()}) for d in sets]
        out[f"case{seed}_num_sets"] = np.int64(len(sets))
        for i, d in enumerate(sets):
            for k, v in d.items():
                out[f"case{seed}_set{i}_{k}"] = v
        thresholds = [0.5 + 0.05 * i for i in range(10)]
        out[f"case{seed}_thresholds"] = np.asarray(thresholds)
        out[f"case{seed}_aps"] = np.asarray([ref_grouping.compute_ap(insts, 10, t) for t in thresholds])
    np.savez_compressed(os.path.join(HERE, "eval_ap.npz"), **out)
    return out


def main():
    assert os.path.isdir(REF), "reference tree not present: fixtures can only be regenerated in the build container"
    os.environ.setdefault("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")  # the reference's torch.load(path) reads numpy tuples
    install_reference_environment()
    from gapartnet_amd import backend
    from oracle import torch_ops
    cfg = write_yaml_fixture()
    with backend.using(torch_ops):
        import dataset.gapartnet as ref_ds
        loader = write_loader_fixture(ref_ds)
        pipe = write_pipeline_fixture(cfg, ref_ds)
        aps = write_eval_ap_fixture()
    print("eval_ap.npz: mean AP per case", [float(np.nanmean(aps[f"case{c}_aps"])) for c in (0, 1, 2)],
          os.path.getsize(os.path.join(HERE, "eval_ap.npz")) / 1e6, "MB")
    for name, d in (("loader.npz", loader), ("glue_step.npz", pipe)):
        print(f"{name}: {len(d)} arrays, {os.path.getsize(os.path.join(HERE, name)) / 1e6:.2f} MB")
    print("proposals (train):", pipe["train_prop_proposal_offsets"].shape[0] - 1,
          " (eval):", pipe["eval_prop_proposal_offsets"].shape[0] - 1,
          " kept after NMS:", pipe["eval_kept_proposal_offsets"].shape[0] - 1)
    print({k: v for k, v in pipe.items() if k.startswith("train_log/")})
    print({k: v for k, v in pipe.items() if k.startswith("eval_log/")})


if __name__ == "__main__":
    main()
