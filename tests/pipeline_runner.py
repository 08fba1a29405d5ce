"""Runs THIS repo's GAPartNet on the inputs of tests/golden/glue_step.npz and records the same intermediates the fixture
holds for the reference's model.py (see tests/golden/make_golden_pipeline.py).  Used by the CPU test (oracle backend) and
the GPU test (HIP backend): same code, only the device / raw-op backend differ."""
import json
import os

import numpy as np
import torch

from gapartnet_amd.dataset import gapartnet as ds
from gapartnet_amd.structure.point_cloud import PointCloud
from tests.golden import recipe

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def yaml_args():
    with open(os.path.join(HERE, "yaml_init_args.json")) as fh:
        return json.load(fh)


def build_model(device):
    """GAPartNet(**gapartnet.yaml model.init_args) with the fixture's name-keyed weights"""
    from gapartnet_amd.network.model import GAPartNet
    init_args = dict(yaml_args()["model"]["init_args"])
    init_args["visualize_cfg"] = dict(init_args["visualize_cfg"], visualize=False)
    model = GAPartNet(**init_args)
    model.load_state_dict(recipe.name_keyed_state(model))
    return model.to(device)


def load_scenes(device, per_scene_voxelisation=True):
    """the fixture's scenes through this repo's loader functions (the reference contract: per-scene voxelisation in the
    loader, concatenated by collate); ``per_scene_voxelisation=False`` leaves it to the batched device path"""
    scenes = []
    for seed, cat in recipe.PIPELINE_SCENES:
        xyz, rgb, sem, ins, npcs, _ = recipe.scene_arrays(seed, recipe.PIPELINE_POINTS)
        pc = PointCloud(pc_id=f"{cat}_{seed}_00_000", obj_cat=0, points=np.concatenate([xyz, rgb], 1).astype(np.float32),
                        sem_labels=sem.astype(np.int64), instance_labels=ins.astype(np.int32), gt_npcs=npcs.astype(np.float32))
        pc = ds.generate_inst_info(ds.compact_instance_labels(pc)).to_tensor().to(device)
        if per_scene_voxelisation:
            pc = ds.apply_voxelization(pc, voxel_size=(0.01, 0.01, 0.01))
        scenes.append(pc)
    return scenes


class Tap:
    """instance-level recorders on the model's sub-forwards (same method names as the reference's)"""
    NAMES = ("forward_backbone", "forward_sem_seg", "forward_offset", "proposal_clustering_and_revoxelize",
             "forward_proposal_score", "forward_proposal_npcs")

    def __init__(self, model):
        self.rec, self.model = {}, model
        for name in self.NAMES:
            self._wrap(name)

    def _wrap(self, name):
        bound = getattr(self.model, name)

        def wrapper(*a, **k):
            res = bound(*a, **k)
            self.rec[name] = res
            return res
        setattr(self.model, name, wrapper)

    def close(self):
        for name in self.NAMES:
            if name in self.model.__dict__:
                delattr(self.model, name)


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


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


def _record_forward(prefix, tap, out):
    out[prefix + "pc_feature"] = _np(tap.rec["forward_backbone"])
    out[prefix + "sem_logits"] = _np(tap.rec["forward_sem_seg"])
    out[prefix + "offsets"] = _np(tap.rec["forward_offset"])
    vt, pid, props = tap.rec["proposal_clustering_and_revoxelize"]
    _proposal_fields(prefix + "prop_", vt, pid, props, out)
    out[prefix + "score_logits"] = _np(tap.rec["forward_proposal_score"])
    out[prefix + "npcs_logits"] = _np(tap.rec["forward_proposal_npcs"])


def run_train_step(gold, device, per_scene_voxelisation=True):
    model = build_model(device).train()
    model.record_npcs_preds = True  # the fixture pins the selected NPCS predictions of the training step as well
    model._current_epoch = 10
    logged = {}
    model._log_sink = lambda name, value, bs, sync: logged.setdefault(name, []).append(float(value))
    model.revoxelize_jitter = (torch.from_numpy(gold["jitter_a"]).to(device), torch.from_numpy(gold["jitter_b"]).to(device))
    tap = Tap(model)
    loss = model.training_step(load_scenes(device, per_scene_voxelisation), 0)
    loss.backward()
    out = {"train_loss": float(loss)}
    _record_forward("train_", tap, out)
    for k, v in logged.items():
        out["train_log/" + k] = np.asarray(v)
    out["grads"] = {n: (_np(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
                    for n, p in model.named_parameters()}
    sd = model.state_dict()
    out["buffers"] = {k: _np(v) for k, v in sd.items() if "running_" in k}
    tap.close()
    return out


def run_validation_epoch(gold, device, per_scene_voxelisation=True, record_npcs_preds=True):
    model = build_model(device).eval()
    # True: the fixture pins the selected NPCS predictions of the validation step as well (a host read; the torch formulation of the
    # post-processing).  False: the product's default validation step - device-counted proposal stage from the second step on,
    # fused post-processing (csrc/postprocess.hip)
    model.record_npcs_preds = record_npcs_preds
    model._current_epoch = 10
    logged = {}
    model._log_sink = lambda name, value, bs, sync: logged.setdefault(name, []).append(float(value))
    model.revoxelize_jitter = (torch.from_numpy(gold["jitter_a"]).to(device), torch.from_numpy(gold["jitter_b"]).to(device))
    tap = Tap(model)
    out = {}
    with torch.no_grad():
        for loader_idx in range(3):
            scenes = load_scenes(device, per_scene_voxelisation)
            if loader_idx == 1:
                scenes = scenes[::-1]
            if loader_idx == 2:
                scenes = scenes[:1]
            pc_ids, sem_seg, kept = model.validation_step(scenes, 0, loader_idx)
            if loader_idx == 0:
                _record_forward("eval_", tap, out)
                out["eval_all_accu"], out["eval_pixel_accu"] = float(sem_seg.all_accu), float(sem_seg.pixel_accu)
                for f in ("score_preds", "pt_sem_classes", "batch_indices", "instance_sem_labels", "ious",
                          "proposal_offsets", "valid_mask"):
                    out["eval_kept_" + f] = _np(getattr(kept, f))
                out["eval_pc_ids"] = np.asarray(pc_ids)
        model.on_validation_epoch_end()
    for k, v in logged.items():
        out["eval_log/" + k] = np.asarray(v)
    tap.close()
    return out
