"""GAPartNet perception model (reference: gapartnet/network/model.py:27-1055) on the MI355X operators.

API surface kept from the reference so this class is a drop-in for ``train.py fit/test`` with ``gapartnet.yaml``:
constructor keyword arguments (model.py:28-55), sub-module names (=> state_dict keys), the forward_* /
loss_* methods, the Lightning hooks (training_step, validation_step, on_validation_epoch_end, test_step,
on_test_epoch_end, configure_optimizers) and every logged key.  Visualisation / pose rendering in
on_test_epoch_end (model.py:930-1049) is out of scope.

Pipeline of one step (model.py:466-659):
  collate -> backbone U-Net (71 sparse convs) -> voxel->point gather -> semantic head + offset head
  -> [epoch >= min(schedule)] dual-set clustering (ball query + CCL on xyz and xyz+offset) -> proposals >= 5 points
     -> per-proposal re-voxelisation into 28^3 grids
  -> [epoch >= schedule[0]] score U-Net -> per-proposal max-pool -> score head, IoU-derived targets
  -> [epoch >= schedule[1]] NPCS U-Net -> per-point NPCS, symmetry-aware loss
"""
import functools
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import backend
from .. import functional as GF
from ..epic_ops.iou import batch_instance_seg_iou
from ..epic_ops.reduce import segmented_maxpool
from ..lightning_lite import LightningModule
from ..misc.info import PART_ID2NAME, get_symmetry_matrix
from ..spconv import pytorch as spconv
from ..structure.instances import Instances
from ..structure.point_cloud import PointCloud, PointCloudBatch
from ..structure.segmentation import Segmentation
from .backbone import SparseUNet
from .grouping_utils import (apply_nms, cluster_proposals, compute_ap, compute_ap_multi, compute_npcs_loss, compute_npcs_loss_grouped, SymmetryTables,
                             filter_invalid_proposals, get_gt_scores, offsets_from_counts, segmented_voxelize)
from .losses import dice_loss, focal_loss, mean_iou, pixel_accuracy

_SPLITS = ["val", "test_intra", "test_inter"]


class _PendingKept:
    """the proposals a validation step keeps, before the one host read of its post-processing (GAPartNet.defer_validation_outputs):
    ``resolve()`` -> Instances (or the torch formulation's result after a table overflow inside the kernel)"""

    def __init__(self, model, proposals, handle):
        self.model, self.proposals, self.handle = model, proposals, handle

    def resolve(self):
        kept = GAPartNet._kept_from(self.proposals, self.handle.result())
        if kept is None:  # a neighbour table inside the kernel overflowed: the torch formulation on the live rows
            kept = self.model._post_process_kept_torch(self.proposals)
        self.model = self.proposals = self.handle = None
        return kept


class GAPartNet(LightningModule):
    def __init__(
        self,
        in_channels: int,
        num_part_classes: int,
        backbone_type: str = "SparseUNet",
        backbone_cfg: Dict = {},
        learning_rate: float = 1e-3,
        # semantic segmentation
        ignore_sem_label: int = -100,
        use_sem_focal_loss: bool = True,
        use_sem_dice_loss: bool = True,
        # instance segmentation
        instance_seg_cfg: Dict = {},
        # npcs segmentation
        symmetry_indices: List = [],
        # training
        training_schedule: List = [],
        # validation
        val_score_threshold: float = 0.09,
        val_min_num_points_per_proposal: int = 3,
        val_nms_iou_threshold: float = 0.3,
        val_ap_iou_threshold: float = 0.5,
        # testing
        visualize_cfg: Dict = {},
        debug: bool = True,
        ckpt: str = "",
        # not in the reference: voxel size used when scenes reach the model un-voxelised (on-device voxelisation)
        voxel_size: Sequence[float] = (0.01, 0.01, 0.01),
    ):
        super().__init__()
        self.save_hyperparameters()
        self.validation_step_outputs = []

        self.in_channels = in_channels
        self.num_part_classes = num_part_classes
        self.backbone_type = backbone_type
        self.backbone_cfg = backbone_cfg
        self.learning_rate = learning_rate
        self.ignore_sem_label = ignore_sem_label
        self.use_sem_focal_loss = use_sem_focal_loss
        self.use_sem_dice_loss = use_sem_dice_loss
        self.visualize_cfg = visualize_cfg
        self.start_scorenet, self.start_npcs = training_schedule
        self.start_clustering = min(self.start_scorenet, self.start_npcs)
        self.val_nms_iou_threshold = val_nms_iou_threshold
        self.val_ap_iou_threshold = val_ap_iou_threshold
        self.val_score_threshold = val_score_threshold
        self.val_min_num_points_per_proposal = val_min_num_points_per_proposal
        self.symmetry_indices = torch.as_tensor(symmetry_indices, dtype=torch.int64)
        self.voxel_size = [float(v) for v in voxel_size]
        self.revoxelize_jitter = None  # tests inject the two uniform 3-vectors of segmented_voxelize here
        self.record_npcs_preds = False  # True: keep proposals.npcs_preds / gt_npcs in training steps too (costs a host read)
        self.defer_validation_outputs = False  # True (set by an evaluation loop): validation_step's host read happens a step later
        self._want_npcs_preds = False  # this step keeps proposals.npcs_preds (test steps; training steps with record_npcs_preds)
        self.use_fused_proposals = True  # csrc/proposals.hip on the GPU; False = the torch formulation of the same stage
        # Training steps issue the proposal stage and everything behind it WITHOUT reading its sizes back (include/gpn.h section
        # DEV): buffers at their bounds, counts on the device, the previous step's counts (copied to pinned memory, taken over
        # without waiting) as the plan that sizes grids.  The first step of a run - no plan yet - and evaluation steps read the
        # counts as before.  The attribute set to False restores the blocking read for every step (tests/test_gpu_sync_free.py).
        self.sync_free_proposals = True
        self._prop_plan = None       # [Q, M, P, V, dropped, runs, coarse] of the latest step whose counts have arrived
        self._prop_pending = []      # (pinned int64 [8], event) of counts still on their way
        self._prop_hist = []         # the last few plans (floor of the next one: a step without proposals must not shrink the grids)
        self._prop_gate = None       # (device counts, index) deciding whether this step's proposal networks take an Adam step
        self._prop_pinned = []       # pinned buffers ready for reuse

        self.ball_query_radius = instance_seg_cfg["ball_query_radius"]
        self.max_num_points_per_query = instance_seg_cfg["max_num_points_per_query"]
        self.min_num_points_per_proposal = instance_seg_cfg["min_num_points_per_proposal"]
        self.max_num_points_per_query_shift = instance_seg_cfg["max_num_points_per_query_shift"]
        self.score_fullscale = instance_seg_cfg["score_fullscale"]
        self.score_scale = instance_seg_cfg["score_scale"]

        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        if self.backbone_type != "SparseUNet":
            raise NotImplementedError(
                f"backbone type {self.backbone_type!r}: only the SparseUNet path is on the accelerated hot path "
                "(the dense PointNet backbone of the reference is out of scope, SURVEY.md §2.1 #9)")
        channels = self.backbone_cfg["channels"]
        block_repeat = self.backbone_cfg["block_repeat"]
        width = channels[0]
        self.backbone = SparseUNet.build(in_channels, channels, block_repeat, norm_fn)
        self.sem_seg_head = nn.Linear(width, self.num_part_classes)
        self.offset_head = nn.Sequential(nn.Linear(width, width), norm_fn(width), nn.ReLU(inplace=True),
                                         nn.Linear(width, 3))
        self.score_unet = SparseUNet.build(width, channels[:2], block_repeat, norm_fn, without_stem=True)
        self.score_head = nn.Linear(width, self.num_part_classes - 1)
        self.npcs_unet = SparseUNet.build(width, channels[:2], block_repeat, norm_fn, without_stem=True)
        self.npcs_head = nn.Linear(width, 3 * (self.num_part_classes - 1))

        self.symmetry_matrix_1, self.symmetry_matrix_2, self.symmetry_matrix_3 = get_symmetry_matrix()
        self._prefetch_hook = None  # set by dataset.prefetch.DevicePrefetcher while it feeds this model

        if ckpt != "":
            print("Loading pretrained model from:", ckpt)
            state_dict = torch.load(ckpt, map_location="cpu")["state_dict"]
            missing_keys, unexpected_keys = self.load_state_dict(state_dict, strict=False)
            if len(missing_keys) > 0:
                print("missing_keys:", missing_keys)
            if len(unexpected_keys) > 0:
                print("unexpected_keys:", unexpected_keys)

    # ------------------------------------------------------------------------------------------ forward pieces
    def forward_backbone(self, pc_batch: PointCloudBatch) -> torch.Tensor:
        voxel_features = self.backbone(pc_batch.voxel_tensor)
        return GF.gather_rows(voxel_features.features, pc_batch.pc_voxel_id, getattr(pc_batch, "pc_voxel_csr", None))

    def forward_sem_seg(self, pc_feature: torch.Tensor) -> torch.Tensor:
        return GF.linear(pc_feature, self.sem_seg_head.weight, self.sem_seg_head.bias)

    def loss_sem_seg(self, sem_logits: torch.Tensor, sem_labels: torch.Tensor) -> torch.Tensor:
        if self.use_sem_focal_loss:
            loss = focal_loss(sem_logits, sem_labels, alpha=None, gamma=2.0, ignore_index=self.ignore_sem_label,
                              reduction="mean")
        else:
            loss = F.cross_entropy(sem_logits, sem_labels, weight=None, ignore_index=self.ignore_sem_label,
                                   reduction="mean")
        if self.use_sem_dice_loss:
            loss = loss + dice_loss(sem_logits[:, :, None, None], sem_labels[:, None, None])
        return loss

    def forward_offset(self, pc_feature: torch.Tensor) -> torch.Tensor:
        fc0, norm, act, fc1 = self.offset_head[0], self.offset_head[1], self.offset_head[2], self.offset_head[3]
        if not (isinstance(fc0, nn.Linear) and isinstance(norm, nn.BatchNorm1d) and isinstance(act, nn.ReLU)
                and isinstance(fc1, nn.Linear)):
            return self.offset_head(pc_feature)
        # Linear -> BatchNorm1d -> ReLU -> Linear on the same kernels as the backbone layers (GF.linear, GF.bn_act)
        hidden = GF.bn_act(GF.linear(pc_feature, fc0.weight, fc0.bias), norm, relu=True)
        return GF.linear(hidden, fc1.weight, fc1.bias)

    def loss_offset(self, offsets: torch.Tensor, gt_offsets: torch.Tensor, sem_labels: torch.Tensor,
                    instance_labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """L1 distance + negative cosine between predicted and true point->instance-centre offsets, on points of
        labelled part instances (model.py:204-226)."""
        on_part = (sem_labels > 0) & (instance_labels >= 0)
        # masked means written as sum / count: the reference's boolean-mask selection (x[on_part].mean()) costs a
        # device->host round trip per selection; the value is the same (NaN for an empty selection, as .mean() gives)
        count = on_part.sum()
        zero = offsets.new_zeros(())
        loss_dist = torch.where(on_part, (offsets - gt_offsets).abs().sum(dim=-1), zero).sum() / count
        gt_dir = gt_offsets / (torch.norm(gt_offsets, p=2, dim=-1)[:, None] + 1e-8)
        pred_dir = offsets / (torch.norm(offsets, p=2, dim=-1)[:, None] + 1e-8)
        loss_dir = torch.where(on_part, -(gt_dir * pred_dir).sum(-1), zero).sum() / count
        return loss_dist, loss_dir

    def proposal_clustering_and_revoxelize(self, pt_xyz: torch.Tensor, batch_indices: torch.Tensor,
                                           pt_features: torch.Tensor, sem_preds: torch.Tensor,
                                           offset_preds: torch.Tensor, instance_labels: Optional[torch.Tensor],
                                           batch_size: Optional[int] = None):
        """dual-set clustering and per-proposal re-voxelisation (model.py:228-346).
        -> (voxel_tensor, pc_voxel_id, proposals) or (None, None, None) when no proposal survives."""
        device = pt_xyz.device
        ops = backend.raw()
        if self.use_fused_proposals and batch_size is not None and pt_xyz.is_cuda and hasattr(ops, "proposals_build"):
            # the stage as ONE library call with ONE host read (csrc/proposals.hip); the torch formulation below is what
            # runs over other operator backends (the CPU oracle in tests) and what this call is tested against
            return self._proposals_fused(ops, pt_xyz, batch_indices, pt_features, sem_preds, offset_preds, instance_labels,
                                         batch_size)
        valid_mask = sem_preds > 0
        if instance_labels is not None:
            valid_mask = valid_mask & (instance_labels >= 0)

        # one compaction index for every per-point array (each boolean-mask selection would be its own host sync)
        valid_indices = torch.nonzero(valid_mask).squeeze(1)
        pt_xyz, batch_indices = pt_xyz[valid_indices], batch_indices[valid_indices]
        sem_preds, offset_preds = sem_preds[valid_indices].int(), offset_preds[valid_indices]
        if instance_labels is not None:
            instance_labels = instance_labels[valid_indices]

        if batch_size is not None:
            # CSR over all scenes (empty ones included: they own no points and no queries, so the clusters are the
            # same): batch_indices is sorted, so the offsets are a binary search - no host sync
            scene_compact = batch_indices.int()
            scene_offsets = torch.searchsorted(
                scene_compact, torch.arange(batch_size + 1, dtype=torch.int32, device=device)).to(torch.int32)
        else:
            # CSR over the scenes that still have points
            _, scene_compact, scene_counts = torch.unique_consecutive(batch_indices, return_inverse=True,
                                                                      return_counts=True)
            scene_compact = scene_compact.int()
            scene_offsets = offsets_from_counts(scene_counts)

        # set 1: clusters in xyz; set 2: clusters in xyz shifted by the predicted centre offsets
        labels_a, order_a = cluster_proposals(pt_xyz, scene_compact, scene_offsets, sem_preds,
                                              self.ball_query_radius, self.max_num_points_per_query)
        labels_b, order_b = cluster_proposals(pt_xyz + offset_preds, scene_compact, scene_offsets, sem_preds,
                                              self.ball_query_radius, self.max_num_points_per_query_shift)
        labels = torch.cat([labels_a, labels_b + labels_a.shape[0]], dim=0)
        sorted_indices = torch.cat([order_a, order_b], dim=0)

        # size of the run (cluster) every point belongs to, with the run count bounded by the point count instead of
        # read back from the device (the reference's unique_consecutive here is one more host sync)
        n_pts = labels.shape[0]
        run_start = torch.ones((n_pts,), dtype=torch.bool, device=device)
        run_start[1:] = labels[1:] != labels[:-1]
        proposal_indices = run_start.long().cumsum(0) - 1
        sizes = torch.zeros((n_pts,), dtype=torch.int64, device=device).scatter_add_(
            0, proposal_indices, torch.ones((n_pts,), dtype=torch.int64, device=device))
        keep_point = torch.nonzero((sizes >= self.min_num_points_per_proposal)[proposal_indices]).squeeze(1)
        sorted_indices = sorted_indices[keep_point]
        if sorted_indices.shape[0] == 0:
            return None, None, None

        batch_indices, pt_xyz = batch_indices[sorted_indices], pt_xyz[sorted_indices]
        # the only differentiable selection of the function, done once on the original rows: index_select back-propagates
        # with an index_add (a point is in at most two proposals, so the sum is order-independent) instead of the sort-based
        # index_put the advanced-indexing form uses
        pt_features = pt_features.index_select(0, valid_indices[sorted_indices])
        sem_preds = sem_preds[sorted_indices]
        if instance_labels is not None:
            instance_labels = instance_labels[sorted_indices]

        _, proposal_indices, sizes = torch.unique_consecutive(proposal_indices[keep_point], return_inverse=True,
                                                              return_counts=True)
        num_proposals = sizes.shape[0]
        proposal_offsets = offsets_from_counts(sizes)

        voxel_features, voxel_coords, pc_voxel_id, extras = segmented_voxelize(
            pt_xyz, pt_features, proposal_offsets, proposal_indices, sizes, self.score_fullscale, self.score_scale,
            jitter=self.revoxelize_jitter, with_extras=True)
        voxel_tensor = spconv.SparseConvTensor(voxel_features, voxel_coords.int(),
                                               spatial_shape=[self.score_fullscale] * 3, batch_size=num_proposals)
        voxel_tensor.point_csr = extras["csr"]  # points grouped by voxel: the transpose of the voxel->point gathers
        if extras["dropped"] != 0:
            raise RuntimeError("re-voxelisation dropped points: a proposal left its score_fullscale^3 grid "
                               "(the reference stops in pdb here, model.py:328-330)")

        proposals = Instances(valid_mask=valid_mask, valid_indices=valid_indices, sorted_indices=sorted_indices, pt_xyz=pt_xyz,
                              batch_indices=batch_indices, proposal_offsets=proposal_offsets,
                              proposal_indices=proposal_indices, num_points_per_proposal=sizes, sem_preds=sem_preds,
                              instance_labels=instance_labels)
        return voxel_tensor, pc_voxel_id, proposals

    def _proposals_fused(self, ops, pt_xyz, batch_indices, pt_features, sem_preds, offset_preds, instance_labels, batch_size):
        jitter = self.revoxelize_jitter
        rng_before = None
        if jitter is None:  # the reference's two torch.rand(3) draws, in its order, from the device generator
            # (the reference - and the unfused path - draw them inside segmented_voxelize, i.e. only when a proposal survives;
            # here they are needed before that is known, so the generator is put back if none does: same random stream)
            gen = torch.cuda.default_generators[pt_xyz.device.index if pt_xyz.device.index is not None else torch.cuda.current_device()] \
                if pt_xyz.is_cuda else torch.default_generator
            rng_before = (gen, gen.get_state())
            jitter = (torch.rand(3, dtype=torch.float32, device=pt_xyz.device),
                      torch.rand(3, dtype=torch.float32, device=pt_xyz.device))
        self._prop_gate = None
        # (training AND validation steps: round 5 - a validation step needs its sizes only once, after its post-processing)
        sync_free = (self.sync_free_proposals and not self._want_npcs_preds and self._proposal_unets_take_device_counts())
        if sync_free:
            self._take_over_proposal_counts()
            sync_free = self._prop_plan is not None
        built = ops.proposals_build(pt_xyz, offset_preds, sem_preds, instance_labels, batch_indices, batch_size,
                                    self.ball_query_radius, self.max_num_points_per_query,
                                    self.max_num_points_per_query_shift, self.min_num_points_per_proposal,
                                    float(self.score_fullscale), float(self.score_scale), jitter, read_counts=not sync_free)
        if sync_free:
            return self._proposals_without_a_read(pt_features, built)
        if built is None:
            if self.sync_free_proposals:
                self._set_proposal_plan([0] * 7)  # (a step without proposals is a plan too: the next one need not wait for its counts)
            if rng_before is not None:
                rng_before[0].set_state(rng_before[1])
            return None, None, None
        if self.sync_free_proposals:
            Q, M, P, V, dropped, coarse = built["counts_host"]
            self._set_proposal_plan([Q, M, P, V, dropped, 0, coarse])
        if built["dropped"] != 0:
            raise RuntimeError("re-voxelisation dropped points: a proposal left its score_fullscale^3 grid "
                               "(the reference stops in pdb here, model.py:328-330)")
        voxel_features = GF.proposal_voxel_mean(pt_features, built)
        voxel_tensor = spconv.SparseConvTensor(voxel_features, built["voxel_coords"],
                                               spatial_shape=[self.score_fullscale] * 3, batch_size=built["P"])
        voxel_tensor.level_counts = [built["coarse"]]  # rows of the grid's stride-2 level: no read when its rulebook is built
        voxel_tensor.point_csr = (built["point_order"], built["voxel_point_start"])
        proposals = Instances(valid_mask=built["valid_mask"], valid_indices=built["valid_indices"],
                              sorted_indices=built["sorted_indices"], point_indices=built["point_indices"],
                              pt_xyz=built["pt_xyz"], batch_indices=built["batch_indices"],
                              proposal_offsets=built["proposal_offsets"], proposal_indices=built["proposal_indices"],
                              num_points_per_proposal=built["sizes"], sem_preds=built["sem_preds"],
                              instance_labels=built["instance_labels"])
        proposals.member_slot = built["member_slot"]  # (row of every point in the other cluster set: gpn_proposals_postprocess)
        return voxel_tensor, built["pc_voxel_id"], proposals

    def _proposal_unets_take_device_counts(self) -> bool:
        """the device-counted form of the proposal stage hands ScoreNet / NPCS-Net tensors at their 2 N bound: only the native
        executor reads the live row count, so the form is used only when both networks will run there (HIP backend, switch on,
        a program exists for the module tree, BatchNorm training flags uniform)"""
        if backend.raw().name != "hip":
            return False
        from . import net_exec
        for net in (self.score_unet, self.npcs_unet):
            if not getattr(net, "use_native_executor", False) or not net_exec.runs_natively(net):
                return False
        return True

    def _take_over_proposal_counts(self, wait: bool = False):
        """the counts of the step BEFORE the previous one become the plan.  Always that step's (round 5; before: whichever copy
        had arrived, i.e. host timing): the plan picks kernel variants and slice counts - BatchNorm small-N against two-pass,
        the weight-gradient slices whose partial sums are added in slice order - so a plan that depended on when the host looked
        made two runs of one seed differ in the last bits.  The copy of step i - 2 was queued before step i - 1's backbone:
        by the time step i asks, it has long arrived (synchronize() returns at once; `wait` drains everything, e.g. at the end
        of an epoch)."""
        pend = self._prop_pending
        while pend and (wait or len(pend) > 1):
            host, ev = pend.pop(0)
            ev.synchronize()
            counts = host.tolist()
            self._prop_pinned.append(host)
            if counts[4] != 0:
                raise RuntimeError("re-voxelisation dropped points in an earlier training step: a proposal left its "
                                   "score_fullscale^3 grid (the reference stops in pdb there, model.py:328-330)")
            self._set_proposal_plan(counts[:7])

    def _set_proposal_plan(self, counts):
        """plan = the given counts, but no entry below half the largest of the last four plans: after a step without (or
        with few) proposals the next one would otherwise launch single-workgroup grids and pick the small-matrix kernel
        variants for 10^4 - 10^5 live rows (correct, and milliseconds slower)"""
        hist = self._prop_hist
        hist.append(list(counts))
        del hist[:-4]
        floor = [max(h[k] for h in hist) // 2 for k in range(len(counts))]
        self._prop_plan = [max(c, f) for c, f in zip(counts, floor)]

    def _proposals_without_a_read(self, pt_features, built):
        """the outputs of gpn_proposals_build as they are - every tensor at its bound, the counts as device counters - for a
        training step that never learns the sizes on the host (include/gpn.h section DEV)"""
        from ..hip_ops import DevCount
        counts = built["counts"]
        host = self._prop_pinned.pop() if self._prop_pinned else torch.empty((8,), dtype=torch.int64).pin_memory()
        host.copy_(counts, non_blocking=True)  # arrives some time during this step; read by a later one
        ev = torch.cuda.Event()
        ev.record()
        self._prop_pending.append((host, ev))
        # ScoreNet / NPCS-Net and their heads take an optimizer step only if this step had a proposal (optim.FusedAdam.set_gate:
        # the reference does not run them otherwise); with several ranks the all-reduced gradient decides for all of them alike
        if self.training and not (torch.distributed.is_available() and torch.distributed.is_initialized()
                                  and torch.distributed.get_world_size() > 1):
            self._prop_gate = (counts, 2)
        plan = self._prop_plan
        dev = dict(M=DevCount(built["M_dev"], max(plan[1], 1)), P=DevCount(built["P_dev"], max(plan[2], 1)),
                   V=DevCount(built["V_dev"], max(plan[3], 1)))
        voxel_features = GF.proposal_voxel_mean(pt_features, built, rows=dev["V"])
        voxel_tensor = spconv.SparseConvTensor(voxel_features, built["voxel_coords"],
                                               spatial_shape=[self.score_fullscale] * 3, batch_size=built["P"])
        voxel_tensor.rows_dev, voxel_tensor.batch_dev = dev["V"], dev["P"]
        voxel_tensor.level_plans = [max(plan[6], 1)]
        voxel_tensor.point_csr = (built["point_order"], built["voxel_point_start"])
        proposals = Instances(valid_mask=built["valid_mask"], valid_indices=built["valid_indices"],
                              sorted_indices=built["sorted_indices"], point_indices=built["point_indices"],
                              pt_xyz=built["pt_xyz"], batch_indices=built["batch_indices"],
                              proposal_offsets=built["proposal_offsets"], proposal_indices=built["proposal_indices"],
                              num_points_per_proposal=built["sizes"], sem_preds=built["sem_preds"],
                              instance_labels=built["instance_labels"], dev_counts=dev)
        proposals.member_slot = built["member_slot"]
        return voxel_tensor, built["pc_voxel_id"], proposals

    # ScoreNet and NPCS-Net read the same proposal grid and have the same structure: with both switched on, their U-Nets run
    # as PAIRED passes of the native executor (network/net_exec.run_pair: layer i of both networks in one launch per kernel,
    # forward and backward) - same values as one after the other, half the launches of a part of the step where the GPU
    # waits for the host to issue them.  The attribute set to False runs them one after the other
    # (tests/test_gpu_model.py::test_paired_passes_equal_one_network_after_the_other).
    pair_proposal_unets = True

    def forward_proposal_unets(self, voxel_tensor: spconv.SparseConvTensor):
        """(score_unet(x), npcs_unet(x)) in paired passes, or None where that form does not apply"""
        if not self.pair_proposal_unets or backend.raw().name != "hip":
            return None
        if not (getattr(self.score_unet, "use_native_executor", False) and getattr(self.npcs_unet, "use_native_executor", False)):
            return None
        from . import net_exec
        return net_exec.run_pair(self.score_unet, self.npcs_unet, voxel_tensor)

    def forward_proposal_score(self, voxel_tensor: spconv.SparseConvTensor, pc_voxel_id: torch.Tensor,
                               proposals: Instances, feats: Optional[spconv.SparseConvTensor] = None) -> torch.Tensor:
        offsets = proposals.proposal_offsets
        if feats is None:
            feats = self.score_unet(voxel_tensor)
        dev = proposals.dev_counts
        if dev is not None:  # sizes on the device (no host read this step): same operators, extents as device counters
            feats = GF.gather_rows(feats.features, pc_voxel_id, voxel_tensor.point_csr, rows=dev["M"], table_rows=dev["V"])
            pooled, _ = GF.segmented_maxpool(feats, offsets[:-1], offsets[1:], rows=dev["P"], m_rows=dev["M"])
            return GF.linear(pooled, self.score_head.weight, self.score_head.bias, rows=dev["P"])
        feats = GF.gather_rows(feats.features, pc_voxel_id, getattr(voxel_tensor, "point_csr", None))
        pooled, _ = segmented_maxpool(feats, offsets[:-1], offsets[1:])
        return GF.linear(pooled, self.score_head.weight, self.score_head.bias)

    def loss_proposal_score(self, score_logits: torch.Tensor, proposals: Instances,
                            num_points_per_instance: torch.Tensor) -> torch.Tensor:
        ious = batch_instance_seg_iou(proposals.proposal_offsets, proposals.instance_labels, proposals.batch_indices,
                                      num_points_per_instance)
        proposals.ious = ious
        proposals.num_points_per_instance = num_points_per_instance
        gt_scores = get_gt_scores(ious.max(-1)[0], 0.75, 0.25)
        return F.binary_cross_entropy_with_logits(score_logits, gt_scores)

    def forward_proposal_npcs(self, voxel_tensor: spconv.SparseConvTensor, pc_voxel_id: torch.Tensor,
                              feats: Optional[spconv.SparseConvTensor] = None, dev: Optional[dict] = None) -> torch.Tensor:
        if feats is None:
            feats = self.npcs_unet(voxel_tensor)
        if dev is not None:
            logits = GF.linear(feats.features, self.npcs_head.weight, self.npcs_head.bias, rows=dev["V"])
            return GF.gather_rows(logits, pc_voxel_id, voxel_tensor.point_csr, rows=dev["M"], table_rows=dev["V"])
        logits = GF.linear(feats.features, self.npcs_head.weight, self.npcs_head.bias)
        return GF.gather_rows(logits, pc_voxel_id, getattr(voxel_tensor, "point_csr", None))

    def loss_proposal_npcs(self, npcs_logits: torch.Tensor, gt_npcs: torch.Tensor, proposals: Instances) -> torch.Tensor:
        """symmetry-aware NPCS loss on points whose predicted part class is right and that carry a non-zero NPCS
        target (model.py:398-462); the per-class 3-vector is selected by the predicted class."""
        sem_preds, sem_labels = proposals.sem_preds, proposals.sem_labels
        # (with the sizes on the device the rows past the live count are undefined: no host-side mask then)
        valid = (sem_preds == sem_labels) & (gt_npcs != 0).any(dim=-1) if proposals.dev_counts is None else None
        ops = backend.raw()
        fused = npcs_logits.is_cuda and hasattr(ops, "npcs_loss_fwd") and npcs_logits.shape[0] > 0
        dev_counts = proposals.dev_counts
        proposals.npcs_valid_mask = valid
        valid_idx = None
        if fused and not self._want_npcs_preds:
            # nobody reads the selected predictions in a training or validation step: skip the compaction (a host read) altogether
            proposals.npcs_preds, proposals.gt_npcs = None, None
        else:
            valid_idx = torch.nonzero(valid).squeeze(1)
            per_class = npcs_logits[valid_idx].reshape(valid_idx.shape[0], npcs_logits.shape[1] // 3, 3)
            cls = sem_preds[valid_idx].long()
            proposals.npcs_preds = per_class.gather(1, (cls - 1)[:, None, None].expand(-1, 1, 3)).squeeze(1).detach()
            proposals.gt_npcs = gt_npcs[valid_idx]

        dev = sem_preds.device
        self.symmetry_indices = self.symmetry_indices.to(dev)
        self.symmetry_matrix_1 = self.symmetry_matrix_1.to(dev)
        self.symmetry_matrix_2 = self.symmetry_matrix_2.to(dev)
        self.symmetry_matrix_3 = self.symmetry_matrix_3.to(dev)
        if fused:
            # the whole loss in two launches (csrc/losses.hip); the torch formulation below is what runs over other operator
            # backends and what the kernel is tested against
            sym = getattr(self, "_npcs_sym", None)
            if sym is None or sym["mats"].device != dev:
                tables = (self.symmetry_matrix_1, self.symmetry_matrix_2, self.symmetry_matrix_3)
                first, count, group, mats = [], [], [], []
                for g, table in enumerate(tables):
                    for t in range(table.shape[0]):
                        first.append(sum(m.shape[0] for m in mats))
                        count.append(int(table.shape[1]))
                        group.append(g)
                        mats.append(table[t].reshape(-1, 3, 3))
                sym = dict(sym_of_class=self.symmetry_indices.to(torch.int64).contiguous(),
                           mats=torch.cat(mats).to(device=dev, dtype=torch.float32).contiguous(), first=first, count=count,
                           group=group)
                self._npcs_sym = sym
            if dev_counts is not None:
                return GF.npcs_loss(npcs_logits, gt_npcs, sem_preds, sem_labels, proposals.proposal_offsets,
                                    proposals.proposal_indices, sym, p_rows=dev_counts["P"], m_rows=dev_counts["M"])
            return GF.npcs_loss(npcs_logits, gt_npcs, sem_preds, sem_labels, proposals.proposal_offsets,
                                proposals.proposal_indices, sym)

        npcs_logits, gt_npcs = npcs_logits[valid_idx], gt_npcs[valid_idx]
        sem_preds = sem_preds[valid_idx].long()
        proposal_indices = proposals.proposal_indices[valid_idx]
        per_class = npcs_logits.reshape(npcs_logits.shape[0], npcs_logits.shape[1] // 3, 3)  # valid for 0 rows too
        npcs_preds = per_class.gather(1, (sem_preds - 1)[:, None, None].expand(-1, 1, 3)).squeeze(1)
        sym = self.symmetry_indices[sem_preds]

        # the reference evaluates compute_npcs_loss once per symmetry group (sym < 3, == 3, == 4) on boolean-mask
        # selections (twelve host syncs); compute_npcs_loss_grouped gives the same sum in one pass with none
        tables = getattr(self, "_symmetry_tables", None)
        if tables is None or tables.flat.device != dev:
            tables = SymmetryTables((self.symmetry_matrix_1, self.symmetry_matrix_2, self.symmetry_matrix_3), dev)
            self._symmetry_tables = tables
        num_proposals = proposals.proposal_offsets.shape[0] - 1
        return compute_npcs_loss_grouped(npcs_preds, gt_npcs, proposal_indices, sym, tables, num_proposals)

    @staticmethod
    def _proposal_rows(proposals: Instances) -> torch.Tensor:
        """row of every proposal point in the batch's point arrays"""
        if proposals.point_indices is not None:
            return proposals.point_indices
        return proposals.valid_indices[proposals.sorted_indices]

    # ------------------------------------------------------------------------------------------ one step
    def _collate(self, point_clouds: Union[Sequence[PointCloud], PointCloudBatch]) -> PointCloudBatch:
        if isinstance(point_clouds, PointCloudBatch):
            return point_clouds
        levels = 0  # the backbone's coarse levels: their row counts ride on the voxelisation's single host read
        if getattr(self.backbone, "use_native_executor", False) and point_clouds and point_clouds[0].points.is_cuda:
            from . import net_exec
            prog = net_exec.program_for(self.backbone)
            levels = prog.n_levels - 1 if prog is not None and prog.n_levels > 2 else 0
        return PointCloud.collate(point_clouds, voxel_size=self.voxel_size, pyramid_levels=levels)

    def _training_or_validation_step(self, point_clouds, batch_idx: int, running_mode: str, want_npcs_preds: Optional[bool] = None):
        # proposals.npcs_preds (the NPCS predictions compacted to the valid points: a host read) are kept by test steps and on
        # request (record_npcs_preds); the reference computes them in every step and reads them only in test_step
        self._want_npcs_preds = bool(self.record_npcs_preds if want_npcs_preds is None else want_npcs_preds)
        data_batch = self._collate(point_clouds)
        batch_size = data_batch.batch_size
        points = data_batch.points
        sem_labels = data_batch.sem_labels
        instance_regions = data_batch.instance_regions
        instance_labels = data_batch.instance_labels
        num_points_per_instance = data_batch.num_points_per_instance
        gt_npcs = data_batch.gt_npcs
        pt_xyz = points[:, :3]

        pc_feature = self.forward_backbone(pc_batch=data_batch)

        # semantic segmentation + centre offsets
        sem_logits = self.forward_sem_seg(pc_feature)
        offsets_preds = self.forward_offset(pc_feature)
        if instance_regions is None:
            raise RuntimeError("batch carries no instance_regions (the reference stops in pdb here, model.py:525)")
        gt_offsets = instance_regions[:, :3] - pt_xyz
        sem_preds = all_accu = pixel_accu = None
        if (sem_labels is not None and self.use_sem_focal_loss and self.use_sem_dice_loss
                and GF.point_losses_available(sem_logits)):
            # loss_sem_seg (focal + dice) and loss_offset (distance, direction) of all points in one fused pass, which also
            # emits the predicted classes and the two accuracies (argmax + ten small torch launches otherwise)
            fused, sem_preds, accu = GF.point_losses_with_metrics(sem_logits, offsets_preds, sem_labels, gt_offsets,
                                                                  instance_labels, self.ignore_sem_label)
            # (unbind, not four selects: its backward is one stack instead of four zero-filled [4] tensors and their sums)
            focal, dice, loss_offset_dist, loss_offset_dir = fused.unbind(0)
            loss_sem_seg = focal + dice
            all_accu, pixel_accu = accu.unbind(0)
        else:
            sem_preds = torch.argmax(sem_logits.detach(), dim=-1)
            loss_sem_seg = self.loss_sem_seg(sem_logits, sem_labels) if sem_labels is not None else 0.0
            loss_offset_dist, loss_offset_dir = self.loss_offset(offsets_preds, gt_offsets, sem_labels, instance_labels)
        if all_accu is None:
            all_accu = (sem_preds == sem_labels).sum().float() / sem_labels.shape[0]
            if sem_labels is not None:
                on_part = sem_labels > 0  # pixel_accuracy(sem_preds[on_part], sem_labels[on_part]) without the host syncs
                pixel_accu = ((sem_preds == sem_labels) & on_part).sum() / on_part.sum()
            else:
                pixel_accu = 0.0
        sem_seg = Segmentation(batch_size=batch_size, sem_preds=sem_preds, sem_labels=sem_labels, all_accu=all_accu,
                               pixel_accu=pixel_accu)

        # the backbone and the point heads are queued; what follows starts with host reads of device results.  A device
        # prefetcher (dataset/prefetch.py) uses this moment to prepare the next batch on its own stream
        hook = self._prefetch_hook
        if hook is not None:
            hook()

        # proposals
        voxel_tensor = pc_voxel_id = proposals = None
        if self.current_epoch >= self.start_clustering:
            voxel_tensor, pc_voxel_id, proposals = self.proposal_clustering_and_revoxelize(
                pt_xyz=pt_xyz, batch_indices=data_batch.batch_indices, pt_features=pc_feature, sem_preds=sem_preds,
                offset_preds=offsets_preds, instance_labels=instance_labels, batch_size=batch_size)
            if proposals is not None:
                if proposals.dev_counts is not None:  # the reference's sem_labels[rows] / gt_npcs[rows] for a device-counted row count
                    want_npcs = gt_npcs is not None and self.current_epoch >= self.start_npcs
                    proposals.sem_labels, proposals.gt_npcs = backend.raw().proposals_targets(
                        sem_labels, gt_npcs if want_npcs else None, proposals.point_indices, proposals.dev_counts["M"])
                elif sem_labels is not None:
                    proposals.sem_labels = sem_labels[self._proposal_rows(proposals)]
                proposals.instance_sem_labels = data_batch.instance_sem_labels

        pair = None
        if (self.current_epoch >= self.start_scorenet and self.current_epoch >= self.start_npcs and voxel_tensor is not None
                and proposals is not None):
            pair = self.forward_proposal_unets(voxel_tensor)
        score_feats, npcs_feats = pair if pair is not None else (None, None)

        loss_prop_score = 0.0
        if self.current_epoch >= self.start_scorenet and voxel_tensor is not None and proposals is not None:
            score_logits = self.forward_proposal_score(voxel_tensor, pc_voxel_id, proposals, score_feats)
            cls_source = proposals.sem_labels if proposals.sem_labels is not None else proposals.sem_preds
            if num_points_per_instance is None:
                raise RuntimeError("batch carries no num_points_per_instance (reference: pdb, model.py:567)")
            if proposals.dev_counts is not None:
                P_dev = proposals.dev_counts["P"]
                with torch.no_grad():
                    ious = backend.raw().instance_iou(proposals.proposal_offsets, proposals.instance_labels, proposals.batch_indices,
                                                      num_points_per_instance, rows=P_dev)
                proposals.ious = ious
                proposals.num_points_per_instance = num_points_per_instance
                loss_prop_score, proposals.score_preds = GF.score_loss(score_logits, cls_source, proposals.proposal_offsets, ious,
                                                                       0.75, 0.25, rows=P_dev)
            elif GF.score_loss_available(score_logits):
                # class selection, soft IoU targets, BCE and the sigmoid scores in one launch (csrc/losses.hip); the torch
                # formulation below is what runs over other operator backends and what the kernel is tested against
                ious = batch_instance_seg_iou(proposals.proposal_offsets, proposals.instance_labels, proposals.batch_indices,
                                              num_points_per_instance)
                proposals.ious = ious
                proposals.num_points_per_instance = num_points_per_instance
                loss_prop_score, proposals.score_preds = GF.score_loss(score_logits, cls_source, proposals.proposal_offsets, ious,
                                                                       0.75, 0.25)
            else:
                first_point = proposals.proposal_offsets[:-1].long()
                proposal_cls = cls_source[first_point].long()
                score_logits = score_logits.gather(1, proposal_cls[:, None] - 1).squeeze(1)
                proposals.score_preds = score_logits.detach().sigmoid()
                loss_prop_score = self.loss_proposal_score(score_logits, proposals, num_points_per_instance)

        loss_prop_npcs = 0.0
        if self.current_epoch >= self.start_npcs and voxel_tensor is not None:
            npcs_logits = self.forward_proposal_npcs(voxel_tensor, pc_voxel_id, npcs_feats, proposals.dev_counts)
            if gt_npcs is not None:
                gt_npcs = proposals.gt_npcs if proposals.dev_counts is not None else gt_npcs[self._proposal_rows(proposals)]
                loss_prop_npcs = self.loss_proposal_npcs(npcs_logits, gt_npcs, proposals)

        loss = loss_sem_seg + loss_offset_dist + loss_offset_dir + loss_prop_score + loss_prop_npcs

        prefix = running_mode
        for key, value in ((f"{prefix}_loss/total_loss", loss), (f"{prefix}_loss/loss_sem_seg", loss_sem_seg),
                           (f"{prefix}_loss/loss_offset_dist", loss_offset_dist),
                           (f"{prefix}_loss/loss_offset_dir", loss_offset_dir),
                           (f"{prefix}_loss/loss_prop_score", loss_prop_score),
                           (f"{prefix}_loss/loss_prop_npcs", loss_prop_npcs), (f"{prefix}/all_accu", all_accu * 100),
                           (f"{prefix}/pixel_accu", pixel_accu * 100)):
            self.log(key, value, batch_size=batch_size, on_epoch=True, prog_bar=False, logger=True, sync_dist=True)
        return data_batch.pc_ids, sem_seg, proposals, loss

    # ------------------------------------------------------------------------------------------ Lightning hooks
    def training_step(self, point_clouds, batch_idx: int):
        return self._training_or_validation_step(point_clouds, batch_idx, "train")[3]

    def _post_process(self, proposals: Instances) -> Instances:
        proposals = filter_invalid_proposals(proposals, score_threshold=self.val_score_threshold,
                                             min_num_points_per_proposal=self.val_min_num_points_per_proposal)
        # (a point is in at most one cluster of each of the two sets: model.py:256-283)
        proposals = apply_nms(proposals, self.val_nms_iou_threshold, max_sets=2)
        proposals.pt_sem_classes = proposals.sem_preds[proposals.proposal_offsets[:-1].long()]
        return proposals

    @staticmethod
    def _sliced_to_live_sizes(proposals: Instances) -> Instances:
        """device-counted proposals (every tensor at its bound, the live counts on the device) as exactly-sized ones: two host
        reads - only the fallback of the fused post-processing takes this path"""
        dev = proposals.dev_counts
        if dev is None:
            return proposals
        m_bound, p_bound = int(proposals.point_indices.shape[0]), int(proposals.score_preds.shape[0])
        M, P = int(dev["M"].t.item()), int(dev["P"].t.item())

        def pts(t):
            return t[:M] if (t is not None and t.shape[0] == m_bound) else t

        def props(t):
            return t[:P] if (t is not None and t.shape[0] == p_bound) else t

        return Instances(
            valid_mask=proposals.valid_mask, valid_indices=proposals.valid_indices, sorted_indices=pts(proposals.sorted_indices),
            point_indices=pts(proposals.point_indices), pt_xyz=pts(proposals.pt_xyz), batch_indices=pts(proposals.batch_indices),
            proposal_offsets=proposals.proposal_offsets[:P + 1], proposal_indices=pts(proposals.proposal_indices),
            num_points_per_proposal=props(proposals.num_points_per_proposal), sem_preds=pts(proposals.sem_preds),
            score_preds=props(proposals.score_preds), npcs_preds=pts(proposals.npcs_preds), sem_labels=pts(proposals.sem_labels),
            instance_labels=pts(proposals.instance_labels), instance_sem_labels=proposals.instance_sem_labels,
            num_points_per_instance=proposals.num_points_per_instance, gt_npcs=pts(proposals.gt_npcs),
            npcs_valid_mask=pts(proposals.npcs_valid_mask), ious=props(proposals.ious))

    def _post_process_kept_torch(self, proposals: Instances) -> Instances:
        """what validation_step keeps of a step's proposals through the torch formulation (``_post_process``): the path of the
        oracle backend and of the unfused proposal stage, and the FALLBACK of the fused call when a proposal shares points with
        more proposals than its kernel tables hold (64 distinct / 32 above the IoU threshold) - also on a device-counted step,
        whose tensors are cut to their live sizes first (round 5 raised there: an aborted validation instead of a slower one)"""
        p = self._post_process(self._sliced_to_live_sizes(proposals))
        return Instances(score_preds=p.score_preds, pt_sem_classes=p.pt_sem_classes, batch_indices=p.batch_indices,
                         instance_sem_labels=p.instance_sem_labels, ious=p.ious, proposal_offsets=p.proposal_offsets,
                         valid_mask=p.valid_mask)

    def _post_process_kept(self, proposals: Instances, defer: bool = False):
        """what validation_step keeps of ``_post_process(proposals)`` - score filter, NMS, re-indexed fields - through ONE library
        call (gpn_proposals_postprocess, csrc/postprocess.hip: flags, one sort, sparse intersections through the proposal
        stage's member_slot, NMS in rounds, one compaction) and ONE host read (the two counts), instead of ~300 torch launches
        and ~25 reads.  Works on exactly-sized and on device-counted proposals.  None: not applicable (no member_slot - the
        unfused proposal path - or a table overflow inside the kernel): the caller runs the torch formulation.
        ``defer``: -> a ``_PendingKept`` whose ``resolve()`` does the read (and the re-indexing) later."""
        slot = getattr(proposals, "member_slot", None)
        if slot is None or proposals.score_preds is None or proposals.point_indices is None:
            return None
        dev = proposals.dev_counts
        out = backend.raw().proposals_postprocess(
            proposals.score_preds, proposals.num_points_per_proposal, proposals.proposal_offsets, proposals.point_indices,
            proposals.proposal_indices, slot, self.val_score_threshold, self.val_min_num_points_per_proposal,
            self.val_nms_iou_threshold, rows=dev["P"] if dev is not None else None, defer=defer)
        if defer:
            return _PendingKept(self, proposals, out)
        return self._kept_from(proposals, out)

    @staticmethod
    def _kept_from(proposals: Instances, out) -> Optional[Instances]:
        if out is None:
            return None
        ids, new_offsets, src_row = out
        first = proposals.proposal_offsets.index_select(0, ids).long()
        return Instances(score_preds=proposals.score_preds.index_select(0, ids),
                         pt_sem_classes=proposals.sem_preds.index_select(0, first),
                         batch_indices=proposals.batch_indices.index_select(0, src_row),
                         instance_sem_labels=proposals.instance_sem_labels, ious=proposals.ious.index_select(0, ids),
                         proposal_offsets=new_offsets, valid_mask=proposals.valid_mask)

    def _stash(self, dataloader_idx: int, item) -> None:
        self._resolve_pending_outputs()  # (the previous step's deferred read: its kernels finished while this step was queued)
        while dataloader_idx > len(self.validation_step_outputs) - 1:
            self.validation_step_outputs.append([])
        self.validation_step_outputs[dataloader_idx].append(item)

    def _resolve_pending_outputs(self) -> None:
        for outputs in self.validation_step_outputs:
            if outputs and isinstance(outputs[-1][2], _PendingKept):
                pc_ids, sem_seg, pending = outputs[-1]
                outputs[-1] = (pc_ids, sem_seg, pending.resolve())

    def validation_step(self, point_clouds, batch_idx: int, dataloader_idx: int = 0):
        # validation never reads proposals.npcs_preds: unless asked to keep them (record_npcs_preds) the step runs without a host
        # read of its sizes
        fast = backend.raw().name == "hip" and not self.record_npcs_preds
        pc_ids, sem_seg, proposals, _ = self._training_or_validation_step(point_clouds, batch_idx, _SPLITS[dataloader_idx],
                                                                          want_npcs_preds=not fast)
        kept = None
        if self.current_epoch >= self.start_scorenet and proposals is not None:
            # (an evaluation LOOP - gapartnet_amd.trainer - lets the step's one host read wait until the next step has been
            # queued: defer_validation_outputs; a direct call gets its proposals back at once, as the reference's does)
            kept = self._post_process_kept(proposals, defer=self.defer_validation_outputs) if fast else None
            if kept is None:
                kept = self._post_process_kept_torch(proposals)
        self._stash(dataloader_idx, (pc_ids, sem_seg, kept))
        return pc_ids, sem_seg, kept

    def test_step(self, point_clouds, batch_idx: int, dataloader_idx: int = 0):
        pc_ids, sem_seg, proposals, _ = self._training_or_validation_step(
            point_clouds, batch_idx, ["val", "intra", "inter"][dataloader_idx], want_npcs_preds=True)
        kept = None
        if proposals is not None and proposals.score_preds is not None:  # the reference dereferences None here (model.py:825)
            p = self._post_process(proposals)
            kept = Instances(pt_xyz=p.pt_xyz, score_preds=p.score_preds, pt_sem_classes=p.pt_sem_classes,
                             batch_indices=p.batch_indices, instance_sem_labels=p.instance_sem_labels, ious=p.ious,
                             proposal_offsets=p.proposal_offsets, proposal_indices=p.proposal_indices,
                             valid_mask=p.valid_mask, num_points_per_proposal=p.num_points_per_proposal,
                             num_points_per_instance=p.num_points_per_instance, sorted_indices=p.sorted_indices,
                             npcs_preds=p.npcs_preds, npcs_valid_mask=p.npcs_valid_mask)
        self._stash(dataloader_idx, (pc_ids, sem_seg, kept))
        return pc_ids, sem_seg, kept

    def _epoch_end_metrics(self) -> None:
        """semantic accuracy / mIoU and instance AP@50 / mAP(0.50:0.05:0.95) per split, plus the monitor_metrics means of
        the two test splits (model.py:694-805, 859-1046)."""
        self._resolve_pending_outputs()
        all_accus, pixel_accus, mious, mean_ap50, mAPs = [], [], [], [], []
        data_size = 0
        for split, outputs in zip(_SPLITS, self.validation_step_outputs):
            if len(outputs) == 0:
                continue
            data_size = sum(x[1].batch_size for x in outputs)
            all_accu = sum(x[1].all_accu for x in outputs) / len(outputs)
            pixel_accu = sum(x[1].pixel_accu for x in outputs) / len(outputs)
            sem_preds = torch.cat([x[1].sem_preds for x in outputs], dim=0)
            sem_labels = torch.cat([x[1].sem_labels for x in outputs], dim=0)
            miou = mean_iou(sem_preds, sem_labels, num_classes=self.num_part_classes)
            proposals = [x[2] for x in outputs if x[2] is not None]
            with_instances = self.current_epoch >= self.start_scorenet and len(proposals) > 0

            thresholds = [0.5 + 0.05 * i for i in range(10)]
            if with_instances:
                # matching + AP integration of all ten thresholds on the proposals' device, ONE host read
                # (the reference: ten Python walks over the proposals with .item() round trips, grouping_utils.py:378-404)
                aps = compute_ap_multi(proposals, self.num_part_classes, thresholds).cpu().numpy().astype(np.float64)
                ap50 = aps[0]
                mAP = float(aps.mean())
            else:
                ap50, mAP = 0, 0.0

            log = functools.partial(self.log, batch_size=data_size, on_epoch=True, logger=True, sync_dist=True)
            if with_instances:
                for class_idx in range(1, self.num_part_classes):
                    log(f"{split}/AP@50_{PART_ID2NAME[class_idx]}", float(np.mean(ap50[class_idx - 1])) * 100, prog_bar=False)
            log(f"{split}/AP@50", float(np.mean(ap50)) * 100, prog_bar=True)
            log(f"{split}/mAP", mAP * 100, prog_bar=True)
            log(f"{split}/all_accu", all_accu * 100.0, prog_bar=False)
            log(f"{split}/pixel_accu", pixel_accu * 100.0, prog_bar=False)
            log(f"{split}/miou", miou * 100.0, prog_bar=True)
            all_accus.append(all_accu); pixel_accus.append(pixel_accu); mious.append(miou)
            mean_ap50.append(float(np.mean(ap50))); mAPs.append(mAP)

        if len(all_accus) == 3:  # the monitor is the mean of test_intra / test_inter and needs all three loaders
            log = functools.partial(self.log, batch_size=data_size, on_epoch=True, logger=True, sync_dist=True)
            log("monitor_metrics/mean_all_accu", (all_accus[1] + all_accus[2]) / 2 * 100.0, prog_bar=False)
            log("monitor_metrics/mean_pixel_accu", (pixel_accus[1] + pixel_accus[2]) / 2 * 100.0, prog_bar=False)
            log("monitor_metrics/mean_imou", (mious[1] + mious[2]) / 2 * 100.0, prog_bar=True)
            log("monitor_metrics/mean_AP@50", (mean_ap50[1] + mean_ap50[2]) / 2 * 100.0, prog_bar=True)
            log("monitor_metrics/mean_mAP", (mAPs[1] + mAPs[2]) / 2 * 100.0, prog_bar=True)
        self.validation_step_outputs.clear()

    def on_validation_epoch_end(self):
        self._epoch_end_metrics()

    def on_test_epoch_end(self):
        if self.visualize_cfg.get("visualize", False):
            print("[gapartnet_amd] visualisation / pose rendering of on_test_epoch_end is outside the accelerated hot "
                  "path (SURVEY.md §2.1 #10) and is skipped; metrics are computed as in validation")
        self._epoch_end_metrics()

    def configure_optimizers(self):
        """Adam(lr) over every parameter (model.py:1051-1055); FusedAdam is a torch.optim.Adam whose step on the GPU is one
        launch (gapartnet_amd/optim.py), with torch's implementation for everything else"""
        from ..optim import FusedAdam
        opt = FusedAdam(list(self.parameters()), lr=self.learning_rate)
        gated = [p for m in (self.score_unet, self.score_head, self.npcs_unet, self.npcs_head) for p in m.parameters()]
        opt.set_gate(gated, lambda: self._prop_gate)
        return opt
