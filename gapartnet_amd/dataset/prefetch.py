"""On-device batch preparation one step ahead (SURVEY.md §8f "next": loader side of the hot path).

The reference voxelises every scene in CPU DataLoader workers (dataset/gapartnet.py:179-205) and builds the spconv
rulebooks inside the forward pass.  Here both run on the GPU, and this iterator runs them for batch i+1 on a SECOND
stream while the GPU is still busy with the backward pass of batch i-1 / forward of batch i: the host reads that batch
preparation needs (voxel count, grid extent, per-level row counts) then wait only for that stream's own small kernels
instead of draining the training stream, which removes the longest host stall of the step.  The preparation is triggered
from INSIDE the training step (a hook the model calls between launching the backbone and the first host read of the
clustering code), i.e. exactly where the host would otherwise wait for the GPU.

Yields ``PointCloudBatch`` objects (what ``GAPartNet.training_step`` accepts directly) whose ``voxel_tensor`` already
carries the rulebook pyramid of the backbone in its ``indice_dict``.
"""
import os
from typing import Iterable, Optional

import numpy as np
import torch

from .. import _C
from ..structure.point_cloud import PointCloud, PointCloudBatch

# (Round 4 measured two variants of this iterator and round 5 removed them - profiles/r04_findings.md: the voxelisation queued one
# step before its sizes are read (the 0.54 ms wait disappears from the host and the step gets SLOWER, 8.10 -> 8.44 ms: the
# preparation kernels then start beside the backbone), and the preparation issued by a Python worker thread (hundreds of short
# GIL-releasing calls per batch: 7.8 -> 8.14 ms per step, 12.9 -> 20.5 CPU-ms).)
_LEAF_TYPES = (str, bytes, int, float, bool, type(None))


def _tensors(root):
    """every distinct tensor reachable from a prepared batch (dataclasses, containers, sparse tensors, rulebooks): an
    explicit stack over ``__dict__`` / container items - this runs once per step over ~150 tensors, and the generic
    recursive walk over dataclasses.fields() it replaces cost 0.9 ms of host time per step"""
    out, seen, stack = [], set(), [root]
    tensor_t = torch.Tensor
    while stack:
        obj = stack.pop()
        if isinstance(obj, tensor_t):
            if id(obj) not in seen:
                seen.add(id(obj))
                out.append(obj)
            continue
        if isinstance(obj, _LEAF_TYPES) or id(obj) in seen:
            continue
        seen.add(id(obj))
        if isinstance(obj, dict):
            stack.extend(obj.values())
        elif isinstance(obj, (list, tuple)):
            stack.extend(obj)
        else:
            d = getattr(obj, "__dict__", None)
            if d:
                stack.extend(d.values())
    return out


class DevicePrefetcher:
    """``for batch in DevicePrefetcher(loader, model, device)``: batches come out collated, voxelised and with the
    backbone's rulebooks built, each prepared on a side stream while the previous one trains."""

    def __init__(self, batches: Iterable, model, device: torch.device, augmentation: Optional[dict] = None,
                 native: Optional[bool] = None):
        assert device.type == "cuda", "batch preparation runs on the GPU (the product has no CPU path)"
        self.batches, self.model, self.device = batches, model, device
        self.augmentation = augmentation  # for raw scenes (dataset device_pipeline=True): drawn per batch, applied on the GPU
        self.stream = torch.cuda.Stream(device=device)
        self._consumer_mark = None
        # native: voxelisation, its one host read and the backbone's rulebook pyramid as ONE library call into ONE arena
        # (gpn_backbone_prepare, csrc/prepare.hip) instead of 17 calls and ~230 allocations; bit-identical tables
        # (tests/test_gpu_prepare.py).  Issued by THIS thread: round 5 measured the same call on a worker thread (no Python in
        # it at all) - 7.52 -> 7.78 ms per step and 12.7 -> 19.5 CPU-ms, every one of three interleaved pairs, with or without
        # holding its kernels back until the backbone has run; from a packed cache 812 - 856 -> 690 - 720 point-clouds/s
        # (profiles/r05_findings.md): a second launching host thread beside the training thread costs more than the 1.3 ms of
        # preparation it takes off it, as round 4's Python worker did.
        self.native = True if native is None else bool(native)

    def _program(self):
        backbone = getattr(self.model, "backbone", None)
        if backbone is not None and getattr(backbone, "use_native_executor", False):
            from ..network import net_exec
            return net_exec.program_for(backbone)
        return None

    def _prepare(self, raw):
        """a batch's preparation on the side stream: collate here; voxelisation, its one host read and the backbone's
        rulebooks as one native call (``native``) or call by call.  -> (batch, event the consumer waits for) or None"""
        if raw is None:
            return None
        with torch.cuda.stream(self.stream):
            if self._consumer_mark is not None:
                # Everything this stream allocates from here on may reuse blocks of batches the training stream has
                # finished with: wait for the point of the training stream up to which that is true (see __iter__).
                self.stream.wait_event(self._consumer_mark)
            prog = self._program()
            native = self.native and prog is not None and not isinstance(raw, PointCloudBatch)
            if isinstance(raw, PointCloudBatch):
                batch = raw
            else:
                if hasattr(raw, "scenes"):  # dataset.packed_cache.StagedBatch: one pinned block per field, four copies
                    pcs = raw.scenes(self.device)
                else:
                    pcs = [pc.to(self.device) if hasattr(pc, "to") else pc for pc in raw]
                raw_scenes = pcs[0].num_instances is None and pcs[0].instance_labels is not None
                native = native and pcs[0].voxel_coords is None
                # the backbone's coarse levels: their row counts come back with the voxelisation's one host read
                levels = prog.n_levels - 1 if prog is not None and prog.n_levels > 2 else 0
                batch = PointCloud.collate(pcs, voxel_size=self.model.voxel_size,
                                           augmentation=self.augmentation if raw_scenes else None, pyramid_levels=levels,
                                           voxels=not native)
            if native and batch.voxel_tensor is None:
                from .. import hip_ops
                counts = batch.scene_counts
                if len(set(counts)) == 1:
                    offsets = torch.arange(len(counts) + 1, dtype=torch.int64, device=self.device) * int(counts[0])
                else:
                    offsets = torch.as_tensor([0] + list(np.cumsum(counts)), dtype=torch.int64).pin_memory().to(self.device, non_blocking=True)
                job = hip_ops.PreparedBackbone(batch.points[:, :3], batch.points, offsets, self.model.voxel_size, prog.n_levels,
                                               prog.ident_levels(), self.stream)
                job.run()  # (blocks for the voxeliser's sizes: where the call-by-call path waits too)
                if job.rc:
                    raise _C.GpnError(f"gpn_backbone_prepare failed (code {job.rc}): {job.error}")
                if not job.fallback():  # (else: an empty level, a cell index beyond the packed keys, ...: the per-call path copes)
                    from ..structure.point_cloud import spconv
                    prepared = job.wrap()
                    vt = spconv.SparseConvTensor(prepared["features"], prepared["indices"], prepared["spatial_shape"], batch.batch_size)
                    vt.level_counts = list(prepared["level_counts"])
                    prog.adopt(vt, prepared["levels"])
                    batch.voxel_tensor, batch.pc_voxel_id, batch.pc_voxel_csr = vt, prepared["pc_voxel_id"], prepared["csr"]
            batch = self._finish_here(batch, prog)
            done = torch.cuda.Event()
            done.record(self.stream)
        return batch, done

    def _finish_here(self, batch, prog):
        """the per-call path (on the calling thread, current stream = the side stream): voxelise if still needed, rulebooks"""
        if batch.voxel_tensor is None and batch.scene_counts is not None:
            from ..structure.point_cloud import spconv, voxelize_scenes
            levels = prog.n_levels - 1 if prog is not None and prog.n_levels > 2 else 0
            vox = voxelize_scenes(batch.points[:, :3], batch.points, batch.scene_counts, self.model.voxel_size, levels)
            indices, feats, shape, batch.pc_voxel_id, batch.pc_voxel_csr = vox[:5]
            batch.voxel_tensor = spconv.SparseConvTensor(feats, indices, shape, batch.batch_size)
            if levels:
                batch.voxel_tensor.level_counts = list(vox[5])
        if prog is not None and batch.voxel_tensor is not None and batch.voxel_tensor.features.shape[0] > 0:
            prog.rulebooks(batch.voxel_tensor)  # cached in voxel_tensor.indice_dict under the modules' keys
        return batch

    def _prepare_pending(self):
        """called by the model in the middle of its step (GAPartNet._prefetch_hook): after the backbone and the point heads
        have been launched and before the clustering code - where the host would otherwise run ahead of the GPU.  Prepares
        the NEXT batch."""
        if self._has_pending:
            self._has_pending = False
            pending, self._pending = self._pending, None
            self._ahead = self._prepare(pending)

    def _mark_consumer(self):
        """order everything the side stream does from now on behind what the training stream has been given so far"""
        mark = torch.cuda.Event()
        mark.record(torch.cuda.current_stream(self.device))
        self._consumer_mark = mark

    def __iter__(self):
        it = iter(self.batches)
        # Invariant of the ordering scheme below: a batch may be used by the training stream until the second batch after it is
        # handed out (`_held` keeps the last two referenced; one more is in preparation at any time).  A (re-)started
        # iteration has no such history: the last batches of a previous pass over this object (or anything else the consumer
        # still has in flight) may still be read by kernels enqueued AFTER the last mark, and their blocks are free for the side
        # stream to reuse once `_held` was dropped - so the first preparation waits for the training stream as it stands now.
        self._mark_consumer()
        self._ahead = self._prepare(next(it, None))
        self._pending, self._has_pending = None, False
        hook_owner = self.model if hasattr(self.model, "_prefetch_hook") else None
        try:
            while self._ahead is not None:
                batch, done = self._ahead
                self._ahead = None
                consumer = torch.cuda.current_stream(self.device)
                consumer.wait_event(done)
                # The batch was allocated on the side stream and is used on the training stream.  Instead of telling the
                # allocator about every one of its ~150 tensors (Tensor.record_stream: 0.3 ms of host time per step with the
                # walk that finds them), the side stream is ordered behind the training stream: a batch stays referenced
                # here until the NEXT one is handed out, and the preparation after that waits for this mark - every kernel
                # that read a batch whose blocks the side stream can get back has then completed.  (The mark is in the
                # past by the time the side stream reaches it: no stall.)
                self._held = (self._held[1], batch) if hasattr(self, "_held") else (None, batch)
                mark = torch.cuda.Event()
                mark.record(consumer)
                self._consumer_mark = mark
                self._pending, self._has_pending = next(it, None), True
                if hook_owner is not None:
                    hook_owner._prefetch_hook = self._prepare_pending
                yield batch
                self._prepare_pending()  # the consumer's step did not reach the hook (eval, early exit): prepare now
        finally:
            # the batches held back are released here: whatever the side stream allocates next (a later iteration of this
            # object) must come after every kernel the consumer enqueued on them
            try:
                self._mark_consumer()
            except Exception:  # interpreter teardown: no device left to order against
                self._consumer_mark = None
            self._held = (None, None)
            if hook_owner is not None:
                hook_owner.__dict__["_prefetch_hook"] = None  # plain attribute; safe at interpreter teardown too
