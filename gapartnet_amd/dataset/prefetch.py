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
import queue
import threading
from typing import Iterable, Optional

import torch

from ..structure.point_cloud import PointCloud, PointCloudBatch, finish_voxels


import os

# GPN_PREFETCH_DEFER=1: the voxelisation of batch i + 2 is queued one step before its sizes are read, so that the read never
# waits (hip_ops.voxelize_scenes_begin / _finish).  Measured (round 4, four interleaved same-box pairs): the wait of 0.54 ms per
# step is gone from the host (tools/host_wait.py: 0 ms) and the step is SLOWER, 8.10 -> 8.44 ms, every pair - the host is the
# critical resource of the step now and the wait was the moment the GPU caught up with it; the earlier start of the batch
# preparation kernels beside the backbone costs more than the wait did.  Default off; kept as a switch.
_DEFER_VOXELS = os.environ.get("GPN_PREFETCH_DEFER", "0") == "1"
# GPN_PREFETCH_THREAD=1: batch preparation is issued by a worker thread of the prefetcher instead of by the thread that runs the
# training step (from its mid-step hook).  Round 4: the host's main thread needs as long for a step as the GPU does (DESIGN.md
# 5.3), and ~1.3 ms of that is this preparation - 15 rulebook-builder calls, the voxeliser, tile orders, ~100 small allocations.
# Measured (four interleaved same-box pairs): 7.79 / 7.82 / 7.82 / 7.97 ms per step without, 8.14 / 8.14 / 8.12 / 8.17 with the
# worker, and 20.5 instead of 12.9 CPU-ms per step: the preparation is hundreds of short calls, each releasing and re-taking the
# GIL, and the two threads spend their time handing it to each other.  Default off; the way to take this work off the main
# thread is one library call per batch, not a second Python thread.
_THREADED = os.environ.get("GPN_PREFETCH_THREAD", "0") == "1"
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


class _Worker:
    """one daemon thread that runs submitted callables in order on ``device``; ``result()`` returns (or re-raises) the oldest
    outstanding one's outcome"""

    def __init__(self, device: torch.device):
        self._jobs, self._results = queue.SimpleQueue(), queue.SimpleQueue()
        self._device = device
        self._thread = threading.Thread(target=self._run, name="gpn-batch-preparation", daemon=True)
        self._thread.start()

    def _run(self):
        torch.cuda.set_device(self._device)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            try:
                with torch.no_grad():
                    self._results.put((job(), None))
            except BaseException as exc:  # handed to the consumer thread, which re-raises it
                self._results.put((None, exc))

    def submit(self, job):
        self._jobs.put(job)

    def result(self):
        value, exc = self._results.get()
        if exc is not None:
            raise exc
        return value

    def stop(self):
        self._jobs.put(None)


class DevicePrefetcher:
    """``for batch in DevicePrefetcher(loader, model, device)``: batches come out collated, voxelised and with the
    backbone's rulebooks built, each prepared on a side stream (with GPN_PREFETCH_THREAD=1 by a worker thread) while the
    previous one trains."""

    def __init__(self, batches: Iterable, model, device: torch.device, augmentation: Optional[dict] = None):
        assert device.type == "cuda", "batch preparation runs on the GPU (the product has no CPU path)"
        self.batches, self.model, self.device = batches, model, device
        self.augmentation = augmentation  # for raw scenes (dataset device_pipeline=True): drawn per batch, applied on the GPU
        self.stream = torch.cuda.Stream(device=device)
        self._consumer_mark = None
        self._worker = None       # created on first use (threaded mode)
        self._outstanding = False  # a preparation job is running on the worker

    def _begin(self, raw):
        """first half of a batch's preparation, on the side stream: collate and QUEUE the voxelisation - its sizes are not read
        here.  -> (batch still without its voxel part, or the finished batch when nothing could be deferred)"""
        if raw is None:
            return None
        with torch.cuda.stream(self.stream):
            if self._consumer_mark is not None:
                # Everything this stream allocates from here on may reuse blocks of batches the training stream has
                # finished with: wait for the point of the training stream up to which that is true (see __iter__).
                self.stream.wait_event(self._consumer_mark)
            if isinstance(raw, PointCloudBatch):
                return raw
            backbone = getattr(self.model, "backbone", None)
            pcs = [pc.to(self.device) if hasattr(pc, "to") else pc for pc in raw]
            raw_scenes = pcs[0].num_instances is None and pcs[0].instance_labels is not None
            levels = 0  # the backbone's coarse levels: their row counts come back with the voxelisation's one host read
            if backbone is not None and getattr(backbone, "use_native_executor", False):
                from ..network import net_exec
                prog = net_exec.program_for(backbone)
                levels = prog.n_levels - 1 if prog is not None and prog.n_levels > 2 else 0
            return PointCloud.collate(pcs, voxel_size=self.model.voxel_size,
                                      augmentation=self.augmentation if raw_scenes else None, pyramid_levels=levels,
                                      defer_voxels=_DEFER_VOXELS)

    def _finish(self, batch):
        """second half, one step later: the read of the voxelisation's sizes (its kernels ran long ago: no wait), the voxel
        tensor, the backbone's rulebooks.  -> (batch, event the consumer waits for)"""
        if batch is None:
            return None
        with torch.cuda.stream(self.stream):
            if self._consumer_mark is not None:
                self.stream.wait_event(self._consumer_mark)
            batch = finish_voxels(batch)
            backbone = getattr(self.model, "backbone", None)
            if backbone is not None and getattr(backbone, "use_native_executor", False) and batch.voxel_tensor is not None:
                from ..network import net_exec
                prog = net_exec.program_for(backbone)
                if prog is not None and batch.voxel_tensor.features.shape[0] > 0:
                    prog.rulebooks(batch.voxel_tensor)  # cached in voxel_tensor.indice_dict under the modules' keys
            done = torch.cuda.Event()
            done.record(self.stream)
        return batch, done

    def _prepare_pending(self):
        """called by the model in the middle of its step (GAPartNet._prefetch_hook): after the backbone and the point heads
        have been launched and before the clustering code.  Finishes the NEXT batch (whose voxelisation was queued a step ago)
        and queues the voxelisation of the one after it."""
        if self._has_pending:
            self._has_pending = False
            begun, pending, self._pending = self._begun, self._pending, None
            if _THREADED:
                if self._worker is None:
                    self._worker = _Worker(self.device)
                self._outstanding = True
                self._worker.submit(lambda: (self._finish(begun), self._begin(pending)))
            else:
                self._ahead = self._finish(begun)
                self._begun = self._begin(pending)

    def _collect(self):
        """take over what the worker prepared since the last hand-out (waits for it if it is still at it)"""
        if self._outstanding:
            self._outstanding = False
            self._ahead, self._begun = self._worker.result()

    def _mark_consumer(self):
        """order everything the side stream does from now on behind what the training stream has been given so far"""
        mark = torch.cuda.Event()
        mark.record(torch.cuda.current_stream(self.device))
        self._consumer_mark = mark

    def __iter__(self):
        it = iter(self.batches)
        # Invariant of the ordering scheme below: a batch may be used by the training stream until the third batch after it is
        # handed out (two are in preparation at any time: one finished, one with its voxelisation queued).  A (re-)started
        # iteration has no such history: the last batches of a previous pass over this object (or anything else the consumer
        # still has in flight) may still be read by kernels enqueued AFTER the last mark, and their blocks are free for the side
        # stream to reuse once `_held` was dropped - so the first preparation waits for the training stream as it stands now.
        self._mark_consumer()
        self._ahead = self._finish(self._begin(next(it, None)))
        self._begun = self._begin(next(it, None))
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
                self._collect()
        finally:
            if self._outstanding:  # (the consumer left mid-iteration: the job's batches are dropped with the rest)
                self._outstanding = False
                try:
                    self._worker.result()
                except BaseException:
                    pass
            # the batches held back are released here: whatever the side stream allocates next (a later iteration of this
            # object) must come after every kernel the consumer enqueued on them
            try:
                self._mark_consumer()
            except Exception:  # interpreter teardown: no device left to order against
                self._consumer_mark = None
            self._held = (None, None)
            self._begun = None
            if hook_owner is not None:
                hook_owner.__dict__["_prefetch_hook"] = None  # plain attribute; safe at interpreter teardown too
