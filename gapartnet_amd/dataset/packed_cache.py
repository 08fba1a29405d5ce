"""Packed scene cache: the ``.pth`` scenes of a split as ONE memory-mapped array file, and a loader that hands whole batches to
the GPU with one pinned copy per field (SURVEY.md §8f "next" rank 1, the disk side of the loader).

The reference reads every scene with ``torch.load`` in a DataLoader worker (dataset/gapartnet.py:208-229 - a pickle of six numpy
arrays, 1.34 MB per 20k-point scene), prepares it there and pickles the result back to the main process.  Measured in round 4
(profiles/r04_pth_loader.json): 8 workers on a 16-core quota deliver ~250 scenes/s - half of what the training step consumes, so a
``fit`` from disk ran at half the bench's rate.  None of that work depends on the step: the bytes on disk never change.  So, once per
split (first use; rebuilt when the file list changes):

    <cache_dir>/<split>-<digest>/points.f32  [T, 6]  xyz + rgb        (T = points of all scenes)
                                 sem.i16     [T]     semantic labels  (the reference's are small non-negative ints)
                                 ins.i32     [T]     instance labels  (raw: -100 = unlabelled; compacted per batch on the GPU)
                                 npcs.f32    [T, 3]
                                 meta.json           names, object categories, point offsets, source list digest

= 42 bytes per point (0.84 MB per 20k-point scene), read through ``numpy.memmap`` (page cache after the first epoch).  A batch is
``batch_size`` slices copied into one PINNED staging block per field by a background thread (numpy releases the GIL for the
copies) and four asynchronous host-to-device copies on the consumer's stream; the scenes then are views of those four device
tensors, raw, exactly what ``GAPartNetDataset(device_pipeline=True)`` hands over - label compaction, augmentation, instance
statistics and voxelisation run per batch on the GPU (dataset/device_pipeline.py), unchanged.
"""
import hashlib
import json
import os
import queue
import threading
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..misc.info import OBJECT_NAME2ID
from ..structure.point_cloud import PointCloud

FIELDS = (("points", np.float32, 6), ("sem", np.int16, 0), ("ins", np.int32, 0), ("npcs", np.float32, 3))
_FILES = {"points": "points.f32", "sem": "sem.i16", "ins": "ins.i32", "npcs": "npcs.f32"}


def _digest(paths: Sequence[str]) -> str:
    h = hashlib.sha256()
    for p in paths:
        st = os.stat(p)
        # (nanosecond mtime: a file rewritten within the same second at the same size must not find the old cache)
        h.update(f"{os.path.basename(p)}\0{st.st_size}\0{st.st_mtime_ns}\n".encode())
    return h.hexdigest()[:16]


class _RawScenes(torch.utils.data.Dataset):
    """the six arrays of a ``.pth`` file as they are (worker side of the cache build)"""

    def __init__(self, paths):
        self.paths = list(paths)

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        d = torch.load(self.paths[i], weights_only=False)
        xyz, rgb = np.asarray(d[0], np.float32), np.asarray(d[1], np.float32)
        return (np.concatenate([xyz, rgb], axis=-1), np.asarray(d[2]).astype(np.int64), np.asarray(d[3]).astype(np.int32),
                np.asarray(d[4], np.float32))


class PackedScenes:
    """the memory-mapped arrays of one split; ``PackedScenes.open(paths, cache_dir)`` builds them on first use"""

    def __init__(self, directory: str):
        with open(os.path.join(directory, "meta.json")) as fh:
            meta = json.load(fh)
        self.names: List[str] = meta["names"]
        self.obj_cat: List[int] = meta["obj_cat"]
        self.offsets = np.asarray(meta["offsets"], np.int64)
        total = int(self.offsets[-1])
        self.arrays = {}
        for name, dtype, width in FIELDS:
            shape = (total, width) if width else (total,)
            self.arrays[name] = np.memmap(os.path.join(directory, _FILES[name]), dtype=dtype, mode="r", shape=shape)
        self.directory = directory

    def __len__(self):
        return len(self.names)

    @staticmethod
    def open(paths: Sequence[str], cache_dir: str, split: str, num_workers: int = 0, max_points: Optional[int] = None) -> "PackedScenes":
        """``max_points``: the dataset's bound on a scene's size (GAPartNetDataset applies it per scene through ``downsample``,
        dataset/gapartnet.py:43-46, which raises on a larger scene): checked here for every scene of the cache, so that an
        oversized scene is reported when the loader is made instead of entering a batch silently"""
        paths = list(paths)
        directory = os.path.join(cache_dir, f"{split}-{_digest(paths)}")
        if not os.path.exists(os.path.join(directory, "meta.json")):
            PackedScenes._build(paths, directory, num_workers)
        scenes = PackedScenes(directory)
        if max_points is not None and len(scenes):
            counts = np.diff(scenes.offsets)
            worst = int(counts.argmax())
            if int(counts[worst]) > max_points:
                raise AssertionError((scenes.names[worst], int(counts[worst]), max_points))
        return scenes

    @staticmethod
    def _build(paths, directory, num_workers):
        tmp = directory + f".tmp{os.getpid()}"
        os.makedirs(tmp, exist_ok=True)
        loader = torch.utils.data.DataLoader(_RawScenes(paths), batch_size=None, shuffle=False, num_workers=num_workers)
        counts = []
        files = {name: open(os.path.join(tmp, _FILES[name]), "wb") for name, _, _ in FIELDS}
        try:
            for points, sem, ins, npcs in loader:
                points, sem, ins, npcs = (np.asarray(a) for a in (points, sem, ins, npcs))
                if sem.size and (sem.min() < -32768 or sem.max() > 32767):
                    raise ValueError("semantic labels do not fit int16")
                counts.append(int(points.shape[0]))
                files["points"].write(np.ascontiguousarray(points, np.float32).tobytes())
                files["sem"].write(np.ascontiguousarray(sem, np.int16).tobytes())
                files["ins"].write(np.ascontiguousarray(ins, np.int32).tobytes())
                files["npcs"].write(np.ascontiguousarray(npcs, np.float32).tobytes())
        finally:
            for fh in files.values():
                fh.close()
        names = [os.path.basename(p).split(".")[0] for p in paths]
        meta = dict(names=names, obj_cat=[int(OBJECT_NAME2ID.get(n.split("_")[0], -1)) for n in names],
                    offsets=[0] + [int(v) for v in np.cumsum(counts)], version=1)
        with open(os.path.join(tmp, "meta.json"), "w") as fh:
            json.dump(meta, fh)
        try:
            os.replace(tmp, directory)  # (atomic: another rank building the same cache loses the race harmlessly)
        except OSError:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
            if not os.path.exists(os.path.join(directory, "meta.json")):
                raise

    def scene(self, i: int) -> PointCloud:
        """one scene as host tensors (copies) - what GAPartNetDataset(device_pipeline=True)[i] returns"""
        a, b = int(self.offsets[i]), int(self.offsets[i + 1])
        arr = self.arrays
        return PointCloud(pc_id=self.names[i], obj_cat=self.obj_cat[i], points=torch.from_numpy(np.array(arr["points"][a:b])),
                          sem_labels=torch.from_numpy(arr["sem"][a:b].astype(np.int64)),
                          instance_labels=torch.from_numpy(np.array(arr["ins"][a:b])),
                          gt_npcs=torch.from_numpy(np.array(arr["npcs"][a:b])))


class SceneList(list):
    """a list of raw scenes that are consecutive slices of four tensors (``packed`` = points, sem_labels, instance_labels,
    gt_npcs of the whole batch): device_pipeline.prepare_batch takes those instead of concatenating the slices again"""
    packed = None


class StagedBatch:
    """a batch of raw scenes in ONE pinned block per field; ``scenes(device)`` queues the four host-to-device copies on the
    current stream and returns the scenes as views of the device tensors"""

    def __init__(self, slot, ids, counts, names, obj_cat):
        self.slot, self.ids, self.counts, self.names, self.obj_cat = slot, ids, counts, names, obj_cat

    def __len__(self):
        return len(self.ids)

    def scenes(self, device) -> List[PointCloud]:
        total = int(sum(self.counts))
        dev = {}
        for name in ("points", "sem", "ins", "npcs"):
            dev[name] = self.slot.host[name][:total].to(device, non_blocking=True)
        sem64 = dev["sem"].to(torch.int64)
        if device.type == "cuda":
            self.slot.copied = torch.cuda.Event()
            self.slot.copied.record()  # the staging block may be refilled once these copies have run
        else:
            self.slot.copied = None
        out, a = SceneList(), 0
        out.packed = (dev["points"], sem64, dev["ins"], dev["npcs"])
        for n, name, cat in zip(self.counts, self.names, self.obj_cat):
            b = a + n
            out.append(PointCloud(pc_id=name, obj_cat=cat, points=dev["points"][a:b], sem_labels=sem64[a:b],
                                  instance_labels=dev["ins"][a:b], gt_npcs=dev["npcs"][a:b]))
            a = b
        return out

    # (so that code written for lists of scenes - trainer.move_batch, len() - keeps working)
    def to(self, device):
        return self.scenes(torch.device(device))


class _Slot:
    def __init__(self, capacity: int, pin: bool):
        def block(dtype, width):
            shape = (capacity, width) if width else (capacity,)
            t = torch.empty(shape, dtype=dtype)
            return t.pin_memory() if pin else t
        self.host = {"points": block(torch.float32, 6), "sem": block(torch.int16, 0), "ins": block(torch.int32, 0),
                     "npcs": block(torch.float32, 3)}
        self.np = {k: v.numpy() for k, v in self.host.items()}
        self.copied = None  # event behind the last host-to-device copies out of this block


class PackedSceneLoader:
    """iterable of ``StagedBatch``: scene indices from ``sampler`` (or a fresh permutation per epoch from torch's global
    generator when ``shuffle``), gathered out of the memory-mapped cache into a ring of pinned staging blocks by a background
    thread, ``depth`` batches ahead"""

    def __init__(self, scenes: PackedScenes, batch_size: int, shuffle: bool, drop_last: bool, sampler=None, depth: int = 3,
                 pin: Optional[bool] = None, index_map: Optional[Sequence[int]] = None):
        self.scenes, self.batch_size, self.shuffle, self.drop_last, self.sampler = scenes, batch_size, shuffle, drop_last, sampler
        self.depth = depth
        # position i of the dataset this loader stands for = scene index_map[i] of the cache (the dataset shuffles / truncates its
        # file list per run, the cache is in sorted order); None: the identity
        self.index_map = None if index_map is None else [int(i) for i in index_map]
        self.pin = torch.cuda.is_available() if pin is None else pin
        sizes = np.diff(scenes.offsets)
        self.capacity = int(np.sort(sizes)[-batch_size:].sum()) if len(sizes) else 0  # the largest batch there can be
        self._slots = None
        self.dataset = scenes  # (what code that looks at loader.dataset expects to find)

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else self._n()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _n(self):
        return len(self.index_map) if self.index_map is not None else len(self.scenes)

    def _order(self):
        if self.sampler is not None:
            order = [int(i) for i in self.sampler]
        elif self.shuffle:
            order = torch.randperm(self._n()).tolist()
        else:
            order = list(range(self._n()))
        return order if self.index_map is None else [self.index_map[i] for i in order]

    def _fill(self, slot: _Slot, ids) -> StagedBatch:
        if slot.copied is not None:
            slot.copied.synchronize()  # the previous batch staged here has left for the device
            slot.copied = None
        sc, a, counts = self.scenes, 0, []
        for i in ids:
            lo, hi = int(sc.offsets[i]), int(sc.offsets[i + 1])
            n = hi - lo
            for name in ("points", "sem", "ins", "npcs"):
                slot.np[name][a:a + n] = sc.arrays[name][lo:hi]
            a += n
            counts.append(n)
        return StagedBatch(slot, list(ids), counts, [sc.names[i] for i in ids], [sc.obj_cat[i] for i in ids])

    def __iter__(self):
        order = self._order()
        bs = self.batch_size
        batches = [order[i:i + bs] for i in range(0, len(order), bs)]
        if self.drop_last and batches and len(batches[-1]) < bs:
            batches.pop()
        if self._slots is None:
            # depth batches queued + one being filled + one in the consumer's hands + one whose copy may still be running
            self._slots = [_Slot(self.capacity, self.pin) for _ in range(self.depth + 3)]
        out: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def work():
            try:
                for k, ids in enumerate(batches):
                    if stop.is_set():
                        return
                    item = self._fill(self._slots[k % len(self._slots)], ids)
                    while not stop.is_set():
                        try:
                            out.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                out.put(None)
            except BaseException as exc:  # handed to the consumer
                out.put(exc)

        thread = threading.Thread(target=work, name="gpn-packed-loader", daemon=True)
        thread.start()
        try:
            while True:
                item = out.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            thread.join(timeout=5.0)
