"""Synthetic GAPartNet-like scenes — there is no dataset in the container, so benchmarks, the
smoke test and the model tests use seeded scenes with the same tuple layout as the reference's ``.pth`` files
(dataset/process_tools/convert_rendered_into_input.py:1-11,156-158):
    (xyz f32 [N,3], rgb f32 [N,3], sem i32/i64 [N], instance i32 [N], npcs f32 [N,3], pixel_idx i32 [N,2])

A scene is a partial scan of a cuboid "object" (its three camera-facing faces) carrying K raised rectangular
"parts" (handles, buttons, lids ...): jittered-grid surface samples (FPS-like spacing), 2 mm depth noise, random
rotation, then centred and scaled into the unit ball exactly like convert_rendered_into_input.py:71-87.
Semantic label 0 = "others" (instance -100), parts get a class in 1..9 and NPCS coordinates in [-0.5, 0.5]^3.
"""
from typing import Tuple

import numpy as np

from ..structure.point_cloud import PointCloud


def _random_rotation(rng: np.random.Generator) -> np.ndarray:
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_scene_arrays(seed: int, n_points: int = 20000, extent_range: Tuple[float, float] = (0.4, 1.0),
                      parts_range: Tuple[int, int] = (3, 12)):
    """-> the reference's 6-tuple of numpy arrays for one scene."""
    rng = np.random.default_rng(seed)
    half = rng.uniform(*extent_range, size=3)
    n_parts = int(rng.integers(parts_range[0], parts_range[1] + 1))

    # three visible faces: normal axis a, in-plane axes (b, c); area-proportional point budget
    faces = [(0, 1, 2), (1, 0, 2), (2, 0, 1)]
    areas = np.array([half[b] * half[c] for _, b, c in faces])
    budget = np.floor(n_points * areas / areas.sum()).astype(int)
    budget[0] += n_points - budget.sum()

    # parts: rectangle on a face, raised by h along the normal
    parts = []
    for k in range(n_parts):
        f = int(rng.integers(0, 3))
        a, b, c = faces[f]
        size = rng.uniform(0.10, 0.35, size=2) * np.array([half[b], half[c]])
        centre = rng.uniform(-1, 1, size=2) * (np.array([half[b], half[c]]) - size)
        parts.append(dict(face=f, centre=centre, size=size, height=rng.uniform(0.02, 0.08),
                          sem=int(rng.integers(1, 10))))

    xyz_all, sem_all, ins_all, npcs_all = [], [], [], []
    for f, (a, b, c) in enumerate(faces):
        n = int(budget[f])
        # jittered grid: one sample per cell of a g0 x g1 grid, random subset of n cells
        ratio = half[b] / half[c]
        g1 = max(int(np.ceil(np.sqrt(n / ratio))), 1)
        g0 = max(int(np.ceil(n / g1)), 1)
        cells = rng.choice(g0 * g1, size=n, replace=False)
        u = ((cells // g1) + rng.uniform(0, 1, n)) / g0 * 2 - 1
        v = ((cells % g1) + rng.uniform(0, 1, n)) / g1 * 2 - 1
        p = np.zeros((n, 3))
        p[:, b], p[:, c], p[:, a] = u * half[b], v * half[c], half[a]
        sem = np.zeros(n, np.int64)
        ins = np.full(n, -100, np.int32)
        npcs = np.zeros((n, 3))
        for k, part in enumerate(parts):
            if part["face"] != f:
                continue
            d = np.stack([p[:, b], p[:, c]], 1) - part["centre"]
            inside = (np.abs(d) <= part["size"]).all(1) & (ins < 0)
            p[inside, a] += part["height"]
            sem[inside], ins[inside] = part["sem"], k
            npcs[inside, 0] = d[inside, 0] / (2 * part["size"][0])
            npcs[inside, 1] = d[inside, 1] / (2 * part["size"][1])
            npcs[inside, 2] = rng.uniform(-0.5, 0.5, int(inside.sum()))
        p[:, a] += rng.normal(0, 0.002, n)
        xyz_all.append(p); sem_all.append(sem); ins_all.append(ins); npcs_all.append(npcs)

    xyz = np.concatenate(xyz_all) @ _random_rotation(rng).T
    sem, ins, npcs = np.concatenate(sem_all), np.concatenate(ins_all), np.concatenate(npcs_all)
    perm = rng.permutation(xyz.shape[0])
    xyz, sem, ins, npcs = xyz[perm], sem[perm], ins[perm], npcs[perm]
    # guarantee at least one labelled instance (the reference data loader stops otherwise, dataset/gapartnet.py:69-70)
    if not (ins >= 0).any():
        ins[:50], sem[:50], npcs[:50] = 0, 1, rng.uniform(-0.5, 0.5, (50, 3))
    # centre on the bounding-box middle and scale so the farthest point lies on the unit sphere
    centre = (xyz.max(0) + xyz.min(0)) / 2
    xyz = xyz - centre
    xyz = xyz / np.linalg.norm(xyz, axis=1).max()
    rgb = rng.uniform(0, 1, (xyz.shape[0], 3))
    pixel_idx = np.stack([np.arange(xyz.shape[0]) // 800, np.arange(xyz.shape[0]) % 800], 1).astype(np.int32)
    return (xyz.astype(np.float32), rgb.astype(np.float32), sem.astype(np.int32), ins.astype(np.int32),
            npcs.astype(np.float32), pixel_idx)


def make_scene(seed: int, n_points: int = 20000, **kw) -> PointCloud:
    """numpy PointCloud with the fields load_data() produces (dataset/gapartnet.py:208-229)."""
    xyz, rgb, sem, ins, npcs, _ = make_scene_arrays(seed, n_points, **kw)
    return PointCloud(pc_id=f"Synthetic_{seed}_0_0", obj_cat=-1, points=np.concatenate([xyz, rgb], axis=-1, dtype=np.float32),
                      sem_labels=sem.astype(np.int64), instance_labels=ins.astype(np.int32),
                      gt_npcs=npcs.astype(np.float32))
