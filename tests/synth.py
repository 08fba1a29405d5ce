"""Seeded synthetic inputs shared by the CPU and GPU tests (no reference data involved)."""
import numpy as np


def random_sparse_indices(rng, batch, shape, n):
    """n unique (b,x,y,z) int32 rows, in random order."""
    total = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(total, size=min(n, total), replace=False)
    z = lin % shape[2]; lin //= shape[2]
    y = lin % shape[1]; lin //= shape[1]
    x = lin % shape[0]; b = lin // shape[0]
    return np.stack([b, x, y, z], 1).astype(np.int32)


def surface_indices(rng, batch, shape, n_per):
    """voxels on a noisy 2-D sheet per batch element (the occupancy pattern of a depth scan), sorted by key."""
    out = []
    for b in range(batch):
        u = rng.uniform(0, 1, (n_per, 2))
        x = (u[:, 0] * (shape[0] - 1)).astype(np.int64)
        y = (u[:, 1] * (shape[1] - 1)).astype(np.int64)
        z = ((0.5 + 0.3 * np.sin(4 * u[:, 0]) * np.cos(3 * u[:, 1])) * (shape[2] - 1)).astype(np.int64)
        c = np.unique(np.stack([np.full_like(x, b), x, y, z], 1), axis=0)
        out.append(c)
    return np.concatenate(out, 0).astype(np.int32)


def clustered_points(rng, n_scenes, pts_per_scene, n_clusters=6, spread=0.03):
    """point sets made of tight blobs (so ball query / CCL find real components)."""
    pts, batch = [], []
    for s in range(n_scenes):
        centers = rng.uniform(-0.8, 0.8, (n_clusters, 3))
        which = rng.integers(0, n_clusters, pts_per_scene)
        p = centers[which] + rng.normal(0, spread, (pts_per_scene, 3))
        pts.append(p.astype(np.float32))
        batch.append(np.full(pts_per_scene, s, np.int32))
    return np.concatenate(pts), np.concatenate(batch)
