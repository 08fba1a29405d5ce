#!/usr/bin/env python
"""CPU analysis: how evenly the masked-tile conv kernel's work (live taps per 16-row tile) falls on XCDs / CUs / SIMDs for the
bench's level shapes, in the rulebook's tile order (rows sorted by neighbour mask inside 16384-row blocks) and the kernel's
workgroup -> XCD mapping (contiguous eighths).  No GPU needed."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd.smoke import make_batch

def level_coords(n_scenes=8, n_points=20000, voxel=0.01):
    scenes = make_batch(n_scenes, n_points)
    out = []
    for b, pc in enumerate(scenes):
        xyz = pc.points[:, :3].numpy() if hasattr(pc.points, "numpy") else np.asarray(pc.points)[:, :3]
        v = np.floor((xyz - xyz.min(0)) / voxel).astype(np.int64)
        v = np.unique(v, axis=0)
        out.append(np.concatenate([np.full((len(v), 1), b), v + 1], 1))
    return np.concatenate(out)

def key(c):
    return ((c[:, 0] * 4096 + c[:, 1]) * 4096 + c[:, 2]) * 4096 + c[:, 3]

def masks(c):
    k = key(c); order = np.argsort(k); c = c[order]; k = k[order]
    m = np.zeros(len(c), np.uint32)
    t = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                q = c.copy(); q[:, 1] += dx; q[:, 2] += dy; q[:, 3] += dz
                kq = key(q); pos = np.searchsorted(k, kq); pos[pos >= len(k)] = 0
                m |= ((k[pos] == kq).astype(np.uint32) << np.uint32(t)); t += 1
    return c, m

def analyse(name, m, block=16384, nt_groups=1):
    n = len(m)
    perm = np.concatenate([s + np.argsort(m[s:s + block], kind="stable") for s in range(0, n, block)])
    mp = m[perm]
    n_tiles = (n + 15) // 16
    pad = np.zeros(n_tiles * 16, np.uint32); pad[:n] = mp
    tile_mask = np.bitwise_or.reduce(pad.reshape(n_tiles, 16), axis=1)
    live = np.array([bin(int(x)).count("1") for x in tile_mask])
    units = np.repeat(live, nt_groups)  # one wave per (tile, col group)
    n_units = len(units)
    n_wg = (n_units + 3) // 4
    grid = (n_wg + 7) // 8 * 8
    per = grid // 8
    u = np.zeros(grid * 4); u[:n_units] = units
    wg_load = u.reshape(grid, 4)
    xcd = wg_load.reshape(8, per, 4)  # xcd x -> wgs x*per .. (x+1)*per
    xcd_sum = xcd.sum((1, 2))
    print(f"{name}: rows {n} tiles {n_tiles} waves {n_units} live taps/tile mean {live.mean():.2f} max {live.max()}  pairs/row {np.mean([bin(int(x)).count('1') for x in m[:20000]]):.2f}")
    print("   XCD load (sum of live taps) / mean:", np.round(xcd_sum / xcd_sum.mean(), 3))
    # CU level inside an XCD: WG j of the XCD -> CU j % 32 (round robin while all fit), SIMD = wave
    worst = 0
    for x in range(8):
        cu = np.zeros((32, 4))
        for j in range(per):
            cu[j % 32] += xcd[x, j]
        worst = max(worst, cu.max())
    mean_simd = u.sum() / 1024
    print(f"   SIMD load: mean {mean_simd:.1f} live taps, worst SIMD {worst:.0f} ({worst / mean_simd:.2f}x), longest wave {live.max()}")

if __name__ == "__main__":
    c0 = level_coords()
    c0, m0 = masks(c0)
    analyse("L0 16->16 (nt 1)", m0, nt_groups=1)
    c1 = np.unique(np.concatenate([c0[:, :1], (c0[:, 1:] ) // 2], 1), axis=0)
    c1, m1 = masks(c1)
    analyse("L1 32->32 (nt groups 1)", m1, nt_groups=1)
    c2 = np.unique(np.concatenate([c1[:, :1], c1[:, 1:] // 2], 1), axis=0)
    c2, m2 = masks(c2)
    analyse("L2 48->48", m2, nt_groups=1)


def live_per_tile(m, perm):
    mp = m[perm]; n = len(m); nt = (n + 15) // 16
    pad = np.zeros(nt * 16, np.uint32); pad[:n] = mp
    tm = np.bitwise_or.reduce(pad.reshape(nt, 16), axis=1)
    return np.array([bin(int(x)).count("1") for x in tm])


def compare_orders(name, m, block=16384):
    n = len(m)
    pc = np.array([bin(int(x)).count("1") for x in m], dtype=np.uint32)
    useful = pc.sum()
    def order(keyfn):
        return np.concatenate([s + np.argsort(keyfn(slice(s, s + block)), kind="stable") for s in range(0, n, block)])
    res = {}
    res["mask"] = live_per_tile(m, order(lambda sl: m[sl]))
    res["popcount desc, mask"] = live_per_tile(m, order(lambda sl: ((27 - pc[sl]).astype(np.uint64) << np.uint64(27)) | m[sl]))
    print(name, "useful pairs/row %.2f" % (useful / n))
    for k, v in res.items():
        print(f"   order by {k:22s}: live taps per tile mean {v.mean():.2f}  slots/pair {v.sum() * 16 / useful:.3f}")


if __name__ == "__main__" and os.environ.get("ORDERS"):
    compare_orders("L0", m0); compare_orders("L1", m1); compare_orders("L2", m2)
