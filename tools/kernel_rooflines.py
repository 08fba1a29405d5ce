"""Every kernel family of the hot path at the BASELINE config-3 shapes (8 scenes x 20k points, voxel 0.01), one line each:
time per call (events around back-to-back calls on the launch stream), ALGORITHMIC bytes per SURVEY.md §8(d), achieved GB/s
and its fraction of (a) the HBM peak of the guide (8 TB/s) and (b) the float4-copy bandwidth measured in this run; the conv
rows also give TFLOP/s against the fp32 MFMA peak.  Output committed as profiles/r01_kernel_rooflines.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gapartnet_amd import hip_ops as H
from gapartnet_amd.smoke import make_batch
from gapartnet_amd.structure.point_cloud import PointCloud

dev = torch.device("cuda:0")
HBM_PEAK, MFMA_PEAK = 8000.0, 157.3


def timeit(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
t = timeit(lambda: dst.copy_(src), iters=10)
COPY_BW = 2 * src.numel() * 4 / t / 1e3
del src, dst
print(f"measured float4 copy bandwidth: {COPY_BW:.0f} GB/s ({COPY_BW / HBM_PEAK:.2f} of the 8 TB/s peak)\n")
print("(/copy = fraction of the 1 GiB copy bandwidth; /same = fraction of what a float4 copy moving the SAME number of bytes\n"
      " achieves in this run - the bandwidth a kernel of this size can reach at all: launch + ramp are a fixed ~4 us)\n")
print(f"{'kernel family (row of SURVEY 8a)':58s} {'us':>8s} {'alg MB':>8s} {'GB/s':>7s} {'/peak':>6s} {'/copy':>6s} {'/same':>6s} {'TF':>6s} {'/mfma':>6s}")
_copy_cache = {}


def same_size_copy_us(nbytes):
    n = max(int(nbytes) // 8 // 4 * 4, 4)  # floats read = floats written = nbytes / 8
    if n not in _copy_cache:
        a, b = torch.empty(n, dtype=torch.float32, device=dev).normal_(), torch.empty(n, dtype=torch.float32, device=dev)
        _copy_cache[n] = timeit(lambda: b.copy_(a))
    return _copy_cache[n]


def row(name, us, nbytes, flops=0.0):
    gbs = nbytes / us / 1e3
    tf = flops / us / 1e6
    print(f"{name:58s} {us:8.1f} {nbytes / 1e6:8.2f} {gbs:7.0f} {gbs / HBM_PEAK:6.3f} {gbs / COPY_BW:6.3f} "
          f"{same_size_copy_us(nbytes) / us:6.3f} {tf:6.2f} {tf / MFMA_PEAK:6.3f}")


pcs = [pc.to(dev) for pc in make_batch(8, 20000)]
batch = PointCloud.collate(pcs, voxel_size=(0.01, 0.01, 0.01))
pts, N = batch.points, batch.points.shape[0]
idx0 = batch.voxel_tensor.indices
V0, shape0 = idx0.shape[0], list(batch.voxel_tensor.spatial_shape)
print(f"# {N} points -> {V0} voxels, grid {shape0}")

# V -- scene voxelisation (one host read of the voxel count inside the wrapper)
from gapartnet_amd.structure.point_cloud import voxelize_scenes
us = timeit(lambda: voxelize_scenes(pts[:, :3], pts, [20000] * 8, (0.01, 0.01, 0.01), pyramid_levels=6), iters=10)
row("V  voxelize 160k pts x 6 ch + 6 coarse-level counts (ONE host read)", us, 4 * N * 9 + 4 * V0 * 9 + 4 * N)

# K1 / K2 -- rulebooks
us = timeit(lambda: H.rulebook_subm3(idx0, shape0), iters=10)
rb0 = H.rulebook_subm3(idx0, shape0)
P0 = int(rb0.num_pairs.item())
row(f"K1 subm3 rulebook + tile order, L0 ({V0} rows, {P0} pairs)", us, 16 * V0 + 4 * 27 * V0 * 2 + 8 * P0)
us = timeit(lambda: H.rulebook_down(idx0, shape0, 8), iters=10)
idx1, shape1, rbd, rbu = H.rulebook_down(idx0, shape0, 8)
V1 = idx1.shape[0]
row(f"K2 down rulebook L0->L1 ({V1} rows; incl. 1 host read)", us, 16 * V0 + 16 * V1 + 4 * 8 * (V0 + V1) + 16 * V0)
rb1 = H.rulebook_subm3(idx1, shape1)
P1 = int(rb1.num_pairs.item())
idx2, shape2, _, _ = H.rulebook_down(idx1, shape1, 8)
rb2 = H.rulebook_subm3(idx2, shape2)
P2, V2 = int(rb2.num_pairs.item()), idx2.shape[0]


# C -- fused conv forward / wgrad at the three large levels
def conv_rows(tag, rb, n, pairs, c):
    x = torch.randn(n, c, device=dev)
    g = torch.randn(n, c, device=dev)
    w = torch.randn(27, c, c, device=dev) * 0.05
    nbytes = 4 * n * c * 2 + 8 * pairs + 4 * 27 * c * c
    flops = 2.0 * pairs * c * c
    row(f"C  conv fwd {tag} ({n} rows, {c}->{c}, {pairs / n:.1f} pairs/row)", timeit(lambda: H.conv_fwd(x, w, rb)), nbytes, flops)
    row(f"C  conv wgrad {tag}", timeit(lambda: H.conv_wgrad(x, g, rb)), nbytes, flops)


conv_rows("L0", rb0, V0, P0, 16)
conv_rows("L1", rb1, V1, P1, 32)
conv_rows("L2", rb2, V2, P2, 48)

# BatchNorm (+ReLU) training forward / backward at L0
x = torch.randn(V0, 16, device=dev)
w, b = torch.ones(16, device=dev), torch.zeros(16, device=dev)
rm, rv = torch.zeros(16, device=dev), torch.ones(16, device=dev)
us = timeit(lambda: H.bn_fwd(x, None, w, b, rm, rv, True, 0.1, 1e-4, True))
row("BN forward (stats + apply + ReLU), L0 x 16 ch", us, 4 * V0 * 16 * 3)
y, mean, invstd = H.bn_fwd(x, None, w, b, rm, rv, True, 0.1, 1e-4, True)[:3]
dy = torch.randn_like(x)
us = timeit(lambda: H.bn_bwd(x, y, dy, w, mean, invstd, True, True, False))
row("BN backward (reduce + dx), L0 x 16 ch", us, 4 * V0 * 16 * 7)

# G -- voxel -> point gather and its transpose
feat = torch.randn(V0, 16, device=dev)
pid = batch.pc_voxel_id
row("G  gather rows voxels->points (160k x 16 ch)", timeit(lambda: H.gather_rows(feat, pid)), 4 * N + 64 * V0 + 64 * N)
csr = batch.pc_voxel_csr
dpt = torch.randn(N, 16, device=dev)
row("G' scatter rows points->voxels (CSR, ordered)", timeit(lambda: H.scatter_rows(dpt, pid, V0, csr=csr)), 8 * N + 64 * N + 64 * V0)

# the same two kernels at 4x and 16x the rows (the bench's batch of 32 and a 128-scene batch): the index -> row chain is one more
# memory round trip than a copy has, a fixed ~1 us that the small case cannot hide
for mult in (4, 16):
    Vm, Nm = V0 * mult, N * mult
    featm = torch.randn(Vm, 16, device=dev)
    pidm = torch.cat([pid + i * V0 for i in range(mult)]).contiguous()
    row(f"G  gather rows, {mult}x ({Nm} x 16 ch)", timeit(lambda: H.gather_rows(featm, pidm)), 4 * Nm + 64 * Vm + 64 * Nm)
    csrm = H.rows_csr(pidm, Vm)
    dptm = torch.randn(Nm, 16, device=dev)
    row(f"G' scatter rows, {mult}x (CSR, ordered)", timeit(lambda: H.scatter_rows(dptm, pidm, Vm, csr=csrm)), 8 * Nm + 64 * Nm + 64 * Vm)
    del featm, pidm, csrm, dptm

# B / L -- ball query on the foreground points (semantic label > 0), CCL on its result
sem = batch.sem_labels
fg = torch.nonzero(sem > 0).squeeze(1)
q = pts[fg, :3].contiguous()
bi = batch.batch_indices[fg].contiguous()
bo = torch.searchsorted(bi.long(), torch.arange(9, device=dev)).to(torch.int32)
lab = sem[fg].to(torch.int32)
Q = q.shape[0]
for K in (50, 300):
    us = timeit(lambda: H.ball_query(q, q, bi, bo, 0.04, K, lab, lab), iters=10)
    row(f"B  ball query r=0.04 K={K} ({Q} queries, grid)", us, 12 * Q * 2 + 8 * Q + 4 * Q * K + 4 * Q)
nbr_idx, cnt = H.ball_query(q, q, bi, bo, 0.04, 50, lab, lab)
begin = torch.arange(Q, device=dev, dtype=torch.int32) * 50
be = torch.stack([begin, begin + cnt], 1).reshape(-1).contiguous()
edges = nbr_idx.reshape(-1)
row(f"L  CCL ({Q} vertices, {int(cnt.sum())} edges)", timeit(lambda: H.ccl(be, edges), iters=10), 4 * (Q * 50 + 2 * Q) + 4 * Q)

# R -- segmented reduce / max-pool over proposals (400 segments over the foreground points)
P = 400
cuts = torch.sort(torch.randint(0, Q, (P - 1,), device=dev))[0].to(torch.int32)
beg = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), cuts])
end = torch.cat([cuts, torch.full((1,), Q, dtype=torch.int32, device=dev)])
vals = torch.randn(Q, 16, device=dev)
row("R  segmented max-pool forward (16 ch, 400 segments)", timeit(lambda: H.segmented_maxpool_fwd(vals, beg, end)), 64 * Q + 128 * P)
row("R  segmented reduce sum (3 ch)", timeit(lambda: H.segmented_reduce(q, beg, end, "sum")), 12 * Q + 12 * P)
