// voxelize.hip — kernel V: point cloud -> unique voxels, mean features, point->voxel map.
// Replaces epic_ops.voxelize (dataset/gapartnet.py:188-195, network/grouping_utils.py:93-101).
//
// Design (MI355X): the whole op is HBM-streaming integer work.  Points are keyed by a 64-bit linear
// (segment,x,y,z) key, ordered by ONE stable LSD radix sort (rocPRIM device primitive, key_bits wide),
// and voxel ids fall out of a boundary-flag scan.  Because the sort is stable, each voxel's points are
// contiguous and in ascending point order, so the mean is a short ordered fp32 sum per (voxel,channel):
// bit-identical to the CPU oracle and run-to-run deterministic (no fp atomics).  Voxels come out in
// ascending key order, which is also the row order that gives the sparse-conv gathers their locality.
#include "gpn_common.h"  // first: pulls <cstring> ahead of the HIP/rocPRIM headers

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace {

constexpr int kThreads = 256;

__global__ void vox_keys_kernel(const float* __restrict__ points, const int64_t* __restrict__ seg_offsets,
                                const float* __restrict__ rmin, const float* __restrict__ rmax, int64_t M,
                                int64_t S, float vs0, float vs1, float vs2, int d0, int d1, int d2,
                                uint64_t invalid_key, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  // segment of point i: last s with seg_offsets[s] <= i
  int64_t lo = 0, hi = S - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi + 1) >> 1;
    if (seg_offsets[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const int64_t s = lo;
  const float vs[3] = {vs0, vs1, vs2};
  const int dims[3] = {d0, d1, d2};
  int c[3];
  bool ok = i >= seg_offsets[0] && i < seg_offsets[S];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p = points[i * 3 + a];
    float mn = rmin[s * 3 + a], mx = rmax[s * 3 + a];
    ok = ok && (p >= mn) && (p < mx);
    float q = __fdiv_rn(__fsub_rn(p, mn), vs[a]);
    int ci = (int)floorf(q);
    ok = ok && ci >= 0 && ci < dims[a];
    c[a] = ci;
  }
  uint64_t key = invalid_key;
  if (ok) key = (((uint64_t)s * (uint64_t)d0 + (uint64_t)c[0]) * (uint64_t)d1 + (uint64_t)c[1]) * (uint64_t)d2 + (uint64_t)c[2];
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

__global__ void vox_flags_kernel(const uint64_t* __restrict__ ks, int64_t M, uint64_t invalid_key,
                                 int32_t* __restrict__ flags) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  uint64_t k = ks[j];
  flags[j] = (k != invalid_key && (j == 0 || ks[j - 1] != k)) ? 1 : 0;
}

__global__ void vox_emit_kernel(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ order,
                                const int32_t* __restrict__ incl, int64_t M, uint64_t invalid_key, int d0,
                                int d1, int d2, int32_t* __restrict__ voxel_coords,
                                int32_t* __restrict__ voxel_seg, int32_t* __restrict__ pc_voxel_id,
                                int32_t* __restrict__ vstart, int64_t* __restrict__ num_voxels,
                                int32_t* __restrict__ point_order) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  uint64_t k = ks[j];
  int32_t vid = incl[j] - 1;
  if (point_order) point_order[j] = (int32_t)order[j];
  if (j == M - 1) {
    num_voxels[0] = incl[j];
    if (k != invalid_key) vstart[incl[j]] = (int32_t)M;  // no dropped points: last voxel ends at M
  }
  if (k == invalid_key) {
    pc_voxel_id[order[j]] = -1;
    // first invalid entry closes the last voxel
    if (j == 0 || ks[j - 1] != invalid_key) vstart[incl[j]] = (int32_t)j;
    return;
  }
  pc_voxel_id[order[j]] = vid;
  if (j == 0 || ks[j - 1] != k) {
    vstart[vid] = (int32_t)j;
    uint64_t r = k;
    voxel_coords[(int64_t)vid * 3 + 2] = (int32_t)(r % (uint64_t)d2); r /= (uint64_t)d2;
    voxel_coords[(int64_t)vid * 3 + 1] = (int32_t)(r % (uint64_t)d1); r /= (uint64_t)d1;
    voxel_coords[(int64_t)vid * 3 + 0] = (int32_t)(r % (uint64_t)d0); r /= (uint64_t)d0;
    voxel_seg[vid] = (int32_t)r;
  }
}

// one thread per (voxel, channel): ordered fp32 sum over the voxel's points, then / count
__global__ void vox_mean_kernel(const float* __restrict__ feats, const uint32_t* __restrict__ order,
                                const int32_t* __restrict__ vstart, const int64_t* __restrict__ num_voxels,
                                int64_t M, int C, float* __restrict__ voxel_feats) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t v = t / C;
  int c = (int)(t % C);
  if (v >= num_voxels[0]) return;
  int32_t b = vstart[v], e = vstart[v + 1];
  float acc = 0.f;
  for (int32_t j = b; j < e; ++j) acc = __fadd_rn(acc, feats[(int64_t)order[j] * C + c]);
  voxel_feats[v * C + c] = __fdiv_rn(acc, (float)(e - b));
}

// ---- scene batches without a host read before the launch sequence (gpn_voxelize_scenes) -----------------------------------
// per-scene range [min - 1e-4, max + 1e-4] (dataset/gapartnet.py:186-187): one workgroup per scene, fixed-order reduction
__global__ __launch_bounds__(256) void vox_scene_range_kernel(const float* __restrict__ points, const int64_t* __restrict__ seg_offsets,
                                                              float* __restrict__ rmin, float* __restrict__ rmax) {
  __shared__ float lo[256][3], hi[256][3];
  const int s = blockIdx.x, t = threadIdx.x;
  const int64_t b = seg_offsets[s], e = seg_offsets[s + 1];
  float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = b + t; i < e; i += 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float p = points[i * 3 + a];
      l[a] = fminf(l[a], p);
      h[a] = fmaxf(h[a], p);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) lo[t][a] = l[a], hi[t][a] = h[a];
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) {
#pragma unroll
      for (int a = 0; a < 3; ++a) lo[t][a] = fminf(lo[t][a], lo[t + off][a]), hi[t][a] = fmaxf(hi[t][a], hi[t + off][a]);
    }
    __syncthreads();
  }
  if (t < 3) {
    rmin[s * 3 + t] = __fsub_rn(lo[0][t], 1e-4f);
    rmax[s * 3 + t] = __fadd_rn(hi[0][t], 1e-4f);
  }
}

// packed key (segment << 30 | x << 20 | y << 10 | z): the same (segment, x, y, z) order as the linear key of
// vox_keys_kernel without knowing the grid extent (cells per axis < 1024; a larger cell index raises stats[5])
constexpr int kPackBits = 10;
__global__ void vox_keys_packed_kernel(const float* __restrict__ points, const int64_t* __restrict__ seg_offsets,
                                       const float* __restrict__ rmin, const float* __restrict__ rmax, int64_t M, int64_t S,
                                       float vs0, float vs1, float vs2, uint64_t invalid_key, uint64_t* __restrict__ keys,
                                       uint32_t* __restrict__ vals, int64_t* __restrict__ stats) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  int64_t lo = 0, hi = S - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi + 1) >> 1;
    if (seg_offsets[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const int64_t s = lo;
  const float vs[3] = {vs0, vs1, vs2};
  int c[3];
  bool ok = i >= seg_offsets[0] && i < seg_offsets[S];
  bool overflow = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p = points[i * 3 + a];
    float mn = rmin[s * 3 + a], mx = rmax[s * 3 + a];
    ok = ok && (p >= mn) && (p < mx);
    float q = __fdiv_rn(__fsub_rn(p, mn), vs[a]);
    int ci = (int)floorf(q);
    ok = ok && ci >= 0;
    overflow = overflow || (ok && ci >= (1 << kPackBits));
    c[a] = ci;
  }
  if (overflow) {
    atomicMax(reinterpret_cast<unsigned long long*>(stats + 5), 1ull);
    ok = false;
  }
  uint64_t key = invalid_key;
  if (ok) key = ((((uint64_t)s << kPackBits | (uint64_t)c[0]) << kPackBits | (uint64_t)c[1]) << kPackBits) | (uint64_t)c[2];
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// as vox_emit_kernel for packed keys; writes indices [V,4] = (segment, x, y, z) directly and the batch statistics the host
// reads once: stats[0] = #voxels, [1..3] = largest cell index per axis, [4] = dropped points
__global__ __launch_bounds__(256) void vox_emit_packed_kernel(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ order,
                                                              const int32_t* __restrict__ incl, int64_t M, uint64_t invalid_key,
                                                              int32_t* __restrict__ indices4, int32_t* __restrict__ pc_voxel_id,
                                                              int32_t* __restrict__ vstart, int64_t* __restrict__ stats,
                                                              int32_t* __restrict__ point_order) {
  __shared__ int wg_max[3];
  if (threadIdx.x < 3) wg_max[threadIdx.x] = -1;
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int x = -1, y = -1, z = -1;  // cell of the voxel this entry starts (-1: none)
  if (j < M) {
    const uint64_t k = ks[j];
    const int32_t vid = incl[j] - 1;
    if (point_order) point_order[j] = (int32_t)order[j];
    if (j == M - 1) {
      stats[0] = incl[j];
      if (k != invalid_key) vstart[incl[j]] = (int32_t)M;
    }
    if (k == invalid_key) {
      pc_voxel_id[order[j]] = -1;
      if (j == 0 || ks[j - 1] != invalid_key) {
        vstart[incl[j]] = (int32_t)j;
        stats[4] = M - j;  // every entry from here on is a dropped point (invalid keys sort last)
      }
    } else {
      pc_voxel_id[order[j]] = vid;
      if (j == 0 || ks[j - 1] != k) {
        vstart[vid] = (int32_t)j;
        const int mask = (1 << kPackBits) - 1;
        z = (int)(k & mask), y = (int)((k >> kPackBits) & mask), x = (int)((k >> (2 * kPackBits)) & mask);
        reinterpret_cast<int4*>(indices4)[vid] = make_int4((int)(k >> (3 * kPackBits)), x, y, z);
      }
    }
  }
  // largest cell index per axis: wave maximum (shuffles), workgroup maximum (LDS), ONE atomic per workgroup and axis - an
  // atomic per voxel was 430k atomics on three addresses, 90 us of this kernel's 95
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    x = max(x, __shfl_xor(x, off, 64));
    y = max(y, __shfl_xor(y, off, 64));
    z = max(z, __shfl_xor(z, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    if (x >= 0) atomicMax(&wg_max[0], x);
    if (y >= 0) atomicMax(&wg_max[1], y);
    if (z >= 0) atomicMax(&wg_max[2], z);
  }
  __syncthreads();
  if (threadIdx.x < 3 && wg_max[threadIdx.x] >= 0)
    atomicMax(reinterpret_cast<unsigned long long*>(stats + 1 + threadIdx.x), (unsigned long long)wg_max[threadIdx.x]);
}

size_t sort_temp_bytes(int64_t M) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                            (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)(M > 0 ? M : 1), 0u, 64u,
                            (hipStream_t) nullptr);
  return bytes;
}
size_t scan_temp_bytes(int64_t M) {
  size_t bytes = 0;
  rocprim::inclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                          (size_t)(M > 0 ? M : 1), rocprim::plus<int32_t>(), (hipStream_t) nullptr);
  return bytes;
}

struct VoxWs {
  uint64_t *keys, *keys_sorted;
  uint32_t *vals, *order;
  int32_t *flags, *incl, *vstart;
  void* prim_tmp;
  size_t prim_bytes;
};

bool carve(gpn::WsCarver& w, int64_t M, VoxWs& o) {
  size_t m = (size_t)(M > 0 ? M : 1);
  o.keys = w.take<uint64_t>(m);
  o.keys_sorted = w.take<uint64_t>(m);
  o.vals = w.take<uint32_t>(m);
  o.order = w.take<uint32_t>(m);
  o.flags = w.take<int32_t>(m);
  o.incl = w.take<int32_t>(m);
  o.vstart = w.take<int32_t>(m + 1);
  size_t a = sort_temp_bytes(M), b = scan_temp_bytes(M);
  o.prim_bytes = a > b ? a : b;
  o.prim_tmp = w.take<char>(o.prim_bytes);
  return w.ok();
}

}  // namespace

extern "C" size_t gpn_voxelize_ws_bytes(int64_t M, int C) {
  (void)C;
  gpn::WsCarver w(nullptr, 0);
  VoxWs o;
  carve(w, M, o);
  return w.used;
}

// extended entry: also returns the point order grouped by voxel and each voxel's start in that order
extern "C" int gpn_voxelize_ex(const float* points, const float* feats, const int64_t* seg_offsets,
                               const float* seg_range_min, const float* seg_range_max, int64_t M, int C,
                               int64_t S, const float* voxel_size_host, const int32_t* grid_dims_host,
                               float* voxel_feats, int32_t* voxel_coords, int32_t* voxel_seg,
                               int32_t* pc_voxel_id, int64_t* num_voxels, int32_t* point_order,
                               int32_t* voxel_point_start, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(M >= 0 && C >= 1 && S >= 1);
  GPN_CHECK_ARG(voxel_size_host && grid_dims_host && num_voxels);
  if (M == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(num_voxels, 0, sizeof(int64_t), stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(points && feats && seg_offsets && seg_range_min && seg_range_max);
  GPN_CHECK_ARG(voxel_feats && voxel_coords && voxel_seg && pc_voxel_id);
  GPN_CHECK_ARG(M < (int64_t)0x7fffffff);
  const int d0 = grid_dims_host[0], d1 = grid_dims_host[1], d2 = grid_dims_host[2];
  GPN_CHECK_ARG(d0 > 0 && d1 > 0 && d2 > 0);
  // key space must fit 63 bits
  long double total = (long double)S * d0 * d1 * d2;
  GPN_CHECK_ARG(total < 9.0e18L);
  const uint64_t invalid_key = (uint64_t)S * (uint64_t)d0 * (uint64_t)d1 * (uint64_t)d2;
  unsigned key_bits = 1;
  while (key_bits < 64 && (invalid_key >> key_bits) != 0) ++key_bits;

  gpn::WsCarver w(ws, ws_bytes);
  VoxWs o;
  carve(w, M, o);
  GPN_CHECK_WS(w);
  if (voxel_point_start) o.vstart = voxel_point_start;  // caller keeps the CSR (capacity M+1)

  const int grid = (int)gpn::cdiv(M, kThreads);
  gpn::ProfScope prof(GPN_K_VOXELIZE, stream, 0.0,
                      4.0 * (double)M * (3 + C) + 4.0 * (double)M * (3 + C) + 4.0 * (double)M);
  hipLaunchKernelGGL(vox_keys_kernel, dim3(grid), dim3(kThreads), 0, stream, points, seg_offsets,
                     seg_range_min, seg_range_max, M, S, voxel_size_host[0], voxel_size_host[1],
                     voxel_size_host[2], d0, d1, d2, invalid_key, o.keys, o.vals);
  GPN_CHECK_LAUNCH();
  size_t tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::radix_sort_pairs(o.prim_tmp, tmp, o.keys, o.keys_sorted, o.vals, o.order, (size_t)M,
                                          0u, key_bits, stream));
  hipLaunchKernelGGL(vox_flags_kernel, dim3(grid), dim3(kThreads), 0, stream, o.keys_sorted, M, invalid_key,
                     o.flags);
  GPN_CHECK_LAUNCH();
  tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::inclusive_scan(o.prim_tmp, tmp, o.flags, o.incl, (size_t)M,
                                        rocprim::plus<int32_t>(), stream));
  hipLaunchKernelGGL(vox_emit_kernel, dim3(grid), dim3(kThreads), 0, stream, o.keys_sorted, o.order, o.incl,
                     M, invalid_key, d0, d1, d2, voxel_coords, voxel_seg, pc_voxel_id, o.vstart, num_voxels,
                     point_order);
  GPN_CHECK_LAUNCH();
  const int64_t mc = M * C;
  hipLaunchKernelGGL(vox_mean_kernel, dim3((int)gpn::cdiv(mc, kThreads)), dim3(kThreads), 0, stream, feats,
                     o.order, o.vstart, num_voxels, M, C, voxel_feats);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_voxelize(const float* points, const float* feats, const int64_t* seg_offsets,
                            const float* seg_range_min, const float* seg_range_max, int64_t M, int C,
                            int64_t S, const float* voxel_size_host, const int32_t* grid_dims_host,
                            float* voxel_feats, int32_t* voxel_coords, int32_t* voxel_seg,
                            int32_t* pc_voxel_id, int64_t* num_voxels, void* ws, size_t ws_bytes,
                            gpn_stream_t stream) {
  return gpn_voxelize_ex(points, feats, seg_offsets, seg_range_min, seg_range_max, M, C, S, voxel_size_host,
                         grid_dims_host, voxel_feats, voxel_coords, voxel_seg, pc_voxel_id, num_voxels,
                         nullptr, nullptr, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Scene batches, the reference's per-scene conventions (dataset/gapartnet.py:179-205: range = [min - 1e-4, max + 1e-4] per
// scene, coordinates from the scene's own minimum), WITHOUT a host read before or between the launches: the per-scene range
// is reduced on the device, keys are packed with 10 bits per axis instead of linearised with the (data-dependent) grid
// extent, and everything the host needs afterwards comes back in ONE read of `stats`:
//   stats[0] #voxels, [1..3] largest cell index per axis (spatial extent = max(that + 1, 128)), [4] dropped points,
//   [5] != 0: a cell index >= 1024 occurred - results are incomplete, use gpn_voxelize_ex with the true grid extent,
//   [8 .. 8 + n_levels): rows of the n_levels stride-2 levels below the voxel set (what gpn_rulebook_level_counts reports).
// Outputs as gpn_voxelize_ex, with indices4 [M,4] = (segment, x, y, z) instead of separate coordinate / segment arrays.
// Same voxel order (ascending (segment, x, y, z)) and bit-identical ordered means.
namespace {
size_t bitmap_path_ws_bytes(int64_t M, int64_t S);
}
extern "C" size_t gpn_voxelize_scenes_ws_bytes(int64_t M, int C, int64_t S, int n_levels) {
  const size_t sorted = gpn_voxelize_ws_bytes(M, C) + gpn::align_up((size_t)(S > 0 ? S : 1) * 6 * sizeof(float));
  return std::max(sorted, bitmap_path_ws_bytes(M, S)) + (n_levels > 0 ? gpn_rulebook_level_counts_ws_bytes(M, n_levels) : 0);
}

extern "C" int gpn_voxelize_scenes_sorted(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C,
                                          int64_t S, const float* voxel_size_host, int n_levels, float* voxel_feats,
                                          int32_t* indices4, int32_t* pc_voxel_id, int32_t* point_order,
                                          int32_t* voxel_point_start, int64_t* stats, void* ws, size_t ws_bytes,
                                          gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(M >= 0 && C >= 1 && S >= 1 && S < (1 << 20) && n_levels >= 0 && n_levels <= 16 && voxel_size_host && stats);
  GPN_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(int64_t) * (size_t)(8 + n_levels), stream));
  if (M == 0) return GPN_OK;
  GPN_CHECK_ARG(points && feats && seg_offsets && voxel_feats && indices4 && pc_voxel_id && M < (int64_t)0x7fffffff);
  gpn::WsCarver w(ws, ws_bytes);
  VoxWs o;
  carve(w, M, o);
  float* rmin = w.take<float>((size_t)S * 3);
  float* rmax = w.take<float>((size_t)S * 3);
  const size_t lc_bytes = n_levels > 0 ? gpn_rulebook_level_counts_ws_bytes(M, n_levels) : 0;
  void* lc_ws = w.take<char>(lc_bytes);
  GPN_CHECK_WS(w);
  if (voxel_point_start) o.vstart = voxel_point_start;
  const uint64_t invalid_key = (uint64_t)S << (3 * kPackBits);
  unsigned key_bits = 1;
  while (key_bits < 64 && (invalid_key >> key_bits) != 0) ++key_bits;
  const int grid = (int)gpn::cdiv(M, kThreads);
  {
    gpn::ProfScope prof(GPN_K_VOXELIZE, stream, 0.0, 4.0 * (double)M * (3 + C) + 4.0 * (double)M * (3 + C) + 4.0 * (double)M);
    hipLaunchKernelGGL(vox_scene_range_kernel, dim3((unsigned)S), dim3(256), 0, stream, points, seg_offsets, rmin, rmax);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_keys_packed_kernel, dim3(grid), dim3(kThreads), 0, stream, points, seg_offsets, rmin, rmax, M, S,
                       voxel_size_host[0], voxel_size_host[1], voxel_size_host[2], invalid_key, o.keys, o.vals, stats);
    GPN_CHECK_LAUNCH();
    size_t tmp = o.prim_bytes;
    GPN_CHECK_HIP(rocprim::radix_sort_pairs(o.prim_tmp, tmp, o.keys, o.keys_sorted, o.vals, o.order, (size_t)M, 0u, key_bits, stream));
    hipLaunchKernelGGL(vox_flags_kernel, dim3(grid), dim3(kThreads), 0, stream, o.keys_sorted, M, invalid_key, o.flags);
    GPN_CHECK_LAUNCH();
    tmp = o.prim_bytes;
    GPN_CHECK_HIP(rocprim::inclusive_scan(o.prim_tmp, tmp, o.flags, o.incl, (size_t)M, rocprim::plus<int32_t>(), stream));
    hipLaunchKernelGGL(vox_emit_packed_kernel, dim3(grid), dim3(kThreads), 0, stream, o.keys_sorted, o.order, o.incl, M,
                       invalid_key, indices4, pc_voxel_id, o.vstart, stats, point_order);
    GPN_CHECK_LAUNCH();
    const int64_t mc = M * C;
    hipLaunchKernelGGL(vox_mean_kernel, dim3((int)gpn::cdiv(mc, kThreads)), dim3(kThreads), 0, stream, feats, o.order, o.vstart,
                       stats /* [0] = #voxels */, M, C, voxel_feats);
    GPN_CHECK_LAUNCH();
  }
  if (n_levels > 0)
    return gpn::rulebook_level_counts_dev(indices4, M, stats, S, stats + 1, n_levels, stats + 8, lc_ws, lc_bytes, stream);
  return GPN_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// gpn_voxelize_scenes WITHOUT a sort (round 5; BASELINE.json's "hash-table voxelization").  The sort only
// served to (a) number the occupied cells in ascending (segment, x, y, z) order and (b) group the points by cell in ascending
// point order.  (a) is a popcount rank in an occupancy bitmap laid out in key order - cell (s, x, y, z) is bit
// ((s DX + x) DY + y) DZ + z, with DX / DY / DZ the batch's cells per axis, reduced on the device from the scene ranges - the
// trick the stride-2 rulebook already uses (rulebook.hip).  (b) is a counting placement (voxel sizes by atomics = exact integers,
// an exclusive scan, atomic cursors) followed by an ascending sort of every voxel's own handful of points (a thread per voxel:
// 1.1 points per voxel on the bench's scenes).  Outputs are bit-identical to the sorting form (tests/test_gpu_ops.py), which
// stays as gpn_voxelize_scenes_sorted and is what the caller falls back to when stats[5] != 0 (a cell index >= 1024, as before,
// or a batch whose grid S x DX x DY x DZ exceeds the bitmap: 2^27 cells).
namespace {

constexpr int64_t kBitmapWords = (int64_t)1 << 22;  // 2^27 cells (8 scenes of 256^3, 32 of 160^3) = 16 MiB
constexpr int kScanBlock = 1024;                    // words (or counters) per workgroup of the three-launch scans

struct GridInfo {  // device-side description of the batch's cell grid: [0..2] cells per axis, [3] words in use (0 = does not fit)
  unsigned int d[4];
};

__global__ __launch_bounds__(256) void voxb_clear_kernel(const GridInfo* __restrict__ g, uint32_t* __restrict__ bitmap) {
  const unsigned words = g->d[3];
  for (unsigned w = blockIdx.x * 256u + threadIdx.x; w < words; w += gridDim.x * 256u) bitmap[w] = 0u;
}

// cell of every point (arithmetic of vox_keys_packed_kernel) -> its bit; cell_of[i] = linear cell index, 0xffffffff = dropped
__global__ __launch_bounds__(256) void voxb_mark_kernel(const float* __restrict__ points, const int64_t* __restrict__ seg_offsets,
                                                        const float* __restrict__ rmin, const float* __restrict__ rmax, int64_t M,
                                                        int64_t S, float vs0, float vs1, float vs2, const GridInfo* __restrict__ g,
                                                        uint32_t* __restrict__ bitmap, uint32_t* __restrict__ cell_of,
                                                        int64_t* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  if (g->d[3] == 0) return;
  int64_t lo = 0, hi = S - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi + 1) >> 1;
    if (seg_offsets[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const int64_t s = lo;
  const float vs[3] = {vs0, vs1, vs2};
  int c[3];
  bool ok = i >= seg_offsets[0] && i < seg_offsets[S];
  bool overflow = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p = points[i * 3 + a];
    float mn = rmin[s * 3 + a], mx = rmax[s * 3 + a];
    ok = ok && (p >= mn) && (p < mx);
    float q = __fdiv_rn(__fsub_rn(p, mn), vs[a]);
    int ci = (int)floorf(q);
    ok = ok && ci >= 0;
    overflow = overflow || (ok && ci >= (1 << kPackBits));
    c[a] = ci;
  }
  if (overflow) {  // (the packed-key form's contract: such a batch is the caller's general path)
    atomicMax(reinterpret_cast<unsigned long long*>(stats + 5), 1ull);
    ok = false;
  }
  uint32_t lin = 0xffffffffu;
  if (ok) {
    lin = (((uint32_t)s * g->d[0] + (uint32_t)c[0]) * g->d[1] + (uint32_t)c[1]) * g->d[2] + (uint32_t)c[2];
    atomicOr(&bitmap[lin >> 5], 1u << (lin & 31));
  }
  cell_of[i] = lin;
}

// ---- exclusive scan of n 32-bit values in three launches (n on the device: *n_dev words / counters are live) ---------------
// A: per workgroup of kScanBlock values, their sum.  POP: the values are bitmap words, summed as popcounts.
template <bool POP>
__global__ __launch_bounds__(256) void voxb_scan_sums_kernel(const uint32_t* __restrict__ v, const unsigned* __restrict__ n_dev,
                                                             int64_t n_max, uint32_t* __restrict__ block_sum) {
  __shared__ uint32_t part[256];
  const int64_t n = n_dev ? (int64_t)*n_dev : n_max;
  const int64_t base = (int64_t)blockIdx.x * kScanBlock;
  uint32_t acc = 0;
  for (int k = threadIdx.x; k < kScanBlock; k += 256) {
    const int64_t w = base + k;
    if (w < n) acc += POP ? (uint32_t)__popc(v[w]) : v[w];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sum[blockIdx.x] = base < n ? part[0] : 0u;
}
// B: one workgroup scans the block sums in place (exclusive) and leaves the total in *total_out
__global__ __launch_bounds__(1024) void voxb_scan_blocks_kernel(uint32_t* __restrict__ block_sum, int64_t n_blocks,
                                                                const unsigned* __restrict__ n_dev, int64_t* __restrict__ total_out) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  if (n_dev) {  // (only the blocks that hold live values: the others' sums are zero and their bases are never read)
    const int64_t live = ((int64_t)*n_dev + kScanBlock - 1) / kScanBlock;
    n_blocks = live < n_blocks ? live : n_blocks;
  }
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
    const int64_t b = b0 + threadIdx.x;
    const uint32_t x = b < n_blocks ? block_sum[b] : 0u;
    part[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      const uint32_t t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (b < n_blocks) block_sum[b] = carry + part[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = (int64_t)carry;
}
// C: exclusive prefix of every value (block base + scan inside the block)
template <bool POP>
__global__ __launch_bounds__(256) void voxb_scan_apply_kernel(const uint32_t* __restrict__ v, const unsigned* __restrict__ n_dev,
                                                              int64_t n_max, const uint32_t* __restrict__ block_base,
                                                              uint32_t* __restrict__ prefix, int32_t* __restrict__ tail_out) {
  __shared__ uint32_t part[256];
  const int64_t n = n_dev ? (int64_t)*n_dev : n_max;
  const int64_t base = (int64_t)blockIdx.x * kScanBlock;
  if (base >= n) return;
  constexpr int PER = kScanBlock / 256;
  uint32_t x[PER], acc = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t w = base + (int64_t)threadIdx.x * PER + k;
    x[k] = w < n ? (POP ? (uint32_t)__popc(v[w]) : v[w]) : 0u;
    acc += x[k];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = block_base[blockIdx.x] + part[threadIdx.x] - acc;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t w = base + (int64_t)threadIdx.x * PER + k;
    if (w < n) prefix[w] = run;
    run += x[k];
    if (tail_out && w == n - 1) tail_out[n] = (int32_t)run;  // (the closing entry of a CSR: prefix[n] = the total)
  }
}

// voxel of every point = rank of its cell's bit; voxel sizes; the voxel's row of indices4; batch statistics
__global__ __launch_bounds__(256) void voxb_rank_kernel(const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ bitmap,
                                                        const uint32_t* __restrict__ word_prefix, const GridInfo* __restrict__ g,
                                                        int64_t M, int32_t* __restrict__ pc_voxel_id, uint32_t* __restrict__ cnt,
                                                        int32_t* __restrict__ indices4, int64_t* __restrict__ stats) {
  __shared__ int wg_max[3];
  __shared__ int wg_dropped;
  if (threadIdx.x < 3) wg_max[threadIdx.x] = -1;
  if (threadIdx.x == 3) wg_dropped = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int x = -1, y = -1, z = -1, dropped = 0;
  if (i < M && g->d[3] == 0) pc_voxel_id[i] = -1;  // (the grid did not fit - stats[5] = 2: nothing is placed, the caller takes another path)
  if (i < M && g->d[3] != 0) {
    const uint32_t lin = cell_of[i];
    if (lin == 0xffffffffu) {
      pc_voxel_id[i] = -1;
      dropped = 1;
    } else {
      const uint32_t w = lin >> 5, bit = lin & 31;
      const int32_t vid = (int32_t)(word_prefix[w] + (uint32_t)__popc(bitmap[w] & ((1u << bit) - 1u)));
      pc_voxel_id[i] = vid;
      atomicAdd(&cnt[vid], 1u);
      uint32_t r = lin;
      z = (int)(r % g->d[2]), r /= g->d[2];
      y = (int)(r % g->d[1]), r /= g->d[1];
      x = (int)(r % g->d[0]), r /= g->d[0];
      reinterpret_cast<int4*>(indices4)[vid] = make_int4((int)r, x, y, z);  // (every point of the voxel writes the same row)
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    x = max(x, __shfl_xor(x, off, 64));
    y = max(y, __shfl_xor(y, off, 64));
    z = max(z, __shfl_xor(z, off, 64));
    dropped += __shfl_xor(dropped, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (x >= 0) atomicMax(&wg_max[0], x);
    if (y >= 0) atomicMax(&wg_max[1], y);
    if (z >= 0) atomicMax(&wg_max[2], z);
    if (dropped) atomicAdd(&wg_dropped, dropped);
  }
  __syncthreads();
  if (threadIdx.x < 3 && wg_max[threadIdx.x] >= 0)
    atomicMax(reinterpret_cast<unsigned long long*>(stats + 1 + threadIdx.x), (unsigned long long)wg_max[threadIdx.x]);
  if (threadIdx.x == 3 && wg_dropped) atomicAdd(reinterpret_cast<unsigned long long*>(stats + 4), (unsigned long long)wg_dropped);
}

// points into their voxel's stretch of point_order (any order inside a voxel; voxb_sort_kernel fixes it); cnt counts down
__global__ __launch_bounds__(256) void voxb_place_kernel(const int32_t* __restrict__ pc_voxel_id, const int32_t* __restrict__ vstart,
                                                         uint32_t* __restrict__ cnt, int64_t M, int32_t* __restrict__ point_order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int32_t v = pc_voxel_id[i];
  if (v < 0) return;
  const uint32_t left = atomicSub(&cnt[v], 1u);
  point_order[vstart[v] + (int32_t)left - 1] = (int32_t)i;
}

// the dropped points behind all others, in ascending point order (the stable sort put the invalid keys last): nothing to do for a
// batch without dropped points (the usual case: one load and out); otherwise ONE workgroup walks the points in order
__global__ __launch_bounds__(256) void voxb_dropped_kernel(const int32_t* __restrict__ pc_voxel_id, int64_t M,
                                                           const int64_t* __restrict__ stats, int32_t* __restrict__ point_order) {
  const int64_t dropped = stats[4];
  if (dropped == 0) return;
  __shared__ int wave_cnt[4];
  int64_t base = M - dropped;
  for (int64_t i0 = 0; i0 < M; i0 += 256) {
    const int64_t i = i0 + threadIdx.x;
    const bool is = i < M && pc_voxel_id[i] < 0;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(is);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int q = 0; q < wave; ++q) before += wave_cnt[q];
    if (is) point_order[base + before + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)i;
    base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// ascending point order inside every voxel (insertion sort of its stretch: a voxel holds a handful of points)
__global__ __launch_bounds__(256) void voxb_sort_kernel(const int32_t* __restrict__ vstart, const int64_t* __restrict__ stats,
                                                        int32_t* __restrict__ point_order) {
  const int64_t V = stats[0];
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = vstart[v], e = vstart[v + 1];
    for (int32_t j = b + 1; j < e; ++j) {
      const int32_t key = point_order[j];
      int32_t k = j - 1;
      while (k >= b && point_order[k] > key) {
        point_order[k + 1] = point_order[k];
        --k;
      }
      point_order[k + 1] = key;
    }
  }
}

// per-scene coordinate range with kRangeSplit workgroups per scene (the one-workgroup-per-scene kernel above is a serial walk
// of 20 000 points by 256 threads: 27 us of a 100 us voxelisation): workgroup minima / maxima meet in ordered-integer atomics
// (min / max are exact: any order gives the same bits), the LAST workgroup to finish - a ticket - applies the reference's
// -1e-4 / +1e-4 margins and derives the batch's cell grid (voxb_extent_kernel's arithmetic).
constexpr int kRangeSplit = 32;
__device__ __forceinline__ int float_order(float f) {  // monotone map float -> int (for atomicMin / atomicMax)
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float order_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void voxb_range_init_kernel(int* __restrict__ lo_bits, int* __restrict__ hi_bits, unsigned* __restrict__ ticket, int64_t n) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) lo_bits[k] = 0x7f800000, hi_bits[k] = (int)(0xff800000u ^ 0x7fffffffu);
  if (k == n) *ticket = 0u;
}

__global__ __launch_bounds__(256) void voxb_range_kernel(const float* __restrict__ points, const int64_t* __restrict__ seg_offsets,
                                                         int64_t S, float vs0, float vs1, float vs2, int* __restrict__ lo_bits,
                                                         int* __restrict__ hi_bits, unsigned* __restrict__ ticket,
                                                         float* __restrict__ rmin, float* __restrict__ rmax,
                                                         GridInfo* __restrict__ g, int64_t* __restrict__ stats) {
  __shared__ float lo[256][3], hi[256][3];
  __shared__ unsigned s_last;
  const int s = blockIdx.x / kRangeSplit, part = blockIdx.x % kRangeSplit, t = threadIdx.x;
  const int64_t b = seg_offsets[s], e = seg_offsets[s + 1];
  float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = b + (int64_t)part * 256 + t; i < e; i += (int64_t)kRangeSplit * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float p = points[i * 3 + a];
      l[a] = fminf(l[a], p);
      h[a] = fmaxf(h[a], p);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) lo[t][a] = l[a], hi[t][a] = h[a];
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) {
#pragma unroll
      for (int a = 0; a < 3; ++a) lo[t][a] = fminf(lo[t][a], lo[t + off][a]), hi[t][a] = fmaxf(hi[t][a], hi[t + off][a]);
    }
    __syncthreads();
  }
  if (t < 3) {
    // RETURNING atomics whose results are consumed: the wave cannot reach its ticket below before these read-modify-writes have
    // been performed (a returned value is the only completion signal a device-scope atomic gives its issuer)
    const int o1 = atomicMin(&lo_bits[s * 3 + t], float_order(lo[0][t]));
    const int o2 = atomicMax(&hi_bits[s * 3 + t], float_order(hi[0][t]));
    asm volatile("" ::"v"(o1), "v"(o2));
  }
  __syncthreads();
  if (t == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  for (int64_t k = t; k < S * 3; k += 256) {
    const float mn = order_float(__hip_atomic_load(&lo_bits[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float mx = order_float(__hip_atomic_load(&hi_bits[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    rmin[k] = __fsub_rn(mn, 1e-4f);
    rmax[k] = __fadd_rn(mx, 1e-4f);
  }
  __syncthreads();
  if (t == 0) {  // (the grid: as voxb_extent_kernel; rmin / rmax just written by this workgroup)
    const float vs[3] = {vs0, vs1, vs2};
    unsigned long long cells = (unsigned long long)S;
    for (int a = 0; a < 3; ++a) {
      int dmax = 1;
      for (int64_t sc = 0; sc < S; ++sc) {
        const float q = __fdiv_rn(__fsub_rn(rmax[sc * 3 + a], rmin[sc * 3 + a]), vs[a]);
        const int c = q >= 0.f && q < 2.0e9f ? (int)floorf(q) + 1 : (q >= 2.0e9f ? 0x7fffffff : 1);
        dmax = c > dmax ? c : dmax;
      }
      g->d[a] = (unsigned)dmax;
      cells = dmax >= (1 << 20) || cells > ((unsigned long long)1 << 40) ? ~0ull : cells * (unsigned long long)dmax;
    }
    const unsigned long long words = cells == ~0ull ? ~0ull : (cells + 31) / 32;
    if (words > (unsigned long long)kBitmapWords) {
      g->d[3] = 0;
      stats[5] = 2;
    } else {
      g->d[3] = (unsigned)words;
    }
  }
}

struct BitmapWs {
  float *rmin, *rmax;
  int *lo_bits, *hi_bits;  // [S, 3] ordered-integer images of the running minima / maxima, then the ticket word
  GridInfo* grid;
  uint32_t *bitmap, *word_prefix, *block_sum, *cell_of, *cnt;
  size_t total;
};
BitmapWs carve_bitmap(void* ws, int64_t M, int64_t S) {
  gpn::WsCarver w(ws, (size_t)-1);
  BitmapWs o;
  const size_t m = (size_t)(M > 0 ? M : 1);
  o.rmin = w.take<float>((size_t)(S > 0 ? S : 1) * 3);
  o.rmax = w.take<float>((size_t)(S > 0 ? S : 1) * 3);
  o.lo_bits = w.take<int>((size_t)(S > 0 ? S : 1) * 6 + 1);
  o.hi_bits = o.lo_bits + (size_t)(S > 0 ? S : 1) * 3;
  o.grid = w.take<GridInfo>(1);
  o.bitmap = w.take<uint32_t>((size_t)kBitmapWords);
  o.word_prefix = w.take<uint32_t>((size_t)kBitmapWords);
  o.block_sum = w.take<uint32_t>((size_t)(kBitmapWords / kScanBlock) + m / kScanBlock + 2);
  o.cell_of = w.take<uint32_t>(m);
  o.cnt = w.take<uint32_t>(m + 1);
  o.total = w.used;
  return o;
}
size_t bitmap_path_ws_bytes(int64_t M, int64_t S) { return carve_bitmap(nullptr, M, S).total; }

}  // namespace

extern "C" int gpn_voxelize_scenes(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C,
                                   int64_t S, const float* voxel_size_host, int n_levels, float* voxel_feats, int32_t* indices4,
                                   int32_t* pc_voxel_id, int32_t* point_order, int32_t* voxel_point_start, int64_t* stats,
                                   void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(M >= 0 && C >= 1 && S >= 1 && S < (1 << 20) && n_levels >= 0 && n_levels <= 16 && voxel_size_host && stats);
  if (!point_order || !voxel_point_start)  // (the sort-free form produces the CSR as part of its work: callers without one are rare)
    return gpn_voxelize_scenes_sorted(points, feats, seg_offsets, M, C, S, voxel_size_host, n_levels, voxel_feats, indices4,
                                      pc_voxel_id, point_order, voxel_point_start, stats, ws, ws_bytes, stream_);
  GPN_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(int64_t) * (size_t)(8 + n_levels), stream));
  if (M == 0) return GPN_OK;
  GPN_CHECK_ARG(points && feats && seg_offsets && voxel_feats && indices4 && pc_voxel_id && M < (int64_t)0x7fffffff);
  BitmapWs o = carve_bitmap(ws, M, S);
  const size_t lc_bytes = n_levels > 0 ? gpn_rulebook_level_counts_ws_bytes(M, n_levels) : 0;
  if (!ws || ws_bytes < o.total + lc_bytes) {
    gpn::set_error("gpn_voxelize_scenes: workspace too small (%zu needed, %zu given)", o.total + lc_bytes, ws_bytes);
    return GPN_ERR_WS;
  }
  void* lc_ws = static_cast<char*>(ws) + o.total;
  const int grid = (int)gpn::cdiv(M, kThreads);
  const float vs0 = voxel_size_host[0], vs1 = voxel_size_host[1], vs2 = voxel_size_host[2];
  {
    gpn::ProfScope prof(GPN_K_VOXELIZE, stream, 0.0, 4.0 * (double)M * (3 + C) + 4.0 * (double)M * (3 + C) + 4.0 * (double)M);
    // running minima start at +inf, maxima at -inf (ordered-integer images), the ticket at 0
    unsigned* ticket = reinterpret_cast<unsigned*>(o.hi_bits + (size_t)S * 3);
    hipLaunchKernelGGL(voxb_range_init_kernel, dim3((unsigned)gpn::cdiv(S * 3 + 1, (int64_t)256)), dim3(256), 0, stream, o.lo_bits, o.hi_bits,
                       ticket, S * 3);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_range_kernel, dim3((unsigned)(S * kRangeSplit)), dim3(256), 0, stream, points, seg_offsets, S, vs0, vs1, vs2,
                       o.lo_bits, o.hi_bits, ticket, o.rmin, o.rmax, o.grid, stats);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_clear_kernel, dim3(1024), dim3(256), 0, stream, o.grid, o.bitmap);
    GPN_CHECK_LAUNCH();
    GPN_CHECK_HIP(hipMemsetAsync(o.cnt, 0, sizeof(uint32_t) * (size_t)(M + 1), stream));
    hipLaunchKernelGGL(voxb_mark_kernel, dim3(grid), dim3(kThreads), 0, stream, points, seg_offsets, o.rmin, o.rmax, M, S, vs0, vs1,
                       vs2, o.grid, o.bitmap, o.cell_of, stats);
    GPN_CHECK_LAUNCH();
    // rank of every bitmap word: blocks over the bound, early exit beyond the live words (read from the device)
    const unsigned* words_dev = &o.grid->d[3];
    const int64_t wblocks = kBitmapWords / kScanBlock;
    hipLaunchKernelGGL((voxb_scan_sums_kernel<true>), dim3((unsigned)wblocks), dim3(256), 0, stream, o.bitmap, words_dev, kBitmapWords,
                       o.block_sum);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, o.block_sum, wblocks, words_dev, stats /* [0] = #voxels */);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL((voxb_scan_apply_kernel<true>), dim3((unsigned)wblocks), dim3(256), 0, stream, o.bitmap, words_dev, kBitmapWords,
                       o.block_sum, o.word_prefix, (int32_t*)nullptr);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_rank_kernel, dim3(grid), dim3(kThreads), 0, stream, o.cell_of, o.bitmap, o.word_prefix, o.grid, M, pc_voxel_id,
                       o.cnt, indices4, stats);
    GPN_CHECK_LAUNCH();
    // voxel_point_start = exclusive scan of the voxel sizes over the bound M (zeros beyond the last voxel: every entry from the
    // voxel count on holds the number of placed points, which is what the sorting form leaves in entry #voxels)
    uint32_t* csum = o.block_sum + wblocks;
    const int64_t cblocks = gpn::cdiv(M, (int64_t)kScanBlock);
    hipLaunchKernelGGL((voxb_scan_sums_kernel<false>), dim3((unsigned)cblocks), dim3(256), 0, stream, o.cnt, (const unsigned*)nullptr, M, csum);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, csum, cblocks, (const unsigned*)nullptr, (int64_t*)nullptr);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL((voxb_scan_apply_kernel<false>), dim3((unsigned)cblocks), dim3(256), 0, stream, o.cnt, (const unsigned*)nullptr, M,
                       csum, reinterpret_cast<uint32_t*>(voxel_point_start), voxel_point_start);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_place_kernel, dim3(grid), dim3(kThreads), 0, stream, pc_voxel_id, voxel_point_start, o.cnt, M, point_order);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_dropped_kernel, dim3(1), dim3(256), 0, stream, pc_voxel_id, M, stats, point_order);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(voxb_sort_kernel, dim3((unsigned)std::min<int64_t>(gpn::cdiv(M, kThreads), 2048)), dim3(kThreads), 0, stream,
                       voxel_point_start, stats, point_order);
    GPN_CHECK_LAUNCH();
    const int64_t mc = M * C;
    hipLaunchKernelGGL(vox_mean_kernel, dim3((int)gpn::cdiv(mc, kThreads)), dim3(kThreads), 0, stream, feats,
                       reinterpret_cast<const uint32_t*>(point_order), voxel_point_start, stats /* [0] = #voxels */, M, C, voxel_feats);
    GPN_CHECK_LAUNCH();
  }
  if (n_levels > 0)
    return gpn::rulebook_level_counts_dev(indices4, M, stats, S, stats + 1, n_levels, stats + 8, lc_ws, lc_bytes, stream);
  return GPN_OK;
}
