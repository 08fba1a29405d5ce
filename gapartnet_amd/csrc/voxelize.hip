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
extern "C" size_t gpn_voxelize_scenes_ws_bytes(int64_t M, int C, int64_t S, int n_levels) {
  return gpn_voxelize_ws_bytes(M, C) + gpn::align_up((size_t)(S > 0 ? S : 1) * 6 * sizeof(float)) +
         (n_levels > 0 ? gpn_rulebook_level_counts_ws_bytes(M, n_levels) : 0);
}

extern "C" int gpn_voxelize_scenes(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C,
                                   int64_t S, const float* voxel_size_host, int n_levels, float* voxel_feats, int32_t* indices4,
                                   int32_t* pc_voxel_id, int32_t* point_order, int32_t* voxel_point_start, int64_t* stats,
                                   void* ws, size_t ws_bytes, gpn_stream_t stream_) {
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
