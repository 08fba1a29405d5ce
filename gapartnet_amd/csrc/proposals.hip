// proposals.hip — the proposal stage of a training / validation step as ONE library call (include/gpn.h section PR).
//
// Reference: GAPartNet.proposal_clustering_and_revoxelize (network/model.py:228-346) with cluster_proposals
// (network/grouping_utils.py:108-140) and segmented_voxelize (:47-104): boolean-mask selections, two
// ball-query + CCL + sort rounds, two unique_consecutive compactions, three segmented reductions, a dozen
// element-wise ops and a voxelisation - ~140 torch launches with five device->host reads in between (each one drains the
// GPU queue).  Here the whole stage runs on upper-bound sized buffers with every data-dependent count kept on the
// device; the caller reads the counts (valid points Q, proposal points M, proposals P, voxels V, dropped) ONCE.
//
// What makes that possible without changing any result:
//  * no compaction of the valid points: ball query and CCL run over all N points with the INVALID ones carrying label -1
//    (inactive: never a hit, never a query - ballquery.hip).  Component labels are minimum point indices, and the map
//    point index -> index among the valid points is monotone, so the clusters and their order are the reference's;
//    `sorted_indices` is reported in the valid-subset numbering the reference uses (exclusive scan of the valid flags).
//  * both cluster sets are ordered by ONE stable radix sort of 2N (label, position) pairs (set B's labels offset by N,
//    invalid points keyed behind everything) - the reference's two torch.sort calls + concatenation.
//  * runs of equal labels -> proposals, size filter, renumbering: flag + exclusive-scan passes (rocPRIM), no host read.
//  * per-proposal centre / extent / scale / jitter shift: one wave per proposal; the centre is an ordered (ascending point)
//    fp32 sum like epic_ops' segmented_reduce, every other expression follows the reference's operation order in fp32
//    without contraction, so the voxel grid is bit-identical (tests/test_golden_pipeline.py pins it to the reference).
//  * re-voxelisation reuses kernel V (gpn_voxelize_ex) on the padded point list: rows past M fall outside the last
//    segment and are dropped by the kernel's own range test.
#include "gpn_common.h"  // first: pulls <cstring> ahead of the HIP/rocPRIM headers

#include <mutex>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int kThreads = 256;

// counts[] slots (int64, device): the caller reads them once after the call
enum { kQ = 0, kM = 1, kP = 2, kV = 3, kDropped = 4, kRuns = 5, kCoarse = 6, kCounts = 8 };  // kCoarse: rows of the grid's stride-2 level

__global__ __launch_bounds__(kThreads) void prop_label_kernel(const float* __restrict__ points, int stride,
                                                              const float* __restrict__ offsets,
                                                              const int64_t* __restrict__ sem_preds,
                                                              const int32_t* __restrict__ inst, int64_t N,
                                                              int32_t* __restrict__ lab, int32_t* __restrict__ flag,
                                                              uint8_t* __restrict__ valid_mask, float* __restrict__ xyz,
                                                              float* __restrict__ xyz_shift) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  const int64_t s = sem_preds[i];
  const bool ok = s > 0 && (inst == nullptr || inst[i] >= 0);
  lab[i] = ok ? (int32_t)s : -1;
  flag[i] = ok ? 1 : 0;
  valid_mask[i] = ok ? 1 : 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = points[i * stride + a];
    xyz[i * 3 + a] = p;
    xyz_shift[i * 3 + a] = __fadd_rn(p, offsets[i * 3 + a]);  // pt_xyz + offset_preds (model.py:267)
  }
}

__global__ __launch_bounds__(kThreads) void prop_valid_kernel(const int32_t* __restrict__ flag,
                                                              const int32_t* __restrict__ rank, int64_t N,
                                                              int64_t* __restrict__ valid_indices, int64_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  if (flag[i]) valid_indices[rank[i]] = i;
  if (i == N - 1) counts[kQ] = (int64_t)rank[i] + flag[i];
}

// CSR of the (sorted) scene ids over ALL points: offsets[b] = first point of scene b, offsets[B] = N
__global__ void prop_scene_offsets_kernel(const int32_t* __restrict__ batch_indices, int64_t N, int64_t B,
                                          int32_t* __restrict__ scene_off) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  int64_t lo = 0, hi = N;  // first position whose scene id >= b
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)batch_indices[mid] < b) lo = mid + 1; else hi = mid;
  }
  scene_off[b] = (int32_t)lo;
}

__global__ __launch_bounds__(kThreads) void prop_begin_end_kernel(const int32_t* __restrict__ count, int64_t N, int K,
                                                                  int32_t* __restrict__ begin_end) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= N) return;
  begin_end[2 * i] = (int32_t)(i * K);
  begin_end[2 * i + 1] = (int32_t)(i * K) + count[i];
}

// sort keys of the concatenated label sets: [labels_A ; N + labels_B], invalid points behind everything
__global__ __launch_bounds__(kThreads) void prop_keys_kernel(const int32_t* __restrict__ lab, const int32_t* __restrict__ la,
                                                             const int32_t* __restrict__ lb, int64_t N,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (c >= 2 * N) return;
  const int64_t i = c < N ? c : c - N;
  uint32_t k = (uint32_t)(2 * N);
  if (lab[i] >= 0) k = c < N ? (uint32_t)la[i] : (uint32_t)(N + lb[i]);
  keys[c] = k;
  vals[c] = (uint32_t)c;
}

__global__ __launch_bounds__(kThreads) void prop_run_flags_kernel(const uint32_t* __restrict__ skeys, int64_t T2,
                                                                  uint32_t invalid, int32_t* __restrict__ start) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= T2) return;
  const uint32_t k = skeys[i];
  start[i] = (k != invalid && (i == 0 || skeys[i - 1] != k)) ? 1 : 0;
}

// position of every run's first element; thread T2-1 publishes the number of runs; active = 2Q elements lead the order
__global__ __launch_bounds__(kThreads) void prop_run_pos_kernel(const int32_t* __restrict__ start,
                                                                const int32_t* __restrict__ incl, int64_t T2,
                                                                int32_t* __restrict__ run_pos, int64_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= T2) return;
  if (start[i]) run_pos[incl[i] - 1] = (int32_t)i;
  if (i == T2 - 1) counts[kRuns] = incl[i];
}

__global__ __launch_bounds__(kThreads) void prop_run_keep_kernel(const int32_t* __restrict__ run_pos,
                                                                 const int64_t* __restrict__ counts, int64_t T2,
                                                                 int min_points, int32_t* __restrict__ run_size,
                                                                 int32_t* __restrict__ keep_run) {
  const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (r >= T2) return;
  const int64_t R = counts[kRuns], active = 2 * counts[kQ];
  int32_t size = 0;
  if (r < R) size = (int32_t)((r + 1 < R ? (int64_t)run_pos[r + 1] : active) - run_pos[r]);
  run_size[r] = size;
  keep_run[r] = size >= min_points ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void prop_elem_keep_kernel(const uint32_t* __restrict__ skeys,
                                                                  const int32_t* __restrict__ incl,
                                                                  const int32_t* __restrict__ keep_run, int64_t T2,
                                                                  uint32_t invalid, int32_t* __restrict__ keep_elem) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= T2) return;
  keep_elem[i] = (skeys[i] != invalid && keep_run[incl[i] - 1]) ? 1 : 0;
}

struct ScatterOut {
  int64_t *sorted_indices, *point_indices, *proposal_indices, *sizes;
  int32_t *batch_p, *sem_p, *inst_p, *proposal_offsets, *member_slot;
  float* xyz_p;
  int64_t* seg64;
};

__global__ __launch_bounds__(kThreads) void prop_scatter_kernel(
    const uint32_t* __restrict__ svals, const int32_t* __restrict__ start, const int32_t* __restrict__ incl,
    const int32_t* __restrict__ keep_elem, const int32_t* __restrict__ slot, const int32_t* __restrict__ run_size,
    const int32_t* __restrict__ new_pid, const int32_t* __restrict__ rank, const int32_t* __restrict__ lab,
    const int32_t* __restrict__ inst, const int32_t* __restrict__ batch_indices, const float* __restrict__ xyz, int64_t N,
    int64_t T2, ScatterOut o, int64_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= T2) return;
  if (i == T2 - 1) counts[kM] = (int64_t)slot[i] + keep_elem[i];
  if (!keep_elem[i]) return;
  const int32_t run = incl[i] - 1;
  const int32_t p = new_pid[run];
  const int64_t m = slot[i];
  const uint32_t c = svals[i];
  const int64_t pt = c < (uint32_t)N ? c : c - (uint32_t)N;
  o.sorted_indices[m] = rank[pt];
  o.point_indices[m] = pt;
  o.proposal_indices[m] = p;
  o.batch_p[m] = batch_indices[pt];
  o.sem_p[m] = lab[pt];
  o.inst_p[m] = inst ? inst[pt] : 0;
  o.xyz_p[m * 3] = xyz[pt * 3], o.xyz_p[m * 3 + 1] = xyz[pt * 3 + 1], o.xyz_p[m * 3 + 2] = xyz[pt * 3 + 2];
  o.member_slot[c] = (int32_t)m;
  if (start[i]) {
    o.sizes[p] = run_size[run];
    o.proposal_offsets[p] = (int32_t)m;
  }
}

// proposal count, closing offset, and the int64 segment offsets kernel V wants (segments past P are empty, at M)
__global__ __launch_bounds__(kThreads) void prop_close_kernel(const int32_t* __restrict__ keep_run,
                                                              const int32_t* __restrict__ new_pid, int64_t T2, int64_t P_ub,
                                                              int32_t* __restrict__ proposal_offsets,
                                                              int64_t* __restrict__ seg64, int64_t* __restrict__ counts) {
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (p > P_ub) return;
  const int64_t P = (int64_t)new_pid[T2 - 1] + keep_run[T2 - 1];
  const int64_t M = counts[kM];
  if (p == 0) counts[kP] = P;
  if (p >= P) proposal_offsets[p] = (int32_t)M;
  seg64[p] = p < P ? (int64_t)proposal_offsets[p] : M;
}

// One wave per proposal (grouping_utils.py:59-90).  mean = ordered fp32 sum / n; lo / hi = min / max of (xyz - mean);
// scale = min(1 / max_axis((hi - lo) / full) - 0.01, max_scale); shift = -lo*scale + clamp(full - ext - 0.001, min 0) * ra
// + clamp(full - ext + 0.001, max 0) * rb with ext = hi*scale - lo*scale.  Every step in the reference's order, fp32.
__global__ __launch_bounds__(kThreads) void prop_stats_kernel(const float* __restrict__ xyz_p,
                                                              const int32_t* __restrict__ proposal_offsets,
                                                              const int64_t* __restrict__ counts, const float* __restrict__ jitter,
                                                              float full, float inv_full, float max_scale,
                                                              float* __restrict__ mean_out,
                                                              float* __restrict__ scale_out, float* __restrict__ shift_out) {
  __shared__ float buf[kThreads / 64][64 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * (kThreads / 64) + wave;
  if (p >= counts[kP]) return;  // whole wave
  const int32_t b = proposal_offsets[p], e = proposal_offsets[p + 1];
  float* sb = buf[wave];
  // ordered sum: 64 points at a time through LDS, lanes 0..2 add their axis in ascending point order
  float acc = 0.f;
  for (int32_t base = b; base < e; base += 64) {
    const int32_t n = e - base < 64 ? e - base : 64;
    for (int t = lane; t < n * 3; t += 64) sb[t] = xyz_p[(int64_t)base * 3 + t];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane < 3)
      for (int j = 0; j < n; ++j) acc = __fadd_rn(acc, sb[j * 3 + lane]);
    __builtin_amdgcn_wave_barrier();
  }
  const float cnt = (float)(e - b);
  const float mean_l = __fdiv_rn(acc, cnt);
  const float m0 = __shfl(mean_l, 0, 64), m1 = __shfl(mean_l, 1, 64), m2 = __shfl(mean_l, 2, 64);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const float mu[3] = {m0, m1, m2};
  for (int32_t j = b + lane; j < e; j += 64) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = __fsub_rn(xyz_p[(int64_t)j * 3 + a], mu[a]);
      lo[a] = fminf(lo[a], c);
      hi[a] = fmaxf(hi[a], c);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  }
  if (lane == 0) {
    float widest = -INFINITY;
#pragma unroll
    // (hi - lo) / fullscale: torch divides a tensor by a host scalar as a * (1 / b) on the GPU (BinaryDivTrueKernel.cu),
    // which is what the reference's CUDA run and this repo's torch glue on the GPU compute
    for (int a = 0; a < 3; ++a) widest = fmaxf(widest, __fmul_rn(__fsub_rn(hi[a], lo[a]), inv_full));
    float scale = __fsub_rn(__fdiv_rn(1.0f, widest), 0.01f);
    scale = fminf(scale, max_scale);
    scale_out[p] = scale;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo_s = __fmul_rn(lo[a], scale), hi_s = __fmul_rn(hi[a], scale);
      const float ext = __fsub_rn(hi_s, lo_s);
      const float room_a = fmaxf(__fsub_rn(__fsub_rn(full, ext), 0.001f), 0.0f);
      const float room_b = fminf(__fadd_rn(__fsub_rn(full, ext), 0.001f), 0.0f);
      shift_out[p * 3 + a] = __fadd_rn(__fadd_rn(-lo_s, __fmul_rn(room_a, jitter[a])), __fmul_rn(room_b, jitter[3 + a]));
      mean_out[p * 3 + a] = mu[a];
    }
  }
}

__global__ __launch_bounds__(kThreads) void prop_scale_points_kernel(const float* __restrict__ xyz_p,
                                                                     const int64_t* __restrict__ proposal_indices,
                                                                     const int64_t* __restrict__ counts,
                                                                     const float* __restrict__ mean, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, int64_t T2,
                                                                     float* __restrict__ scaled) {
  const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (m >= T2) return;
  if (m >= counts[kM]) {  // padding rows: outside every segment anyway, coordinates outside the grid for good measure
    scaled[m * 3] = scaled[m * 3 + 1] = scaled[m * 3 + 2] = -1.0f;
    return;
  }
  const int64_t p = proposal_indices[m];
  const float s = scale[p];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    scaled[m * 3 + a] = __fadd_rn(__fmul_rn(__fsub_rn(xyz_p[m * 3 + a], mean[p * 3 + a]), s), shift[p * 3 + a]);
}

__global__ __launch_bounds__(kThreads) void prop_ranges_kernel(int64_t P_ub, float full, float* __restrict__ rmin,
                                                               float* __restrict__ rmax) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (t >= P_ub * 3) return;
  rmin[t] = 0.0f;
  rmax[t] = full;
}

// [V,4] = (proposal, x, y, z) rows; number of proposal points the voxeliser dropped (a proposal left its grid: the
// reference stops in pdb, model.py:328-330) - counted with integer atomics (exact)
__global__ __launch_bounds__(kThreads) void prop_finish_kernel(const int32_t* __restrict__ vc3, const int32_t* __restrict__ vseg,
                                                               const int32_t* __restrict__ pc_voxel_id, int64_t T2,
                                                               int32_t* __restrict__ coords4, int64_t* __restrict__ counts) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (t >= T2) return;
  if (t < counts[kV]) {
    coords4[t * 4] = vseg[t];
    coords4[t * 4 + 1] = vc3[t * 3], coords4[t * 4 + 2] = vc3[t * 3 + 1], coords4[t * 4 + 3] = vc3[t * 3 + 2];
  }
  if (t < counts[kM] && pc_voxel_id[t] < 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[kDropped]), 1ull);
}

// ---- re-voxelisation of the proposals without a sort (round 4) -----------------------------------------------------------------
// What gpn_voxelize_ex computes for the proposal points - unique (proposal, x, y, z) cells in ascending key order, every
// point's voxel, the points grouped by voxel in ascending point order (CSR), points outside their grid behind all others -
// from ~27 launches (64-bit keys, rocPRIM's merge sort of the 2 N bound, flags, scan, emit, mean) in three: the points arrive
// grouped by proposal, and a proposal's grid (fullscale + 1)^3 cells = 24 389 for the reference's 28 fits one workgroup's LDS.
//   count: per proposal a cell bitmap in LDS -> its number of voxels and of points outside the grid;
//   scan:  one workgroup: exclusive sums over the proposals -> first voxel / first position / first dropped position of each;
//   place: per proposal the bitmap again, points per cell (LDS atomics: integers), popcount prefix of the bitmap = rank of a
//          cell among the proposal's voxels (the stride-2 rulebook's trick, rulebook.hip), exclusive sum of the cell counts =
//          first position of a cell; then the points in order, 256 at a time: position = cell start + points of the cell in
//          earlier chunks + earlier points of the cell in this chunk (stable, no sort).
// Bit-equal to the sort path (tests/test_gpu_proposals.py compare both with the oracle).  GPN_PROPOSALS_REVOX=0 / a grid of
// more than kRevoxMaxCells cells: the sort path.
constexpr int kRevoxMaxCells = 32768;  // (fullscale <= 30)
constexpr int kRevoxWords = kRevoxMaxCells / 32;

struct RevoxCell {
  int cell;  // linear cell of the point inside its proposal's grid, or -1: outside (dropped)
};
__device__ __forceinline__ int revox_cell(const float* __restrict__ scaled, int64_t m, float full, int D) {
  // (vox_keys_kernel's test and arithmetic with range [0, full), voxel size 1)
  int c[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = scaled[m * 3 + a];
    ok = ok && (p >= 0.0f) && (p < full);
    const int ci = (int)floorf(__fdiv_rn(__fsub_rn(p, 0.0f), 1.0f));
    ok = ok && ci >= 0 && ci < D;
    c[a] = ci;
  }
  return ok ? (c[0] * D + c[1]) * D + c[2] : -1;
}

__global__ __launch_bounds__(kThreads) void revox_count_kernel(const float* __restrict__ scaled,
                                                               const int32_t* __restrict__ proposal_offsets,
                                                               const int64_t* __restrict__ counts, float full, int D,
                                                               int32_t* __restrict__ n_vox, int32_t* __restrict__ n_out) {
  __shared__ uint32_t bm[kRevoxWords];
  __shared__ int acc[2];
  const int words = (D * D * D + 31) >> 5;
  const int64_t P = counts[kP];
  for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
    for (int w = threadIdx.x; w < words; w += kThreads) bm[w] = 0u;
    if (threadIdx.x < 2) acc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t b = proposal_offsets[p], e = proposal_offsets[p + 1];
    int outside = 0;
    for (int64_t m = b + threadIdx.x; m < e; m += kThreads) {
      const int cell = revox_cell(scaled, m, full, D);
      if (cell >= 0) atomicOr(&bm[cell >> 5], 1u << (cell & 31));
      else ++outside;
    }
    if (outside) atomicAdd(&acc[1], outside);
    __syncthreads();
    int bits = 0;
    for (int w = threadIdx.x; w < words; w += kThreads) bits += __builtin_popcount(bm[w]);
    if (bits) atomicAdd(&acc[0], bits);
    __syncthreads();
    if (threadIdx.x == 0) n_vox[p] = acc[0], n_out[p] = acc[1];
    __syncthreads();
  }
}

// exclusive sums over the proposals (one workgroup of 1024): first voxel, first position among the kept points, first
// position among the dropped ones; totals -> counts[kV], vstart[V] (= number of kept points: where the dropped ones begin)
__global__ __launch_bounds__(1024) void revox_scan_kernel(const int32_t* __restrict__ proposal_offsets, int64_t* __restrict__ counts,
                                                          const int32_t* __restrict__ n_vox, const int32_t* __restrict__ n_out,
                                                          int32_t* __restrict__ vox_base, int32_t* __restrict__ kept_base,
                                                          int32_t* __restrict__ out_base, int32_t* __restrict__ vstart) {
  __shared__ int part[3][1024];
  __shared__ int carry[3];
  const int t = threadIdx.x;
  const int64_t P = counts[kP];
  if (t < 3) carry[t] = 0;
  __syncthreads();
  for (int64_t p0 = 0; p0 < P; p0 += 1024) {
    const int64_t p = p0 + t;
    int v[3] = {0, 0, 0};
    if (p < P) {
      const int size = proposal_offsets[p + 1] - proposal_offsets[p];
      v[0] = n_vox[p], v[2] = n_out[p], v[1] = size - v[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) part[k][t] = v[k];
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele over the chunk
      int add[3] = {0, 0, 0};
      if (t >= off) {
#pragma unroll
        for (int k = 0; k < 3; ++k) add[k] = part[k][t - off];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 3; ++k) part[k][t] += add[k];
      __syncthreads();
    }
    if (p < P) {
      vox_base[p] = carry[0] + part[0][t] - v[0];
      kept_base[p] = carry[1] + part[1][t] - v[1];
      out_base[p] = carry[2] + part[2][t] - v[2];
    }
    __syncthreads();
    if (t < 3) carry[t] += part[t][1023];
    __syncthreads();
  }
  if (t == 0) {
    counts[kV] = carry[0];
    vstart[carry[0]] = carry[1];
    kept_base[P] = carry[1];  // (total of kept points, read by the place kernel)
  }
}

// exclusive prefix of one value per thread over the workgroup's 256 threads (+ the total): wave shuffles, then the 4 wave totals
__device__ __forceinline__ int revox_thread_scan(int v, int* wave_tot /* [4] */, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) before += w < wave ? wave_tot[w] : 0;
  total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  __syncthreads();
  return before + incl - v;
}
// in-place exclusive sum of a[0 .. n) in LDS by the workgroup (a contiguous run per thread); returns the total
__device__ __forceinline__ int revox_block_scan(int* a, int n, int* wave_tot) {
  const int t = threadIdx.x;
  const int run_len = (n + kThreads - 1) / kThreads;
  const int b = min(t * run_len, n), e = min(b + run_len, n);
  int s = 0;
  for (int i = b; i < e; ++i) s += a[i];
  int total;
  int run = revox_thread_scan(s, wave_tot, total);
  for (int i = b; i < e; ++i) {
    const int c = a[i];
    a[i] = run;
    run += c;
  }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(kThreads) void revox_place_kernel(const float* __restrict__ scaled,
                                                               const int32_t* __restrict__ proposal_offsets,
                                                               const int64_t* __restrict__ counts, float full, int D, int64_t T2,
                                                               const int32_t* __restrict__ vox_base, const int32_t* __restrict__ kept_base,
                                                               const int32_t* __restrict__ out_base, int32_t* __restrict__ vc3,
                                                               int32_t* __restrict__ vseg, int32_t* __restrict__ pc_voxel_id,
                                                               int32_t* __restrict__ point_order, int32_t* __restrict__ vstart) {
  // per VOXEL of the proposal (rank of its cell among the proposal's occupied cells; what a proposal touches scales with its
  // points, not with the 24k cells of its grid): points, then first position, then cursor; [n_vox] = the dropped points
  __shared__ int cur[kRevoxMaxCells + 1];
  __shared__ uint32_t bm[kRevoxWords];
  __shared__ int wprefix[kRevoxWords];  // voxels of the proposal in earlier bitmap words
  __shared__ int wave_tot[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cells = D * D * D, words = (cells + 31) >> 5;
  const int64_t P = counts[kP], T = counts[kM];
  const int kept_total = kept_base[P];
  auto local_voxel = [&](int cell) {
    const uint32_t word = bm[cell >> 5];
    return wprefix[cell >> 5] + __builtin_popcount(word & ((1u << (cell & 31)) - 1u));
  };
  for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
    for (int w = t; w < words; w += kThreads) bm[w] = 0u;
    __syncthreads();
    const int64_t b = proposal_offsets[p], e = proposal_offsets[p + 1];
    for (int64_t m = b + t; m < e; m += kThreads) {
      const int cell = revox_cell(scaled, m, full, D);
      if (cell >= 0) atomicOr(&bm[cell >> 5], 1u << (cell & 31));
    }
    __syncthreads();
    for (int w = t; w < words; w += kThreads) wprefix[w] = __builtin_popcount(bm[w]);
    __syncthreads();
    const int n_vox = revox_block_scan(wprefix, words, wave_tot);
    for (int v = t; v <= n_vox; v += kThreads) cur[v] = 0;
    __syncthreads();
    for (int64_t m = b + t; m < e; m += kThreads) {
      const int cell = revox_cell(scaled, m, full, D);
      atomicAdd(&cur[cell >= 0 ? local_voxel(cell) : n_vox], 1);
    }
    __syncthreads();
    revox_block_scan(cur, n_vox, wave_tot);  // (cur[n_vox], the dropped points' count, is not part of the sum)
    if (t == 0) cur[n_vox] = 0;
    const int vb = vox_base[p], kb = kept_base[p], ob = kept_total + out_base[p];
    for (int w = t; w < words; w += kThreads) {  // the proposal's voxels: first position, coordinates
      uint32_t word = bm[w];
      int vid = vb + wprefix[w];
      while (word) {
        const int bit = __builtin_ctz(word);
        word &= word - 1u;
        const int c = w * 32 + bit;
        vstart[vid] = kb + cur[vid - vb];
        vc3[(int64_t)vid * 3 + 2] = c % D;
        vc3[(int64_t)vid * 3 + 1] = (c / D) % D;
        vc3[(int64_t)vid * 3 + 0] = c / (D * D);
        vseg[vid] = (int32_t)p;
        ++vid;
      }
    }
    __syncthreads();
    // the proposal's points in order, 256 at a time, the four waves one after the other: position = first position of the
    // voxel + its points so far (cursor) + earlier lanes of this wave in the same voxel (ballots over the key's bits)
    for (int64_t m0 = b; m0 < e; m0 += kThreads) {
      const int64_t m = m0 + t;
      int key = -1;  // no point
      if (m < e) {
        const int cell = revox_cell(scaled, m, full, D);
        key = cell >= 0 ? local_voxel(cell) : n_vox;
      }
      uint64_t same = __builtin_amdgcn_ballot_w64(key >= 0);
#pragma unroll
      for (int bit = 0; bit < 16; ++bit) {  // (keys < 2^15 + 1)
        const uint64_t has = __builtin_amdgcn_ballot_w64(((key >> bit) & 1) != 0);
        same &= ((key >> bit) & 1) ? has : ~has;
      }
      const int earlier = __builtin_popcountll(same & ((1ull << lane) - 1ull));
      const int group = __builtin_popcountll(same);
      for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv && key >= 0) {
          const int base = cur[key];
          const int pos = (key < n_vox ? kb : ob) + base + earlier;
          point_order[pos] = (int32_t)m;
          pc_voxel_id[m] = key < n_vox ? vb + key : -1;
          if (earlier == 0) cur[key] = base + group;  // (the wave's lanes have read `base`: one instruction stream)
        }
        __syncthreads();
      }
    }
  }
  // rows behind the last proposal (the buffers' bound): dropped, in their own order behind everything else
  for (int64_t m = T + (int64_t)blockIdx.x * kThreads + t; m < T2; m += (int64_t)gridDim.x * kThreads) {
    point_order[m] = (int32_t)m;
    pc_voxel_id[m] = -1;
  }
}

// ---- differentiable per-voxel mean of gathered point features --------------------------------------------------------
// out[v, c] = mean over the voxel's points (CSR order = ascending proposal-point index) of feats[point_indices[m], c]:
// ordered fp32 sum then one division - what kernel V computes on the gathered features (voxelize.hip vox_mean_kernel)
__global__ __launch_bounds__(kThreads) void prop_voxel_mean_kernel(const float* __restrict__ feats,
                                                                   const int64_t* __restrict__ point_indices,
                                                                   const int32_t* __restrict__ order,
                                                                   const int32_t* __restrict__ vstart, int64_t V, int C,
                                                                   float* __restrict__ out, const int64_t* __restrict__ v_dev) {
  V = gpn::live_rows(v_dev, V);  // (device-counted voxel rows, gpn::DevRows: grid-stride walk)
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < V * C; t += (int64_t)gridDim.x * kThreads) {
    const int64_t v = t / C;
    const int c = (int)(t - v * C);
    const int32_t b = vstart[v], e = vstart[v + 1];
    float acc = 0.f;
    for (int32_t j = b; j < e; ++j) acc = __fadd_rn(acc, feats[point_indices[order[j]] * C + c]);
    out[v * C + c] = __fdiv_rn(acc, (float)(e - b));
  }
}

// per-point targets of the proposal points: sem_labels / gt_npcs rows of the points the proposals are made of (the
// reference's sem_labels[rows], gt_npcs[rows] index ops, model.py:556-571) with the point count on the device
__global__ __launch_bounds__(kThreads) void prop_targets_kernel(const int64_t* __restrict__ sem_labels, const float* __restrict__ gt_npcs,
                                                                const int64_t* __restrict__ point_indices, int64_t M,
                                                                const int64_t* __restrict__ m_dev, int64_t* __restrict__ sem_out,
                                                                float* __restrict__ npcs_out) {
  M = gpn::live_rows(m_dev, M);
  for (int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x; m < M; m += (int64_t)gridDim.x * kThreads) {
    const int64_t i = point_indices[m];
    if (sem_labels) sem_out[m] = sem_labels[i];
    if (gt_npcs) {
      npcs_out[m * 3] = gt_npcs[i * 3], npcs_out[m * 3 + 1] = gt_npcs[i * 3 + 1], npcs_out[m * 3 + 2] = gt_npcs[i * 3 + 2];
    }
  }
}

// d feats[i] = sum over the (at most two) proposals point i belongs to of d out[voxel] / count[voxel]; set A's membership
// first, then set B's (fixed order, no atomics)
__global__ __launch_bounds__(kThreads) void prop_voxel_mean_bwd_kernel(const float* __restrict__ dout,
                                                                       const int32_t* __restrict__ member_slot,
                                                                       const int32_t* __restrict__ pc_voxel_id,
                                                                       const int32_t* __restrict__ vstart, int64_t N, int C,
                                                                       float* __restrict__ dfeats) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t i = t / C;
  const int c = (int)(t - i * C);
  if (i >= N) return;
  float g = 0.f;
#pragma unroll
  for (int set = 0; set < 2; ++set) {
    const int32_t m = member_slot[set * N + i];
    if (m < 0) continue;
    const int32_t v = pc_voxel_id[m];
    if (v < 0) continue;
    g = __fadd_rn(g, __fdiv_rn(dout[(int64_t)v * C + c], (float)(vstart[v + 1] - vstart[v])));
  }
  dfeats[i * C + c] = g;
}

struct PropWs {
  int32_t *lab, *flag, *rank, *scene_off, *cnt, *begin_end, *la, *lb, *start, *incl, *run_pos, *run_size, *keep_run, *new_pid,
      *keep_elem, *slot, *vc3, *vseg, *nbr, *rv_nvox, *rv_nout, *rv_vbase, *rv_kbase, *rv_obase;
  uint32_t *keys, *skeys, *vals, *svals;
  float *xyz, *xyz_shift, *mean, *scale, *shift, *scaled, *rmin, *rmax, *vf;
  int64_t* seg64;
  void *prim, *sub;
  size_t prim_bytes, sub_bytes, total;
};

size_t prim_bytes_for(int64_t T2) {
  size_t a = 0, b = 0, c = 0;
  const size_t n = (size_t)(T2 > 0 ? T2 : 1);
  (void)rocprim::radix_sort_pairs(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                  (uint32_t*)nullptr, n, 0u, 32u, (hipStream_t) nullptr);
  (void)rocprim::inclusive_scan(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, n, rocprim::plus<int32_t>(),
                                (hipStream_t) nullptr);
  (void)rocprim::exclusive_scan(nullptr, c, (const int32_t*)nullptr, (int32_t*)nullptr, 0, n, rocprim::plus<int32_t>(),
                                (hipStream_t) nullptr);
  return std::max(a, std::max(b, c));
}

PropWs carve(void* ws, size_t ws_bytes, int64_t N, int64_t B, int Kmax, int64_t P_ub) {
  gpn::WsCarver w(ws, ws_bytes);
  const size_t n = (size_t)(N > 0 ? N : 1), t2 = 2 * n, pu = (size_t)P_ub + 1;
  PropWs o;
  o.lab = w.take<int32_t>(n), o.flag = w.take<int32_t>(n), o.rank = w.take<int32_t>(n);
  o.scene_off = w.take<int32_t>((size_t)B + 2);
  o.cnt = w.take<int32_t>(n), o.begin_end = w.take<int32_t>(2 * n), o.la = w.take<int32_t>(n), o.lb = w.take<int32_t>(n);
  o.start = w.take<int32_t>(t2), o.incl = w.take<int32_t>(t2), o.run_pos = w.take<int32_t>(t2), o.run_size = w.take<int32_t>(t2);
  o.keep_run = w.take<int32_t>(t2), o.new_pid = w.take<int32_t>(t2), o.keep_elem = w.take<int32_t>(t2), o.slot = w.take<int32_t>(t2);
  o.vc3 = w.take<int32_t>(3 * t2), o.vseg = w.take<int32_t>(t2);
  o.keys = w.take<uint32_t>(t2), o.skeys = w.take<uint32_t>(t2), o.vals = w.take<uint32_t>(t2), o.svals = w.take<uint32_t>(t2);
  o.xyz = w.take<float>(3 * n), o.xyz_shift = w.take<float>(3 * n);
  o.mean = w.take<float>(3 * pu), o.scale = w.take<float>(pu), o.shift = w.take<float>(3 * pu);
  o.scaled = w.take<float>(3 * t2), o.rmin = w.take<float>(3 * pu), o.rmax = w.take<float>(3 * pu), o.vf = w.take<float>(3 * t2);
  o.seg64 = w.take<int64_t>(pu);
  o.rv_nvox = w.take<int32_t>(pu + 1), o.rv_nout = w.take<int32_t>(pu + 1), o.rv_vbase = w.take<int32_t>(pu + 1);
  o.rv_kbase = w.take<int32_t>(pu + 1), o.rv_obase = w.take<int32_t>(pu + 1);  // sort-free re-voxelisation: per-proposal counts / bases
  o.nbr = w.take<int32_t>(n * (size_t)Kmax);
  o.prim_bytes = prim_bytes_for((int64_t)t2);
  o.prim = w.take<char>(o.prim_bytes);
  o.sub_bytes = std::max(std::max(gpn_ball_query_grid_ws_bytes(N), gpn_ccl_ws_bytes(N)), gpn_voxelize_ws_bytes((int64_t)t2, 3));
  o.sub_bytes = std::max(o.sub_bytes, gpn_rulebook_level_counts_ws_bytes((int64_t)t2, 1));
  o.sub = w.take<char>(o.sub_bytes);
  o.total = w.used;
  return o;
}

// (The two cluster sets - independent chains of ~20 small launches each - run one after the other on the caller's stream: the
// second on a stream of its own, forked and joined with events, measured 7.65 / 7.73 / 7.63 / 7.64 -> 7.79 / 7.64 / 7.86 / 7.71 ms
// per step in round 4 - nothing, noisier - and was removed in round 5; profiles/r04_findings.md.)
inline int grid_of(int64_t n) { return (int)gpn::cdiv(n > 0 ? n : 1, kThreads); }

bool revox_fits(float fullscale) {
  const int64_t D = (int64_t)fullscale + 1;
  return fullscale >= 1.0f && D * D * D <= kRevoxMaxCells;
}
int revoxelize(const float* scaled, const int32_t* proposal_offsets, int64_t* counts, int64_t T2, int64_t P_ub, float fullscale,
               int32_t* vc3, int32_t* vseg, int32_t* pc_voxel_id, int32_t* point_order, int32_t* vstart, int32_t* n_vox,
               int32_t* n_out, int32_t* vox_base, int32_t* kept_base, int32_t* out_base, hipStream_t stream) {
  const int D = (int)fullscale + 1;
  const int64_t pw = P_ub > 0 ? P_ub : 1;
  hipLaunchKernelGGL(revox_count_kernel, dim3((unsigned)std::min<int64_t>(pw, 2048)), dim3(kThreads), 0, stream, scaled,
                     proposal_offsets, counts, fullscale, D, n_vox, n_out);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(revox_scan_kernel, dim3(1), dim3(1024), 0, stream, proposal_offsets, counts, n_vox, n_out, vox_base, kept_base,
                     out_base, vstart);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(revox_place_kernel, dim3((unsigned)std::min<int64_t>(std::max<int64_t>(pw, 256), 512)), dim3(kThreads), 0, stream,
                     scaled, proposal_offsets, counts, fullscale, D, T2, vox_base, kept_base, out_base, vc3, vseg, pc_voxel_id,
                     point_order, vstart);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

}  // namespace

extern "C" int64_t gpn_proposals_max_proposals(int64_t N, int min_points) {
  return 2 * N / (min_points > 0 ? min_points : 1) + 1;
}

extern "C" size_t gpn_proposals_build_ws_bytes(int64_t N, int64_t B, int K1, int K2, int min_points) {
  return carve(nullptr, 0, N, B, K1 > K2 ? K1 : K2, gpn_proposals_max_proposals(N, min_points)).total;
}

extern "C" int gpn_proposals_build(const float* points, int point_stride, const float* offset_preds, const int64_t* sem_preds,
                                   const int32_t* instance_labels, const int32_t* batch_indices, int64_t N, int64_t B,
                                   float radius, int K1, int K2, int min_points, float fullscale, float max_scale,
                                   const float* jitter, int64_t* counts, uint8_t* valid_mask, int64_t* valid_indices,
                                   int64_t* sorted_indices, int64_t* point_indices, int64_t* proposal_indices,
                                   int32_t* batch_indices_p, float* pt_xyz_p, int32_t* sem_preds_p, int32_t* instance_labels_p,
                                   int64_t* sizes, int32_t* proposal_offsets, int32_t* member_slot, int32_t* voxel_coords4,
                                   int32_t* pc_voxel_id, int32_t* point_order, int32_t* voxel_point_start, void* ws,
                                   size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(N >= 1 && B >= 1 && K1 >= 1 && K2 >= 1 && min_points >= 1 && point_stride >= 3 && radius > 0.f);
  GPN_CHECK_ARG(N < ((int64_t)1 << 30) && (int64_t)N * (K1 > K2 ? K1 : K2) < ((int64_t)1 << 31));
  GPN_CHECK_ARG(points && offset_preds && sem_preds && batch_indices && jitter && counts && valid_mask && valid_indices);
  GPN_CHECK_ARG(sorted_indices && point_indices && proposal_indices && batch_indices_p && pt_xyz_p && sem_preds_p);
  GPN_CHECK_ARG(instance_labels_p && sizes && proposal_offsets && member_slot && voxel_coords4 && pc_voxel_id);
  GPN_CHECK_ARG(point_order && voxel_point_start);
  const int64_t P_ub = gpn_proposals_max_proposals(N, min_points);
  const int64_t T2 = 2 * N;
  PropWs o = carve(ws, ws_bytes, N, B, K1 > K2 ? K1 : K2, P_ub);
  if (!ws || ws_bytes < o.total) {
    gpn::set_error("gpn_proposals_build: workspace too small (%zu needed, %zu given)", o.total, ws_bytes);
    return GPN_ERR_WS;
  }
  GPN_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * kCounts, stream));
  GPN_CHECK_HIP(hipMemsetAsync(member_slot, 0xff, sizeof(int32_t) * (size_t)T2, stream));

  // ---- valid points: labels (inactive = -1), numbering among the valid ones, contiguous coordinate arrays
  hipLaunchKernelGGL(prop_label_kernel, dim3(grid_of(N)), dim3(kThreads), 0, stream, points, point_stride, offset_preds,
                     sem_preds, instance_labels, N, o.lab, o.flag, valid_mask, o.xyz, o.xyz_shift);
  GPN_CHECK_LAUNCH();
  size_t tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::exclusive_scan(o.prim, tmp, o.flag, o.rank, 0, (size_t)N, rocprim::plus<int32_t>(), stream));
  hipLaunchKernelGGL(prop_valid_kernel, dim3(grid_of(N)), dim3(kThreads), 0, stream, o.flag, o.rank, N, valid_indices, counts);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(prop_scene_offsets_kernel, dim3((int)gpn::cdiv(B + 1, 64)), dim3(64), 0, stream, batch_indices, N, B,
                     o.scene_off);
  GPN_CHECK_LAUNCH();

  // ---- the two cluster sets: ball query (label-aware, inactive points masked) + connected components
  const float* coords[2] = {o.xyz, o.xyz_shift};
  const int ks[2] = {K1, K2};
  int32_t* labels[2] = {o.la, o.lb};
  for (int set = 0; set < 2; ++set) {
    hipStream_t st = stream;
    int32_t *nbr = o.nbr, *cnt = o.cnt, *begin_end = o.begin_end;
    void* sub = o.sub;
    const size_t sub_bytes = o.sub_bytes;
    int rc = gpn_ball_query_grid(coords[set], coords[set], batch_indices, o.scene_off, o.lab, o.lab, N, N, B, radius,
                                 ks[set] | GPN_BQ_NO_PAD, nbr, cnt, sub, sub_bytes, (gpn_stream_t)st);
    if (rc) return rc;
    hipLaunchKernelGGL(prop_begin_end_kernel, dim3(grid_of(N)), dim3(kThreads), 0, st, cnt, N, ks[set], begin_end);
    GPN_CHECK_LAUNCH();
    rc = gpn_ccl(begin_end, nbr, N, N * (int64_t)ks[set], 0, labels[set], sub, sub_bytes, (gpn_stream_t)st);
    if (rc) return rc;
  }
  // ---- one stable sort orders both sets by component (= by the component's first point), members ascending
  hipLaunchKernelGGL(prop_keys_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.lab, o.la, o.lb, N, o.keys, o.vals);
  GPN_CHECK_LAUNCH();
  unsigned bits = 1;
  while (bits < 32 && ((uint64_t)T2 >> bits) != 0) ++bits;
  tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::radix_sort_pairs(o.prim, tmp, o.keys, o.skeys, o.vals, o.svals, (size_t)T2, 0u, bits, stream));

  // ---- runs of equal keys = clusters; size filter; renumbering
  const uint32_t invalid = (uint32_t)T2;
  hipLaunchKernelGGL(prop_run_flags_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.skeys, T2, invalid, o.start);
  GPN_CHECK_LAUNCH();
  tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::inclusive_scan(o.prim, tmp, o.start, o.incl, (size_t)T2, rocprim::plus<int32_t>(), stream));
  hipLaunchKernelGGL(prop_run_pos_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.start, o.incl, T2, o.run_pos, counts);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(prop_run_keep_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.run_pos, counts, T2, min_points,
                     o.run_size, o.keep_run);
  GPN_CHECK_LAUNCH();
  tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::exclusive_scan(o.prim, tmp, o.keep_run, o.new_pid, 0, (size_t)T2, rocprim::plus<int32_t>(), stream));
  hipLaunchKernelGGL(prop_elem_keep_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.skeys, o.incl, o.keep_run, T2,
                     invalid, o.keep_elem);
  GPN_CHECK_LAUNCH();
  tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::exclusive_scan(o.prim, tmp, o.keep_elem, o.slot, 0, (size_t)T2, rocprim::plus<int32_t>(), stream));
  ScatterOut so{sorted_indices, point_indices, proposal_indices, sizes, batch_indices_p, sem_preds_p, instance_labels_p,
                proposal_offsets, member_slot, pt_xyz_p, o.seg64};
  hipLaunchKernelGGL(prop_scatter_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.svals, o.start, o.incl, o.keep_elem,
                     o.slot, o.run_size, o.new_pid, o.rank, o.lab, instance_labels, batch_indices, o.xyz, N, T2, so, counts);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(prop_close_kernel, dim3(grid_of(P_ub + 1)), dim3(kThreads), 0, stream, o.keep_run, o.new_pid, T2, P_ub,
                     proposal_offsets, o.seg64, counts);
  GPN_CHECK_LAUNCH();

  // ---- per-proposal frame, scaled coordinates, re-voxelisation into fullscale^3 grids
  hipLaunchKernelGGL(prop_stats_kernel, dim3((int)gpn::cdiv(P_ub, kThreads / 64)), dim3(kThreads), 0, stream, pt_xyz_p,
                     proposal_offsets, counts, jitter, fullscale, 1.0f / fullscale, max_scale, o.mean, o.scale, o.shift);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(prop_scale_points_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, pt_xyz_p, proposal_indices, counts,
                     o.mean, o.scale, o.shift, T2, o.scaled);
  GPN_CHECK_LAUNCH();
  if (revox_fits(fullscale)) {
    int rc = revoxelize(o.scaled, proposal_offsets, counts, T2, P_ub, fullscale, o.vc3, o.vseg, pc_voxel_id, point_order,
                        voxel_point_start, o.rv_nvox, o.rv_nout, o.rv_vbase, o.rv_kbase, o.rv_obase, stream);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(prop_ranges_kernel, dim3(grid_of(P_ub * 3)), dim3(kThreads), 0, stream, P_ub, fullscale, o.rmin, o.rmax);
    GPN_CHECK_LAUNCH();
    const float vs[3] = {1.0f, 1.0f, 1.0f};
    const int32_t dims[3] = {(int32_t)fullscale + 1, (int32_t)fullscale + 1, (int32_t)fullscale + 1};
    int rc = gpn_voxelize_ex(o.scaled, o.scaled, o.seg64, o.rmin, o.rmax, T2, 3, P_ub, vs, dims, o.vf, o.vc3, o.vseg, pc_voxel_id,
                             counts + kV, point_order, voxel_point_start, o.sub, o.sub_bytes, stream_);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(prop_finish_kernel, dim3(grid_of(T2)), dim3(kThreads), 0, stream, o.vc3, o.vseg, pc_voxel_id, T2,
                     voxel_coords4, counts);
  GPN_CHECK_LAUNCH();
  // rows of the proposal grid's coarse level (the ScoreNet / NPCS-Net U-Nets have one stride-2 level): with it in the same
  // read as the other counts, building their rulebooks needs no host read of its own
  const int32_t grid_shape[3] = {(int32_t)fullscale, (int32_t)fullscale, (int32_t)fullscale};
  return gpn_rulebook_level_counts(voxel_coords4, T2, counts + kV, P_ub > 0 ? P_ub : 1, grid_shape, 1, counts + kCoarse, o.sub, o.sub_bytes,
                                   stream_);
}

static int voxel_mean_impl(const float* feats, const int64_t* point_indices, const int32_t* point_order,
                           const int32_t* voxel_point_start, int64_t V, const gpn::DevRows& rows, int C, float* out, hipStream_t stream) {
  GPN_CHECK_ARG(V >= 0 && C >= 1);
  if (V == 0) return GPN_OK;
  GPN_CHECK_ARG(feats && point_indices && point_order && voxel_point_start && out);
  hipLaunchKernelGGL(prop_voxel_mean_kernel,
                     dim3(gpn::dev_grid(gpn::cdiv(V * C, kThreads), gpn::cdiv(gpn::plan_rows(V, rows) * C, kThreads), rows.dev != nullptr)),
                     dim3(kThreads), 0, stream, feats, point_indices, point_order, voxel_point_start, V, C, out, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_proposals_voxel_mean(const float* feats, const int64_t* point_indices, const int32_t* point_order,
                                        const int32_t* voxel_point_start, int64_t V, int C, float* out, gpn_stream_t stream_) {
  return voxel_mean_impl(feats, point_indices, point_order, voxel_point_start, V, gpn::DevRows(), C, out, (hipStream_t)stream_);
}
// voxel count on the device (V = the bound of `out`)
extern "C" int gpn_proposals_voxel_mean_dev(const float* feats, const int64_t* point_indices, const int32_t* point_order,
                                            const int32_t* voxel_point_start, int64_t V, const int64_t* v_dev, int64_t v_plan, int C,
                                            float* out, gpn_stream_t stream_) {
  GPN_CHECK_ARG(v_dev != nullptr);
  return voxel_mean_impl(feats, point_indices, point_order, voxel_point_start, V, gpn::DevRows{v_dev, v_plan}, C, out,
                         (hipStream_t)stream_);
}
// sem_labels [N] i64 and / or gt_npcs [N,3] f32 at the proposal points: sem_out [M] i64, npcs_out [M,3] f32 for the first *m_dev
// of the M rows (either source may be NULL)
extern "C" int gpn_proposals_targets_dev(const int64_t* sem_labels, const float* gt_npcs, const int64_t* point_indices, int64_t M,
                                         const int64_t* m_dev, int64_t m_plan, int64_t* sem_out, float* npcs_out, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(M >= 1 && m_dev && point_indices && (!sem_labels || sem_out) && (!gt_npcs || npcs_out));
  hipLaunchKernelGGL(prop_targets_kernel,
                     dim3(gpn::dev_grid(gpn::cdiv(M, kThreads), gpn::cdiv(gpn::plan_rows(M, gpn::DevRows{m_dev, m_plan}), kThreads), true)),
                     dim3(kThreads), 0, stream, sem_labels, gt_npcs, point_indices, M, m_dev, sem_out, npcs_out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_proposals_voxel_mean_bwd(const float* dout, const int32_t* member_slot, const int32_t* pc_voxel_id,
                                            const int32_t* voxel_point_start, int64_t N, int C, float* dfeats,
                                            gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(N >= 0 && C >= 1);
  if (N == 0) return GPN_OK;
  GPN_CHECK_ARG(dout && member_slot && pc_voxel_id && voxel_point_start && dfeats);
  hipLaunchKernelGGL(prop_voxel_mean_bwd_kernel, dim3(grid_of(N * C)), dim3(kThreads), 0, stream, dout, member_slot,
                     pc_voxel_id, voxel_point_start, N, C, dfeats);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// The re-voxelisation step of gpn_proposals_build on its own (tests; include/gpn.h): counts[1] = points, counts[2] = proposals
// (inputs, on the device), counts[3] = voxels (output).  ws: 5 x (P_ub + 2) int32.
extern "C" size_t gpn_proposals_revoxelize_ws_bytes(int64_t P_ub) {
  gpn::WsCarver w(nullptr, 0);
  for (int k = 0; k < 5; ++k) w.take<int32_t>((size_t)(P_ub > 0 ? P_ub : 1) + 2);
  return w.used;
}
extern "C" int gpn_proposals_revoxelize(const float* scaled, const int32_t* proposal_offsets, int64_t* counts, int64_t T2, int64_t P_ub,
                                        float fullscale, int32_t* voxel_coords3, int32_t* voxel_seg, int32_t* pc_voxel_id,
                                        int32_t* point_order, int32_t* voxel_point_start, void* ws, size_t ws_bytes,
                                        gpn_stream_t stream_) {
  GPN_CHECK_ARG(T2 >= 1 && P_ub >= 1 && scaled && proposal_offsets && counts);
  GPN_CHECK_ARG(voxel_coords3 && voxel_seg && pc_voxel_id && point_order && voxel_point_start);
  if (!revox_fits(fullscale)) {
    gpn::set_error("gpn_proposals_revoxelize: a grid of (%d + 1)^3 cells does not fit one workgroup's LDS (max %d cells)", (int)fullscale,
                   kRevoxMaxCells);
    return GPN_ERR_ARG;
  }
  gpn::WsCarver w(ws, ws_bytes);
  int32_t* a[5];
  for (int k = 0; k < 5; ++k) a[k] = w.take<int32_t>((size_t)P_ub + 2);
  GPN_CHECK_WS(w);
  return revoxelize(scaled, proposal_offsets, counts, T2, P_ub, fullscale, voxel_coords3, voxel_seg, pc_voxel_id, point_order,
                    voxel_point_start, a[0], a[1], a[2], a[3], a[4], (hipStream_t)stream_);
}
