// ballquery.hip — kernel B, grid-accelerated form (include/gpn.h: gpn_ball_query_grid).
//
// Same contract as gpn_ball_query (cluster.hip), which scans every point of the query's segment: hit = same segment,
// equal label, d2 < r^2 with d2 = (dx*dx + dy*dy) + dz*dz in fp32 without fma; output = the first K hits in ascending
// point index, -1 padded.  Replaces epic_ops.ball_query.ball_query (network/grouping_utils.py:119-128), which the
// model calls twice per step with every valid point as a query (1.2e5 queries x 2e4 candidates each = the O(n^2)
// scan was 0.35-0.7 ms per call, on the critical path between two host syncs).
//
// Here the points are binned into a uniform grid of cell size 1.05 r (keys (segment, cx, cy, cz) sorted with one
// radix sort; a stable sort keeps ascending point index inside a cell).  A WAVE handles one query: 18 lanes
// binary-search the 9 key ranges that cover its 27 neighbour cells (the three cz cells of a column are contiguous in
// key order), the 64 lanes test the candidates of those ranges in parallel, hits are compacted into a per-wave LDS
// list with ballots, ranked by counting (they must come out in ascending POINT index, not cell order) and the first K
// are written.  The cell size margin makes the pruning exact: |dx| < r implies cell indices differ by at most one even
// after fp32 rounding of the cell coordinate (margin 0.05 cells vs. an error below 0.004 cells at 2^15 cells).
// Queries whose neighbourhood holds more hits than the list (dense clusters, e.g. points shifted onto their predicted
// instance centres) fall back to a wave-cooperative scan of the segment in index order, which stops after K hits - the
// regime where the plain scan is already cheap.  Grids that do not fit 15-bit cell coordinates / 16-bit segment ids
// use the fallback for every query (decided on the device: no host sync).
#include "gpn_common.h"  // (<cstring> before rocprim)

#include <rocprim/rocprim.hpp>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kCap = 256;          // hit list entries per wave
constexpr int kMaxCell = 32766;    // cell coordinates live in [1, kMaxCell]

struct GridHeader {  // device-resident
  unsigned int min_enc[3];
  unsigned int overflow;
};

__device__ __forceinline__ unsigned int enc_float(float f) {
  const unsigned int b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotonic in f
}
__device__ __forceinline__ float dec_float(unsigned int u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void bq_init_kernel(GridHeader* hdr) {
  hdr->min_enc[0] = hdr->min_enc[1] = hdr->min_enc[2] = 0xffffffffu;
  hdr->overflow = 0u;
}

__global__ __launch_bounds__(kThreads) void bq_min_kernel(const float* __restrict__ points, int64_t Np, GridHeader* hdr) {
  __shared__ unsigned int smin[3];
  if (threadIdx.x < 3) smin[threadIdx.x] = 0xffffffffu;
  __syncthreads();
  unsigned int m[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x; j < Np; j += (int64_t)gridDim.x * kThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const unsigned int e = enc_float(points[j * 3 + a]);
      m[a] = e < m[a] ? e : m[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) atomicMin(&smin[a], m[a]);
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&hdr->min_enc[threadIdx.x], smin[threadIdx.x]);
}

// cell coordinate of a position along one axis; identical code for points and queries
__device__ __forceinline__ int cell_of(float x, float origin, float inv_cell) {
  const float c = floorf(__fmul_rn(__fsub_rn(x, origin), inv_cell));
  // clamp far-away queries so that the integer arithmetic below cannot overflow
  return (int)fminf(fmaxf(c, -4.0f), 40000.0f) + 1;
}

__device__ __forceinline__ uint64_t pack_key(int seg, int cx, int cy, int cz) {
  return ((uint64_t)(unsigned)seg << 48) | ((uint64_t)(unsigned)cx << 32) | ((uint64_t)(unsigned)cy << 16) | (uint64_t)(unsigned)cz;
}

__global__ __launch_bounds__(kThreads) void bq_keys_kernel(const float* __restrict__ points,
                                                           const int32_t* __restrict__ point_labels,
                                                           const int32_t* __restrict__ batch_offsets, int64_t Np, int64_t S,
                                                           float inv_cell, GridHeader* hdr, uint64_t* __restrict__ keys,
                                                           int32_t* __restrict__ vals) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= Np) return;
  if (point_labels && point_labels[j] < 0) {  // inactive point (negative label): binned behind every searched cell range
    keys[j] = ~0ull;
    vals[j] = (int32_t)j;
    return;
  }
  // segment of point j: last s with batch_offsets[s] <= j
  int64_t lo = 0, hi = S;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)batch_offsets[mid] <= j) lo = mid; else hi = mid;
  }
  const int seg = (int)lo;
  const float ox = dec_float(hdr->min_enc[0]), oy = dec_float(hdr->min_enc[1]), oz = dec_float(hdr->min_enc[2]);
  const int cx = cell_of(points[j * 3], ox, inv_cell), cy = cell_of(points[j * 3 + 1], oy, inv_cell),
            cz = cell_of(points[j * 3 + 2], oz, inv_cell);
  if (cx < 1 || cy < 1 || cz < 1 || cx > kMaxCell || cy > kMaxCell || cz > kMaxCell || seg > 65535) hdr->overflow = 1u;
  keys[j] = pack_key(seg & 0xffff, cx & 0x7fff, cy & 0x7fff, cz & 0x7fff);
  vals[j] = (int32_t)j;
}

__global__ __launch_bounds__(kThreads) void bq_records_kernel(const float* __restrict__ points,
                                                              const int32_t* __restrict__ point_labels,
                                                              const int32_t* __restrict__ order, int64_t Np,
                                                              float4* __restrict__ rec) {
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= Np) return;
  const int32_t o = order[j];
  float4 r;
  r.x = points[(int64_t)o * 3];
  r.y = points[(int64_t)o * 3 + 1];
  r.z = points[(int64_t)o * 3 + 2];
  r.w = __int_as_float(point_labels ? point_labels[o] : 0);
  rec[j] = r;
}

__device__ __forceinline__ bool is_hit(float qx, float qy, float qz, float px, float py, float pz, float r2) {
  const float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
  const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
  return d2 < r2;
}

__global__ __launch_bounds__(kThreads) void bq_query_kernel(
    const float* __restrict__ points, const float* __restrict__ query, const int32_t* __restrict__ batch_indices,
    const int32_t* __restrict__ batch_offsets, const int32_t* __restrict__ point_labels,
    const int32_t* __restrict__ query_labels, int64_t Np, int64_t Q, float r2, float inv_cell, int K,
    const GridHeader* __restrict__ hdr, const uint64_t* __restrict__ keys, const int32_t* __restrict__ order,
    const float4* __restrict__ rec, int32_t* __restrict__ indices, int32_t* __restrict__ count) {
  __shared__ int32_t hits[kWaves][kCap];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * kWaves + wave;
  if (q >= Q) return;  // whole wave; no workgroup barrier below
  const float qx = query[q * 3], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
  const int32_t b = batch_indices[q];
  const int32_t lo = batch_offsets[b], hi = batch_offsets[b + 1];
  const bool use_labels = point_labels != nullptr && query_labels != nullptr;
  const int32_t ql = use_labels ? query_labels[q] : 0;
  if (ql < 0) {  // inactive query (negative label): no neighbours
    if (lane == 0) count[q] = 0;
    return;
  }
  int32_t* out = indices + q * K;
  const uint64_t lane_lt = (1ull << lane) - 1ull;

  bool fallback = hdr->overflow != 0u || b > 65535;
  int H = 0;
  if (!fallback) {
    const float ox = dec_float(hdr->min_enc[0]), oy = dec_float(hdr->min_enc[1]), oz = dec_float(hdr->min_enc[2]);
    const int cx = cell_of(qx, ox, inv_cell), cy = cell_of(qy, oy, inv_cell), cz = cell_of(qz, oz, inv_cell);
    // lanes 0..8: lower bound of column t's first key; lanes 9..17: upper bound of its last key
    int32_t bound = 0;
    if (lane < 18) {
      const int t = lane < 9 ? lane : lane - 9;
      const int x = cx + t / 3 - 1, y = cy + t % 3 - 1;
      int z0 = cz - 1, z1 = cz + 1;
      const bool empty = x < 1 || y < 1 || x > kMaxCell || y > kMaxCell || z1 < 1 || z0 > kMaxCell;
      z0 = z0 < 1 ? 1 : z0;
      z1 = z1 > kMaxCell ? kMaxCell : z1;
      if (!empty) {
        const bool upper = lane >= 9;
        const uint64_t key = pack_key(b, x, y, upper ? z1 : z0);
        int64_t l = 0, h = Np;  // first position whose key is >= key (lower) / > key (upper)
        while (l < h) {
          const int64_t mid = (l + h) >> 1;
          const uint64_t k = keys[mid];
          if (upper ? (k <= key) : (k < key)) l = mid + 1; else h = mid;
        }
        bound = (int32_t)l;
      }
    }
    int32_t start[9], len[9];
    int total = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      start[t] = __shfl(bound, t, 64);
      const int32_t e = __shfl(bound, t + 9, 64);
      len[t] = e > start[t] ? e - start[t] : 0;
      total += len[t];
    }
    for (int base = 0; base < total && !fallback; base += 64) {
      int c = base + lane;
      bool hit = false;
      int32_t o = -1;
      if (c < total) {
        int32_t j = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (c >= 0 && c < len[t]) j = start[t] + c;
          c -= len[t];  // c goes negative once its range has been found
        }
        const float4 r = rec[j];
        if (!use_labels || __float_as_int(r.w) == ql) {
          if (is_hit(qx, qy, qz, r.x, r.y, r.z, r2)) {
            o = order[j];
            hit = o >= lo && o < hi;  // (points past the last offset are binned with the last segment)
          }
        }
      }
      const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
      const int pos = H + __popcll(mask & lane_lt);
      if (hit && pos < kCap) hits[wave][pos] = o;
      H += __popcll(mask);
      if (H > kCap) fallback = true;  // more hits than the list holds: dense neighbourhood
    }
  }

  if (!fallback) {
    // rank by counting (hit indices are distinct): out[rank] = value for rank < K
    for (int i = lane; i < H; i += 64) {
      const int32_t v = hits[wave][i];
      int rank = 0;
      for (int m = 0; m < H; ++m) rank += hits[wave][m] < v ? 1 : 0;
      if (rank < K) out[rank] = v;
    }
    if (lane == 0) count[q] = H < K ? H : K;
    return;
  }

  // wave-cooperative scan of the segment in index order, stopping after K hits
  int cnt = 0;
  for (int32_t base = lo; base < hi && cnt < K; base += 64) {
    const int32_t j = base + lane;
    bool hit = false;
    if (j < hi && (!use_labels || point_labels[j] == ql))
      hit = is_hit(qx, qy, qz, points[(int64_t)j * 3], points[(int64_t)j * 3 + 1], points[(int64_t)j * 3 + 2], r2);
    const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
    const int pos = cnt + __popcll(mask & lane_lt);
    if (hit && pos < K) out[pos] = j;
    cnt += __popcll(mask);
  }
  if (lane == 0) count[q] = cnt < K ? cnt : K;
}

struct BqWs {
  GridHeader* hdr;
  uint64_t *keys, *keys_sorted;
  int32_t *vals, *order;
  float4* rec;
  void* prim_tmp;
  size_t prim_bytes, total;
};

BqWs carve(void* ws, size_t ws_bytes, int64_t Np) {
  gpn::WsCarver w(ws, ws_bytes);
  const size_t n = (size_t)(Np > 0 ? Np : 1);
  BqWs o;
  o.hdr = w.take<GridHeader>(1);
  o.keys = w.take<uint64_t>(n);
  o.keys_sorted = w.take<uint64_t>(n);
  o.vals = w.take<int32_t>(n);
  o.order = w.take<int32_t>(n);
  o.rec = w.take<float4>(n);
  o.prim_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, o.prim_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                  (const int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64, (hipStream_t) nullptr);
  o.prim_tmp = w.take<char>(o.prim_bytes);
  o.total = w.used;
  return o;
}

}  // namespace

extern "C" size_t gpn_ball_query_grid_ws_bytes(int64_t Np) { return carve(nullptr, 0, Np).total; }

extern "C" int gpn_ball_query_grid(const float* points, const float* query, const int32_t* batch_indices,
                                   const int32_t* batch_offsets, const int32_t* point_labels,
                                   const int32_t* query_labels, int64_t Np, int64_t Q, int64_t S, float radius, int K,
                                   int32_t* indices, int32_t* count, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bool pad = !(K & GPN_BQ_NO_PAD);  // callers that only read the first count[q] entries of a row skip the -1 fill
  K &= ~GPN_BQ_NO_PAD;
  GPN_CHECK_ARG(Q >= 0 && Np >= 0 && S >= 0 && K >= 1 && radius > 0.f);
  if (Q == 0) return GPN_OK;
  GPN_CHECK_ARG(points && query && batch_indices && batch_offsets && indices && count);
  GPN_CHECK_ARG(Np < (int64_t)0x7fffffff && Q * (int64_t)K < ((int64_t)1 << 40));
  BqWs o = carve(ws, ws_bytes, Np);
  if (!ws || ws_bytes < o.total) {
    gpn::set_error("gpn_ball_query_grid: workspace too small (%zu needed, %zu given)", o.total, ws_bytes);
    return GPN_ERR_WS;
  }
  const float r2 = radius * radius;
  const float inv_cell = 1.0f / (1.05f * radius);
  gpn::ProfScope prof(GPN_K_BALL_QUERY, stream, 0.0, 12.0 * (double)Np + 4.0 * (double)Q * K);
  if (pad) GPN_CHECK_HIP(hipMemsetAsync(indices, 0xff, sizeof(int32_t) * (size_t)Q * K, stream));
  hipLaunchKernelGGL(bq_init_kernel, dim3(1), dim3(1), 0, stream, o.hdr);
  GPN_CHECK_LAUNCH();
  if (Np > 0) {
    const int pgrid = (int)gpn::cdiv(Np, kThreads);
    hipLaunchKernelGGL(bq_min_kernel, dim3(pgrid < 256 ? pgrid : 256), dim3(kThreads), 0, stream, points, Np, o.hdr);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bq_keys_kernel, dim3(pgrid), dim3(kThreads), 0, stream, points, point_labels, batch_offsets, Np, S,
                       inv_cell, o.hdr, o.keys, o.vals);
    GPN_CHECK_LAUNCH();
    size_t tmp = o.prim_bytes;
    GPN_CHECK_HIP(rocprim::radix_sort_pairs(o.prim_tmp, tmp, o.keys, o.keys_sorted, o.vals, o.order, (size_t)Np, 0, 64,
                                            stream));
    hipLaunchKernelGGL(bq_records_kernel, dim3(pgrid), dim3(kThreads), 0, stream, points, point_labels, o.order, Np,
                       o.rec);
    GPN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(bq_query_kernel, dim3((unsigned)gpn::cdiv(Q, kWaves)), dim3(kThreads), 0, stream, points, query,
                     batch_indices, batch_offsets, point_labels, query_labels, Np, Q, r2, inv_cell, K, o.hdr,
                     o.keys_sorted, o.order, o.rec, indices, count);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
