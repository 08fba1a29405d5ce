// cluster.hip — kernels B, L, R, I, N: label-aware ball query, connected-component
// labelling, segmented reductions, instance IoU and greedy NMS.
// Replaces epic_ops.{ball_query,ccl,reduce,iou,nms} (network/grouping_utils.py:59-70,119-137,244;
// network/model.py:360-362,373-378).
#include "gpn_common.h"

namespace {

constexpr int kThreads = 256;

// ================================================================================================ B
// One thread per query; a 256-query workgroup stages candidate points through LDS in tiles of 256
// (x,y,z,label) records, so every candidate is fetched from HBM/L2 once per workgroup and then
// broadcast from LDS to the 64 lanes of each wave.  Hits are appended in ascending point index.
__global__ __launch_bounds__(kThreads) void ball_query_kernel(
    const float* __restrict__ points, const float* __restrict__ query, const int32_t* __restrict__ batch_indices,
    const int32_t* __restrict__ batch_offsets, const int32_t* __restrict__ point_labels,
    const int32_t* __restrict__ query_labels, int64_t Q, float r2, int K, int32_t* __restrict__ indices,
    int32_t* __restrict__ count) {
  __shared__ float4 tile[kThreads];
  __shared__ int32_t range_lo, range_hi;
  const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const bool active = q < Q;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  int32_t lo = 0x7fffffff, hi = 0, ql = 0;
  if (active) {
    qx = query[q * 3]; qy = query[q * 3 + 1]; qz = query[q * 3 + 2];
    const int32_t b = batch_indices[q];
    lo = batch_offsets[b];
    hi = batch_offsets[b + 1];
    if (query_labels) ql = query_labels[q];
  }
  if (threadIdx.x == 0) { range_lo = 0x7fffffff; range_hi = 0; }
  __syncthreads();
  if (active && lo < hi) { atomicMin(&range_lo, lo); atomicMax(&range_hi, hi); }
  __syncthreads();
  const int32_t blo = range_lo, bhi = range_hi;
  const bool use_labels = point_labels != nullptr && query_labels != nullptr;
  int cnt = 0;
  int32_t* out = indices + q * K;
  for (int32_t base = blo; base < bhi; base += kThreads) {
    const int32_t j = base + threadIdx.x;
    if (j < bhi) {
      float4 rec;
      rec.x = points[(int64_t)j * 3]; rec.y = points[(int64_t)j * 3 + 1]; rec.z = points[(int64_t)j * 3 + 2];
      rec.w = __int_as_float(use_labels ? point_labels[j] : 0);
      tile[threadIdx.x] = rec;
    }
    __syncthreads();
    const int32_t n_here = (bhi - base < kThreads) ? (bhi - base) : kThreads;
    if (active && cnt < K && base < hi && base + n_here > lo) {
      int32_t t0 = lo > base ? lo - base : 0;
      int32_t t1 = hi - base < n_here ? hi - base : n_here;
      for (int32_t t = t0; t < t1 && cnt < K; ++t) {
        const float4 rec = tile[t];
        if (use_labels && __float_as_int(rec.w) != ql) continue;
        const float dx = __fsub_rn(qx, rec.x), dy = __fsub_rn(qy, rec.y), dz = __fsub_rn(qz, rec.z);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (d2 < r2) out[cnt++] = base + t;
      }
    }
    const int done = (!active) || cnt >= K || (base + kThreads >= hi);
    if (__syncthreads_and(done)) break;
  }
  if (active) count[q] = cnt;
}

// ================================================================================================ L
// Lock-free union-find: the larger root is always hooked under the smaller one, so the final root of a
// component is its minimum vertex index regardless of scheduling.  One wave per vertex row.
//
// find() halves the path as it climbs (parent[x] = grandparent): x is not a root there, only roots are ever hooked, and
// the new parent is still an ancestor, so the plain store cannot undo a concurrent union.  The hook step first reduces
// the roots seen by the 64 lanes (one edge each, plus the vertex's own root) to their minimum m and then unions every
// lane's root with m: 64 CAS on 64 DIFFERENT words instead of 64 lanes fighting over the root of the one vertex they
// share - on the dense ball-query graphs of a trained network (K = 50..300 neighbours, components of thousands of
// points) that serial retry chain, over un-compressed paths, was 1.1 ms for 18k vertices / 450k edges.
//
// Loads inside find() are ordinary cached loads (workgroup-scope relaxed: no cache bypass): a stale parent is still an
// ancestor (or the vertex itself, if it was a root when cached), linking under a non-root keeps the forest acyclic
// because parent < child always holds, and every link is a device-scope CAS that fails - and returns the truth - when
// its target is no longer a root.  Device-scope loads made every find of a big component queue on the one memory
// channel that owns the component's root word.
__device__ __forceinline__ int32_t uf_load(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int32_t uf_find(int32_t* parent, int32_t x) {
  int32_t p = uf_load(parent + x);
  while (p != x) {
    const int32_t gp = uf_load(parent + p);
    if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    x = p;
    p = gp;
  }
  return x;
}
// union of two ROOT candidates (either may have been hooked meanwhile); larger under smaller.  The word a CAS targets is
// first read coherently: thousands of lanes of neighbouring vertices want the same link, one CAS makes it, and the rest
// must not queue read-modify-writes on that word just to learn that it is done.
__device__ __forceinline__ void uf_union(int32_t* parent, int32_t ra, int32_t rb) {
  while (ra != rb) {
    if (ra < rb) { const int32_t tmp = ra; ra = rb; rb = tmp; }
    int32_t old = __hip_atomic_load(parent + ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == ra) {
      old = atomicCAS(parent + ra, ra, rb);
      if (old == ra) break;
    }
    ra = uf_find(parent, old);
    rb = uf_find(parent, rb);
  }
}

// initial forest: every vertex points at its smallest neighbour if that is smaller than itself (parent < child: acyclic).
// Most of a dense component is linked here with plain stores, before any atomic is issued.
__global__ void ccl_init_kernel(const int32_t* __restrict__ begin_end, const int32_t* __restrict__ edges, int64_t Q,
                                int32_t* parent) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (v >= Q) return;
  const int32_t b = begin_end[2 * v], e = begin_end[2 * v + 1];
  int32_t m = (int32_t)v;
  for (int32_t t = b + lane; t < e; t += 64) {
    const int32_t u = edges[t];
    if (u >= 0 && u < m) m = u;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int32_t o = __shfl_xor(m, off, 64);
    m = o < m ? o : m;
  }
  if (lane == 0) parent[v] = m;
}

// the initial forest follows index gradients through space: chains of dozens of hops.  Climbing them once per VERTEX here
// (with halving, so that concurrent climbers shorten each other's way) instead of once per EDGE in the hook step is
// what makes the hook step's "same parent" test hit for almost every edge inside a tree.
__global__ void ccl_compress_kernel(int32_t* parent, int64_t Q) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const int32_t r = uf_find(parent, (int32_t)i);
  if (r != (int32_t)i) __hip_atomic_store(parent + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void ccl_hook_kernel(const int32_t* __restrict__ begin_end, const int32_t* __restrict__ edges,
                                int64_t Q, int32_t* parent) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (v >= Q) return;  // whole waves leave together
  const int32_t b = begin_end[2 * v], e = begin_end[2 * v + 1];
  for (int32_t t0 = b; t0 < e; t0 += 64) {
    const int32_t t = t0 + lane;
    int32_t u = t < e ? edges[t] : (int32_t)v;
    if (u < 0 || u >= Q) u = (int32_t)v;
    // neighbours whose parent is the vertex's parent are in its set already (the common case once paths are short):
    // they do not climb to the root at all
    const int32_t pv = uf_load(parent + v);
    const bool same = uf_load(parent + u) == pv;
    if (__builtin_amdgcn_ballot_w64(!same) == 0) continue;
    const int32_t r = uf_find(parent, same ? (int32_t)v : u);
    int32_t m = r;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const int32_t o = __shfl_xor(m, off, 64);
      m = o < m ? o : m;
    }
    // one union per DISTINCT root in the wave (the lanes of a vertex mostly see one or two neighbouring trees)
    uint64_t todo = __builtin_amdgcn_ballot_w64(r != m);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const int32_t rl = __builtin_amdgcn_readlane(r, leader);
      if (lane == leader) uf_union(parent, rl, m);
      todo &= ~__builtin_amdgcn_ballot_w64(r == rl);
    }
  }
}

__global__ void ccl_flatten_kernel(int32_t* parent, int64_t Q, int32_t* __restrict__ labels,
                                   int32_t* __restrict__ is_root) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const int32_t r = uf_find(parent, (int32_t)i);
  labels[i] = r;
  if (is_root) is_root[i] = (r == (int32_t)i) ? 1 : 0;
}

// compaction: rank[i] = number of roots with index < i  (single-workgroup blocked scan; Q is modest)
__global__ void ccl_rank_kernel(const int32_t* __restrict__ is_root, int64_t Q, int32_t* __restrict__ rank) {
  __shared__ int32_t sums[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < Q; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int32_t v = i < Q ? is_root[i] : 0;
    sums[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int32_t t = threadIdx.x >= off ? sums[threadIdx.x - off] : 0;
      __syncthreads();
      sums[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < Q) rank[i] = carry + sums[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sums[1023];
    __syncthreads();
  }
}
__global__ void ccl_relabel_kernel(int32_t* __restrict__ labels, const int32_t* __restrict__ rank, int64_t Q) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Q) labels[i] = rank[labels[i]];
}

// ================================================================================================ R
template <int MODE>
__global__ void segmented_reduce_kernel(const float* __restrict__ values, const int32_t* __restrict__ begin,
                                        const int32_t* __restrict__ end, int64_t P, int C,
                                        float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * C) return;
  const int64_t p = t / C;
  const int c = (int)(t - p * C);
  const int32_t b = begin[p], e = end[p];
  float acc = 0.f;
  for (int32_t r = b; r < e; ++r) {
    const float v = values[(int64_t)r * C + c];
    if (r == b) acc = v;
    else if (MODE == 0) acc = __fadd_rn(acc, v);
    else if (MODE == 1) acc = v < acc ? v : acc;
    else acc = v > acc ? v : acc;
  }
  out[t] = acc;
}

__global__ void segmented_maxpool_fwd_kernel(const float* __restrict__ values, const int32_t* __restrict__ begin,
                                             const int32_t* __restrict__ end, int64_t P, int C,
                                             float* __restrict__ pooled, int32_t* __restrict__ argmax) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * C) return;
  const int64_t p = t / C;
  const int c = (int)(t - p * C);
  float best = 0.f;
  int32_t bi = -1;
  for (int32_t r = begin[p]; r < end[p]; ++r) {
    const float v = values[(int64_t)r * C + c];
    if (bi < 0 || v > best) { best = v; bi = r; }
  }
  pooled[t] = best;
  argmax[t] = bi;
}

// the same per (segment, channel) result with a workgroup per segment: 256 / C row lanes walk the segment's rows, the
// lanes of a channel are merged in row-lane order (larger value wins, equal values keep the smaller row = the first
// occurrence, as the serial loop above does).  The serial form is a chain of one dependent load per member point: 73 us for
// the bench's 400 proposals.
__global__ __launch_bounds__(256) void segmented_maxpool_fwd_wg_kernel(const float* __restrict__ values,
                                                                       const int32_t* __restrict__ begin,
                                                                       const int32_t* __restrict__ end, int C,
                                                                       float* __restrict__ pooled, int32_t* __restrict__ argmax,
                                                                       int64_t P, const int64_t* __restrict__ p_dev) {
  __shared__ float sv[256];
  __shared__ int32_t si[256];
  P = gpn::live_rows(p_dev, P);  // (device-counted segments, gpn::DevRows: a workgroup walks segments with a grid stride)
  const int R = 256 / C;
  const int c = threadIdx.x % C, rl = threadIdx.x / C;
  for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
  float best = 0.f;
  int32_t bi = -1;
  if (rl < R) {
    const int32_t b = begin[p], e = end[p];
    for (int32_t r = b + rl; r < e; r += R) {
      const float v = values[(int64_t)r * C + c];
      if (bi < 0 || v > best) { best = v; bi = r; }
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  if (threadIdx.x < C) {
    for (int q = 1; q < R; ++q) {
      const float v = sv[q * C + c];
      const int32_t i = si[q * C + c];
      if (i >= 0 && (bi < 0 || v > best || (v == best && i < bi))) { best = v; bi = i; }
    }
    pooled[p * C + c] = best;
    argmax[p * C + c] = bi;
  }
  __syncthreads();  // (sv / si are rewritten by the next segment)
  }
}

__global__ void segmented_maxpool_bwd_kernel(const float* __restrict__ dpooled, const int32_t* __restrict__ argmax,
                                             int64_t P, int C, float* __restrict__ dvalues, const int64_t* __restrict__ p_dev) {
  P = gpn::live_rows(p_dev, P);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < P * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const int32_t r = argmax[t];
    if (r >= 0) dvalues[(int64_t)r * C + c] = dpooled[t];  // segments are disjoint: one writer per element
  }
}
// dvalues[0 .. *m_dev) = 0 (the rows the scatter above does not write)
__global__ void zero_rows_kernel(float* __restrict__ p, int64_t rows, int C, const int64_t* __restrict__ m_dev) {
  const int64_t total = gpn::live_rows(m_dev, rows) * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) p[t] = 0.f;
}

// ================================================================================================ I
__global__ void instance_iou_kernel(const int32_t* __restrict__ proposal_offsets,
                                    const int32_t* __restrict__ instance_labels,
                                    const int32_t* __restrict__ batch_indices,
                                    const int32_t* __restrict__ npi, int64_t P, int I, float* __restrict__ ious,
                                    const int64_t* __restrict__ p_dev) {
  extern __shared__ int32_t hist[];
  P = gpn::live_rows(p_dev, P);
  for (int64_t p = blockIdx.x; p < P; p += gridDim.x) {
  const int32_t b0 = proposal_offsets[p], b1 = proposal_offsets[p + 1];
  for (int k = threadIdx.x; k < I; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  for (int32_t m = b0 + threadIdx.x; m < b1; m += blockDim.x) {
    const int32_t l = instance_labels[m];
    if (l >= 0 && l < I) atomicAdd(&hist[l], 1);
  }
  __syncthreads();
  const int32_t b = b1 > b0 ? batch_indices[b0] : 0;
  for (int k = threadIdx.x; k < I; k += blockDim.x) {
    const int32_t n = npi[(int64_t)b * I + k];
    const int32_t uni = (b1 - b0) + n - hist[k];
    ious[p * I + k] = (n > 0 && uni > 0) ? __fdiv_rn((float)hist[k], (float)uni) : 0.f;
  }
  __syncthreads();  // (hist is rewritten by the next proposal)
  }
}

// ================================================================================================ N
// mask[a][w] bit t: proposal at sorted position w*64+t (> a) overlaps the one at position a above thr
__global__ void nms_mask_kernel(const float* __restrict__ ious, const int32_t* __restrict__ order, int64_t P,
                                int64_t W, float thr, unsigned long long* __restrict__ mask) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * W) return;
  const int64_t a = t / W, w = t - a * W;
  const int32_t i = order[a];
  unsigned long long bits = 0;
  for (int bpos = 0; bpos < 64; ++bpos) {
    const int64_t bb = w * 64 + bpos;
    if (bb < P && bb > a && ious[(int64_t)i * P + order[bb]] > thr) bits |= 1ull << bpos;
  }
  mask[t] = bits;
}

__global__ void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int32_t* __restrict__ order,
                                int64_t P, int64_t W, int32_t* __restrict__ keep, int32_t* __restrict__ num_keep) {
  extern __shared__ unsigned long long removed[];
  for (int64_t w = threadIdx.x; w < W; w += blockDim.x) removed[w] = 0;
  __syncthreads();
  int32_t n = 0;
  for (int64_t a = 0; a < P; ++a) {
    const bool dead = (removed[a >> 6] >> (a & 63)) & 1ull;
    __syncthreads();
    if (!dead) {
      if (threadIdx.x == 0) keep[n] = order[a];
      ++n;
      for (int64_t w = threadIdx.x; w < W; w += blockDim.x) removed[w] |= mask[a * W + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) num_keep[0] = n;
}

}  // namespace

// ================================================================================================
extern "C" int gpn_ball_query(const float* points, const float* query, const int32_t* batch_indices,
                              const int32_t* batch_offsets, const int32_t* point_labels,
                              const int32_t* query_labels, int64_t Np, int64_t Q, int64_t S, float radius, int K,
                              int32_t* indices, int32_t* count, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(Q >= 0 && Np >= 0 && S >= 0 && K >= 1);
  if (Q == 0) return GPN_OK;
  GPN_CHECK_ARG(points && query && batch_indices && batch_offsets && indices && count);
  GPN_CHECK_ARG(Np < (int64_t)0x7fffffff);
  const float r2 = radius * radius;
  gpn::ProfScope prof(GPN_K_BALL_QUERY, stream, 0.0, 12.0 * (double)Np + 4.0 * (double)Q * K);
  GPN_CHECK_HIP(hipMemsetAsync(indices, 0xff, sizeof(int32_t) * (size_t)Q * K, stream));
  hipLaunchKernelGGL(ball_query_kernel, dim3((int)gpn::cdiv(Q, kThreads)), dim3(kThreads), 0, stream, points,
                     query, batch_indices, batch_offsets, point_labels, query_labels, Q, r2, K, indices, count);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" size_t gpn_ccl_ws_bytes(int64_t Q) {
  gpn::WsCarver w(nullptr, 0);
  size_t q = (size_t)(Q > 0 ? Q : 1);
  w.take<int32_t>(q);
  w.take<int32_t>(q);
  w.take<int32_t>(q);
  return w.used;
}

extern "C" int gpn_ccl(const int32_t* begin_end, const int32_t* edges, int64_t Q, int64_t E, int compacted,
                       int32_t* labels, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(Q >= 0 && E >= 0);
  if (Q == 0) return GPN_OK;
  GPN_CHECK_ARG(begin_end && labels && (edges || E == 0));
  GPN_CHECK_ARG(Q < (int64_t)0x7fffffff / 64);
  gpn::WsCarver w(ws, ws_bytes);
  int32_t* parent = w.take<int32_t>((size_t)Q);
  int32_t* is_root = w.take<int32_t>((size_t)Q);
  int32_t* rank = w.take<int32_t>((size_t)Q);
  GPN_CHECK_WS(w);
  const int grid = (int)gpn::cdiv(Q, kThreads);
  gpn::ProfScope prof(GPN_K_CCL, stream, 0.0, 4.0 * ((double)E + 2.0 * (double)Q) + 4.0 * (double)Q);
  hipLaunchKernelGGL(ccl_init_kernel, dim3((int)gpn::cdiv(Q * 64, kThreads)), dim3(kThreads), 0, stream, begin_end, edges,
                     Q, parent);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(grid), dim3(kThreads), 0, stream, parent, Q);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(ccl_hook_kernel, dim3((int)gpn::cdiv(Q * 64, kThreads)), dim3(kThreads), 0, stream,
                     begin_end, edges, Q, parent);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(grid), dim3(kThreads), 0, stream, parent, Q, labels,
                     compacted ? is_root : nullptr);
  GPN_CHECK_LAUNCH();
  if (compacted) {
    hipLaunchKernelGGL(ccl_rank_kernel, dim3(1), dim3(1024), 0, stream, is_root, Q, rank);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(ccl_relabel_kernel, dim3(grid), dim3(kThreads), 0, stream, labels, rank, Q);
    GPN_CHECK_LAUNCH();
  }
  return GPN_OK;
}

extern "C" int gpn_segmented_reduce(const float* values, const int32_t* begin, const int32_t* end, int64_t P,
                                    int C, int mode, float* out, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(P >= 0 && C >= 1 && mode >= 0 && mode <= 2);
  if (P == 0) return GPN_OK;
  GPN_CHECK_ARG(values && begin && end && out);
  const dim3 grid((int)gpn::cdiv(P * C, kThreads)), block(kThreads);
  if (mode == 0) hipLaunchKernelGGL(segmented_reduce_kernel<0>, grid, block, 0, stream, values, begin, end, P, C, out);
  else if (mode == 1) hipLaunchKernelGGL(segmented_reduce_kernel<1>, grid, block, 0, stream, values, begin, end, P, C, out);
  else hipLaunchKernelGGL(segmented_reduce_kernel<2>, grid, block, 0, stream, values, begin, end, P, C, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

static int maxpool_fwd_impl(const float* values, const int32_t* begin, const int32_t* end, int64_t P, const gpn::DevRows& rows, int C,
                            float* pooled, int32_t* argmax, hipStream_t stream);
extern "C" int gpn_segmented_maxpool_fwd(const float* values, const int32_t* begin, const int32_t* end,
                                         int64_t P, int C, float* pooled, int32_t* argmax, gpn_stream_t stream_) {
  return maxpool_fwd_impl(values, begin, end, P, gpn::DevRows(), C, pooled, argmax, (hipStream_t)stream_);
}
// segment count on the device (P = the bound of begin / end / pooled / argmax); needs 256 % C == 0
extern "C" int gpn_segmented_maxpool_fwd_dev(const float* values, const int32_t* begin, const int32_t* end, int64_t P,
                                             const int64_t* p_dev, int64_t p_plan, int C, float* pooled, int32_t* argmax,
                                             gpn_stream_t stream_) {
  GPN_CHECK_ARG(p_dev != nullptr && C <= 256 && 256 % C == 0);
  return maxpool_fwd_impl(values, begin, end, P, gpn::DevRows{p_dev, p_plan}, C, pooled, argmax, (hipStream_t)stream_);
}
static int maxpool_fwd_impl(const float* values, const int32_t* begin, const int32_t* end, int64_t P, const gpn::DevRows& rows, int C,
                            float* pooled, int32_t* argmax, hipStream_t stream) {
  GPN_CHECK_ARG(P >= 0 && C >= 1);
  if (P == 0) return GPN_OK;
  GPN_CHECK_ARG(values && begin && end && pooled && argmax);
  if (C <= 256 && 256 % C == 0 && P < (int64_t)0x7fffffff)
    hipLaunchKernelGGL(segmented_maxpool_fwd_wg_kernel, dim3(gpn::dev_grid(P, gpn::plan_rows(P, rows), rows.dev != nullptr)), dim3(256), 0,
                       stream, values, begin, end, C, pooled, argmax, P, rows.dev);
  else
    hipLaunchKernelGGL(segmented_maxpool_fwd_kernel, dim3((int)gpn::cdiv(P * C, kThreads)), dim3(kThreads), 0,
                       stream, values, begin, end, P, C, pooled, argmax);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_segmented_maxpool_bwd(const float* dpooled, const int32_t* argmax, int64_t P, int C,
                                         int64_t M, float* dvalues, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(P >= 0 && C >= 1 && M >= 0);
  if (M > 0) {
    GPN_CHECK_ARG(dvalues);
    GPN_CHECK_HIP(hipMemsetAsync(dvalues, 0, sizeof(float) * (size_t)M * C, stream));
  }
  if (P == 0) return GPN_OK;
  GPN_CHECK_ARG(dpooled && argmax);
  hipLaunchKernelGGL(segmented_maxpool_bwd_kernel, dim3((int)gpn::cdiv(P * C, kThreads)), dim3(kThreads), 0,
                     stream, dpooled, argmax, P, C, dvalues, (const int64_t*)nullptr);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
// segment and row counts on the device (P, M: the bounds)
extern "C" int gpn_segmented_maxpool_bwd_dev(const float* dpooled, const int32_t* argmax, int64_t P, const int64_t* p_dev,
                                             int64_t p_plan, int C, int64_t M, const int64_t* m_dev, int64_t m_plan, float* dvalues,
                                             gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(P >= 1 && C >= 1 && M >= 1 && p_dev && m_dev && dpooled && argmax && dvalues);
  const int64_t mp = gpn::plan_rows(M, gpn::DevRows{m_dev, m_plan}), pp = gpn::plan_rows(P, gpn::DevRows{p_dev, p_plan});
  hipLaunchKernelGGL(zero_rows_kernel, dim3(gpn::dev_grid(gpn::cdiv(M * C, kThreads), gpn::cdiv(mp * C, kThreads), true)), dim3(kThreads), 0,
                     stream, dvalues, M, C, m_dev);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(segmented_maxpool_bwd_kernel, dim3(gpn::dev_grid(gpn::cdiv(P * C, kThreads), gpn::cdiv(pp * C, kThreads), true)),
                     dim3(kThreads), 0, stream, dpooled, argmax, P, C, dvalues, p_dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

static int instance_iou_impl(const int32_t* proposal_offsets, const int32_t* instance_labels, const int32_t* batch_indices,
                             const int32_t* num_points_per_instance, int64_t P, const gpn::DevRows& rows, int64_t B, int I, float* ious,
                             hipStream_t stream) {
  GPN_CHECK_ARG(P >= 0 && B >= 0 && I >= 0);
  if (P == 0 || I == 0) return GPN_OK;
  GPN_CHECK_ARG(proposal_offsets && instance_labels && batch_indices && num_points_per_instance && ious);
  GPN_CHECK_ARG(I <= 8192);
  hipLaunchKernelGGL(instance_iou_kernel, dim3(gpn::dev_grid(P, gpn::plan_rows(P, rows), rows.dev != nullptr)), dim3(128),
                     sizeof(int32_t) * I, stream, proposal_offsets, instance_labels, batch_indices, num_points_per_instance, P, I, ious,
                     rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
extern "C" int gpn_instance_iou(const int32_t* proposal_offsets, const int32_t* instance_labels,
                                const int32_t* batch_indices, const int32_t* num_points_per_instance, int64_t P,
                                int64_t B, int I, float* ious, gpn_stream_t stream_) {
  return instance_iou_impl(proposal_offsets, instance_labels, batch_indices, num_points_per_instance, P, gpn::DevRows(), B, I, ious,
                           (hipStream_t)stream_);
}
// proposal count on the device (P = the bound of proposal_offsets / ious)
extern "C" int gpn_instance_iou_dev(const int32_t* proposal_offsets, const int32_t* instance_labels, const int32_t* batch_indices,
                                    const int32_t* num_points_per_instance, int64_t P, const int64_t* p_dev, int64_t p_plan, int64_t B,
                                    int I, float* ious, gpn_stream_t stream_) {
  GPN_CHECK_ARG(p_dev != nullptr);
  return instance_iou_impl(proposal_offsets, instance_labels, batch_indices, num_points_per_instance, P, gpn::DevRows{p_dev, p_plan}, B,
                           I, ious, (hipStream_t)stream_);
}

extern "C" size_t gpn_nms_ws_bytes(int64_t P) {
  int64_t W = gpn::cdiv(P > 0 ? P : 1, 64);
  return gpn::align_up((size_t)(P > 0 ? P : 1) * W * sizeof(unsigned long long));
}

extern "C" int gpn_nms(const float* ious, const int32_t* order, int64_t P, float threshold, int32_t* keep,
                       int32_t* num_keep, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(P >= 0 && num_keep);
  if (P == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(num_keep, 0, sizeof(int32_t), stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(ious && order && keep);
  const int64_t W = gpn::cdiv(P, 64);
  GPN_CHECK_ARG(W * 8 <= 64 * 1024);
  if (!ws || ws_bytes < (size_t)P * W * sizeof(unsigned long long)) {
    gpn::set_error("gpn_nms: workspace too small");
    return GPN_ERR_WS;
  }
  unsigned long long* mask = static_cast<unsigned long long*>(ws);
  hipLaunchKernelGGL(nms_mask_kernel, dim3((int)gpn::cdiv(P * W, kThreads)), dim3(kThreads), 0, stream, ious,
                     order, P, W, threshold, mask);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), sizeof(unsigned long long) * W, stream, mask, order, P,
                     W, keep, num_keep);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
