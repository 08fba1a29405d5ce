// postprocess.hip — the post-processing of a validation / test step's proposals in one library call (include/gpn.h section PP).
//
// Reference: GAPartNet.validation_step / test_step (network/model.py:667-692, 807-857) run, per batch,
//   filter_invalid_proposals (network/grouping_utils.py:159-218): keep proposals with score > threshold and more than
//     min_points points; every per-point / per-proposal field re-indexed (boolean-mask selections, unique_consecutive);
//   apply_nms (grouping_utils.py:221-298): dense [P, P] point-set intersections (csr @ csr.T), IoU, greedy NMS by descending score,
//     every field re-indexed again.
// Mirrored operation by operation in torch that is ~300 launches and ~25 host reads per validation step (round 4:
// profiles/r04_eval_gpu_time_by_category.txt - sorts / scans 215 launches, element-wise 182, indexing 53).  Here:
//   * the filter is a flag per proposal; the survivors are ordered by descending score with ONE stable radix sort (ties: lower
//     proposal first, as torch.sort(stable=True, descending=True) in the reference's nms wrapper);
//   * intersections are SPARSE: a point is in at most one proposal of each of the two cluster sets (model.py:256-283), so the
//     proposals that share points with proposal p are found by walking p's own points through `member_slot` (the row of a point in
//     the other set, written by the proposal stage): a wave per proposal counts them in a small LDS table - no [P, P] matrix, no
//     sort of the (point, proposal) incidence list.  IoU with the reference's fp32 arithmetic (inter / ((|a| + |b|) - inter + 1e-8));
//   * greedy NMS without the sequential walk: a proposal is kept iff none of its higher-scored neighbours (IoU > threshold) is
//     kept - decided in rounds inside one workgroup (a round decides every proposal whose higher-scored neighbours are decided;
//     the highest-scored undecided one always is), which gives exactly the sequential result;
//   * ONE compaction: ids of the kept proposals (ascending), their new CSR offsets, and the source row of every kept proposal
//     point; the caller re-indexes whatever fields it needs with index_select - no host read until it wants the two counts.
// Row counts may be device counters (gpn::DevRows convention): P / M are then bounds.
#include <atomic>

#include "gpn_common.h"  // first: pulls <cstring> ahead of the HIP/rocPRIM headers

#include <rocprim/rocprim.hpp>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxNbr = 32;   // neighbours (proposals sharing points with IoU > threshold) kept per proposal
constexpr int kTable = 64;    // LDS table slots per wave: distinct proposals that share any point with one proposal

// status of a proposal in the NMS rounds
enum : unsigned char { kOut = 0, kUndecided = 1, kKept = 2, kSuppressed = 3 };

__global__ __launch_bounds__(kThreads) void pp_flags_kernel(const float* __restrict__ score, const int64_t* __restrict__ sizes,
                                                            int64_t P, const int64_t* __restrict__ p_dev, float score_thr,
                                                            int64_t min_points, unsigned char* __restrict__ flag,
                                                            float* __restrict__ key, int32_t* __restrict__ ids) {
  const int64_t live = gpn::live_rows(p_dev, P);
  for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) {
    const bool f = p < live && score[p] > score_thr && sizes[p] > min_points;
    flag[p] = f ? 1 : 0;
    key[p] = f ? score[p] : -__builtin_huge_valf();  // (proposals that are out sort behind every survivor)
    ids[p] = (int32_t)p;
  }
}

__global__ __launch_bounds__(kThreads) void pp_rank_kernel(const int32_t* __restrict__ order, int64_t P, int32_t* __restrict__ rank) {
  for (int64_t a = (int64_t)blockIdx.x * kThreads + threadIdx.x; a < P; a += (int64_t)gridDim.x * kThreads) rank[order[a]] = (int32_t)a;
}

// a wave per surviving proposal: the proposals it shares points with, their intersection sizes, and of those the ones whose
// IoU exceeds the threshold -> nbr[p][0 .. deg[p])
__global__ __launch_bounds__(kThreads) void pp_neighbours_kernel(const unsigned char* __restrict__ flag, const int64_t* __restrict__ sizes,
                                                                 const int32_t* __restrict__ offsets, const int64_t* __restrict__ point_indices,
                                                                 const int64_t* __restrict__ proposal_indices,
                                                                 const int32_t* __restrict__ member_slot, int64_t N, int64_t P,
                                                                 const int64_t* __restrict__ p_dev, float iou_thr,
                                                                 int32_t* __restrict__ nbr, int32_t* __restrict__ deg,
                                                                 int32_t* __restrict__ overflow) {
  __shared__ int32_t t_key[kThreads / 64][kTable];
  __shared__ int32_t t_cnt[kThreads / 64][kTable];
  __shared__ int32_t t_deg[kThreads / 64];
  const int64_t live = gpn::live_rows(p_dev, P);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t waves = (int64_t)gridDim.x * (kThreads / 64);
  for (int64_t p = (int64_t)blockIdx.x * (kThreads / 64) + wave; p < live; p += waves) {
    if (!flag[p]) {
      if (lane == 0) deg[p] = 0;
      continue;  // (wave-uniform)
    }
    t_key[wave][lane] = -1;
    t_cnt[wave][lane] = 0;
    if (lane == 0) t_deg[wave] = 0;
    __builtin_amdgcn_wave_barrier();
    const int32_t r0 = offsets[p], r1 = offsets[p + 1];
    bool full = false;
    for (int32_t r = r0 + lane; r < r1; r += 64) {
      const int64_t i = point_indices[r];
      const int32_t s0 = member_slot[i], s1 = member_slot[N + i];
      const int32_t r2 = s0 == r ? s1 : s0;  // the point's row in the other cluster set
      if (r2 < 0) continue;
      const int32_t q = (int32_t)proposal_indices[r2];
      int h = q & (kTable - 1);
      int tries = 0;
      for (; tries < kTable; ++tries) {
        const int32_t old = atomicCAS(&t_key[wave][h], -1, q);
        if (old == -1 || old == q) {
          atomicAdd(&t_cnt[wave][h], 1);
          break;
        }
        h = (h + 1) & (kTable - 1);
      }
      full = full || tries == kTable;
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t q = t_key[wave][lane];
    if (q >= 0 && flag[q]) {
      // the reference's arithmetic (grouping_utils.py:286-288): union = sizes[:, None] + sizes[None, :] - inter;
      // ious = inter / (union + 1e-8), all float32
      const float inter = (float)t_cnt[wave][lane];
      const float uni = __fsub_rn(__fadd_rn((float)sizes[p], (float)sizes[q]), inter);
      const float iou = __fdiv_rn(inter, __fadd_rn(uni, 1e-8f));
      if (iou > iou_thr) {
        const int32_t pos = atomicAdd(&t_deg[wave], 1);
        if (pos < kMaxNbr) nbr[p * kMaxNbr + pos] = q;
        else full = true;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) deg[p] = t_deg[wave] < kMaxNbr ? t_deg[wave] : kMaxNbr;
    if (__builtin_amdgcn_ballot_w64(full) != 0 && lane == 0) atomicOr(overflow, 1);
    __builtin_amdgcn_wave_barrier();
  }
}

// one workgroup: the NMS rounds, then the compaction tables.  status[] (one byte per LIVE proposal) lives in LDS - 128 KiB of the
// CU's 160; the bound of 8 x 20k-point scenes is 64 000 proposals - or, when a step has more live proposals than that, in the
// workspace (`status_ws`, one byte per proposal of the bound: the workgroup's own global memory, ordered by its barriers).  Which
// of the two is decided from the device count, never from the bound: a validation step of 32 scenes has a bound of 256 001
// proposals (round 5 rejected it on the host) and a few hundred live ones.
constexpr int kNmsThreads = 1024;
constexpr int64_t kLdsProposals = 128 * 1024;

template <class Status>
__device__ __forceinline__ void nms_rounds(Status status, int& s_left, long long (*s_part)[2], const unsigned char* __restrict__ flag,
                                           const int32_t* __restrict__ rank, const int32_t* __restrict__ nbr,
                                           const int32_t* __restrict__ deg, const int64_t* __restrict__ sizes, const int64_t live,
                                           const int32_t* __restrict__ overflow, int32_t* __restrict__ kept_ids,
                                           int32_t* __restrict__ new_offsets, int64_t* __restrict__ counts) {
  const int tid = threadIdx.x;
  for (int64_t p = tid; p < live; p += kNmsThreads) status[p] = flag[p] ? kUndecided : kOut;
  __syncthreads();
  for (;;) {
    if (tid == 0) s_left = 0;
    __syncthreads();
    int left = 0;
    for (int64_t p = tid; p < live; p += kNmsThreads) {
      if (status[p] != kUndecided) continue;
      const int32_t rp = rank[p];
      bool pending = false, hit = false;
      const int d = deg[p];
      for (int e = 0; e < d; ++e) {
        const int32_t q = nbr[p * kMaxNbr + e];
        if (rank[q] < rp) {  // a neighbour visited before p in descending-score order
          const unsigned char s = status[q];
          hit = hit || s == kKept;
          pending = pending || s == kUndecided;
        }
      }
      // (a status read here may already be this round's: a decision never changes, so any interleaving ends the same)
      if (hit) status[p] = kSuppressed;
      else if (!pending) status[p] = kKept;
      else left = 1;
    }
    if (left) s_left = 1;
    __syncthreads();
    if (!s_left) break;
    __syncthreads();
  }
  // kept proposals in ascending id: thread t scans its contiguous chunk, chunk sums are scanned by thread 0 (1024 entries)
  const int64_t chunk = (live + kNmsThreads - 1) / kNmsThreads;
  const int64_t a = tid * chunk < live ? tid * chunk : live, b = a + chunk < live ? a + chunk : live;
  long long n = 0, m = 0;
  for (int64_t p = a; p < b; ++p)
    if (status[p] == kKept) ++n, m += sizes[p];
  s_part[tid][0] = n, s_part[tid][1] = m;
  __syncthreads();
  if (tid == 0) {
    long long accn = 0, accm = 0;
    for (int t = 0; t < kNmsThreads; ++t) {
      const long long tn = s_part[t][0], tm = s_part[t][1];
      s_part[t][0] = accn, s_part[t][1] = accm;
      accn += tn, accm += tm;
    }
    counts[0] = accn;                    // kept proposals
    counts[1] = accm;                    // their points
    counts[2] = overflow[0] ? 1 : 0;     // != 0: a neighbour table was too small - the results are incomplete (caller falls back)
    new_offsets[accn] = (int32_t)accm;
  }
  __syncthreads();
  n = s_part[tid][0], m = s_part[tid][1];
  for (int64_t p = a; p < b; ++p)
    if (status[p] == kKept) {
      kept_ids[n] = (int32_t)p;
      new_offsets[n] = (int32_t)m;
      ++n, m += sizes[p];
    }
}

__global__ __launch_bounds__(kNmsThreads) void pp_nms_kernel(const unsigned char* __restrict__ flag, const int32_t* __restrict__ rank,
                                                             const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg,
                                                             const int64_t* __restrict__ sizes, int64_t P,
                                                             const int64_t* __restrict__ p_dev, const int32_t* __restrict__ overflow,
                                                             unsigned char* status_ws, int64_t lds_proposals,
                                                             int32_t* __restrict__ kept_ids, int32_t* __restrict__ new_offsets,
                                                             int64_t* __restrict__ counts) {
  __shared__ unsigned char status[kLdsProposals];
  __shared__ int s_left;
  __shared__ long long s_part[kNmsThreads][2];
  const int64_t live = gpn::live_rows(p_dev, P);
  if (live <= lds_proposals)  // (uniform)
    nms_rounds<unsigned char*>(status, s_left, s_part, flag, rank, nbr, deg, sizes, live, overflow, kept_ids, new_offsets, counts);
  else  // (volatile: every access goes to memory - the waves of this workgroup exchange decisions through these bytes)
    nms_rounds<volatile unsigned char*>(status_ws, s_left, s_part, flag, rank, nbr, deg, sizes, live, overflow, kept_ids, new_offsets, counts);
}

// src_row[new_offsets[j] + t] = offsets[kept_ids[j]] + t: a wave per kept proposal
__global__ __launch_bounds__(kThreads) void pp_rows_kernel(const int32_t* __restrict__ kept_ids, const int32_t* __restrict__ new_offsets,
                                                           const int32_t* __restrict__ offsets, const int64_t* __restrict__ counts,
                                                           int64_t* __restrict__ src_row) {
  const int64_t kept = counts[0];
  const int lane = threadIdx.x & 63;
  const int64_t waves = (int64_t)gridDim.x * (kThreads / 64);
  for (int64_t j = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); j < kept; j += waves) {
    const int32_t src = offsets[kept_ids[j]], dst = new_offsets[j], n = new_offsets[j + 1] - dst;
    for (int32_t t = lane; t < n; t += 64) src_row[dst + t] = (int64_t)src + t;
  }
}

struct PpWs {
  unsigned char* flag;
  float *key, *skey;
  int32_t *ids, *order, *rank, *nbr, *deg, *overflow;
  unsigned char* status;  // the NMS kernel's status bytes when a step has more live proposals than its LDS table holds
  void* prim;
  size_t prim_bytes, total;
};

PpWs carve(void* ws, int64_t P) {
  gpn::WsCarver w(ws, (size_t)-1);
  PpWs o;
  const size_t n = (size_t)(P > 0 ? P : 1);
  o.flag = w.take<unsigned char>(n);
  o.key = w.take<float>(n), o.skey = w.take<float>(n);
  o.ids = w.take<int32_t>(n), o.order = w.take<int32_t>(n), o.rank = w.take<int32_t>(n);
  o.nbr = w.take<int32_t>(n * kMaxNbr), o.deg = w.take<int32_t>(n), o.overflow = w.take<int32_t>(1);
  o.status = w.take<unsigned char>(n);
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs_desc(nullptr, tmp, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                       n, 0u, 32u, (hipStream_t) nullptr);
  o.prim_bytes = tmp;
  o.prim = w.take<char>(tmp);
  o.total = w.used;
  return o;
}

}  // namespace

// live proposals up to which the NMS kernel keeps its status bytes in LDS (default = the table's size; tests lower it to run the
// workspace form on small inputs)
std::atomic<int64_t> g_lds_proposals{kLdsProposals};

extern "C" int64_t gpn_proposals_postprocess_lds_proposals(int64_t n) {
  if (n < 0) return g_lds_proposals.load(std::memory_order_relaxed);
  return g_lds_proposals.exchange(n > kLdsProposals ? kLdsProposals : n, std::memory_order_relaxed);
}

extern "C" size_t gpn_proposals_postprocess_ws_bytes(int64_t P) { return carve(nullptr, P).total; }

extern "C" int gpn_proposals_postprocess(const float* score_preds, const int64_t* sizes, const int32_t* proposal_offsets,
                                         const int64_t* point_indices, const int64_t* proposal_indices, const int32_t* member_slot,
                                         int64_t N, int64_t P, const int64_t* p_dev, int64_t p_plan, float score_threshold,
                                         int64_t min_points, float iou_threshold, int32_t* kept_ids, int32_t* new_offsets,
                                         int64_t* src_row, int64_t* counts, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(P >= 0 && N >= 0 && counts && new_offsets);
  if (P == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), stream));
    GPN_CHECK_HIP(hipMemsetAsync(new_offsets, 0, sizeof(int32_t), stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(score_preds && sizes && proposal_offsets && point_indices && proposal_indices && member_slot && kept_ids && src_row);
  PpWs o = carve(ws, P);
  if (!ws || ws_bytes < o.total) {
    gpn::set_error("gpn_proposals_postprocess: workspace too small (%zu needed, %zu given)", o.total, ws_bytes);
    return GPN_ERR_WS;
  }
  const gpn::DevRows rows{p_dev, p_plan};
  const int64_t Pp = gpn::plan_rows(P, rows);
  GPN_CHECK_HIP(hipMemsetAsync(o.overflow, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL(pp_flags_kernel, dim3((unsigned)std::min<int64_t>(gpn::cdiv(P, kThreads), 1024)), dim3(kThreads), 0, stream, score_preds,
                     sizes, P, p_dev, score_threshold, min_points, o.flag, o.key, o.ids);
  GPN_CHECK_LAUNCH();
  size_t tmp = o.prim_bytes;
  GPN_CHECK_HIP(rocprim::radix_sort_pairs_desc(o.prim, tmp, o.key, o.skey, o.ids, o.order, (size_t)P, 0u, 32u, stream));
  hipLaunchKernelGGL(pp_rank_kernel, dim3((unsigned)std::min<int64_t>(gpn::cdiv(P, kThreads), 1024)), dim3(kThreads), 0, stream, o.order, P,
                     o.rank);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pp_neighbours_kernel, dim3(gpn::dev_grid(gpn::cdiv(P, kThreads / 64), gpn::cdiv(Pp, kThreads / 64), p_dev != nullptr, 1, 256)),
                     dim3(kThreads), 0, stream, o.flag, sizes, proposal_offsets, point_indices, proposal_indices, member_slot, N, P, p_dev,
                     iou_threshold, o.nbr, o.deg, o.overflow);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pp_nms_kernel, dim3(1), dim3(kNmsThreads), 0, stream, o.flag, o.rank, o.nbr, o.deg, sizes, P, p_dev, o.overflow,
                     o.status, g_lds_proposals.load(std::memory_order_relaxed), kept_ids, new_offsets, counts);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pp_rows_kernel, dim3(gpn::dev_grid(gpn::cdiv(P, kThreads / 64), gpn::cdiv(Pp, kThreads / 64), true, 1, 256)), dim3(kThreads), 0,
                     stream, kept_ids, new_offsets, proposal_offsets, counts, src_row);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
