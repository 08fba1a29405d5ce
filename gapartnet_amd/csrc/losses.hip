// losses.hip — kernel family P (SURVEY.md §8a "glue"): the per-point losses of the training step in one pass.
//
// Reference: network/model.py:177-226 (loss_sem_seg: focal + dice, loss_offset: L1 distance + negative cosine) over
// network/losses.py:35-64 (focal_loss, gamma = 2, rows with label == ignore_index dropped, an all-ignored batch
// gives 0) and :111-158 (dice_loss on [M, C, 1, 1] logits with the 1e-6-smoothed one-hot target, eps 1e-8).
// As PyTorch ops these are ~70 small kernels forward and ~100 backward per step, each a few microseconds on 160k x 10
// elements; here one thread owns one point (its C logits, label, predicted / true offset, instance label), the four
// sums are reduced per workgroup in double and combined in workgroup order (deterministic), and one backward kernel
// writes d logits and d offsets from the four upstream gradients.
//   focal_i = -(1 - p_t)^2 log p_t                                   mean over label != ignore
//   dice_i  = 1 - 2 sum_c p_c y_c / (sum_c (p_c + y_c) + 1e-8),  y = onehot + 1e-6      mean over all points
//   dist_i  = sum_a |o_a - g_a|,   dir_i = -(g / (|g| + 1e-8)) . (o / (|o| + 1e-8))     mean over on-part points
//             (on-part: label > 0 and instance label >= 0; an empty selection gives NaN, as torch's .mean() does)
#include "gpn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxClasses = 32;
constexpr int kMaxBlocks = 1024;
constexpr int kPartials = 9;  // focal, dice, offset distance, offset direction, kept points, on-part points; hits, part points, part hits

struct Stats {  // device-resident, written by the finalize kernel, read by backward
  double keep_count, on_count, points;
};

__device__ __forceinline__ double block_sum(double v, double* scratch /* [4] */) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// softmax of one row into p[], returns log-sum-exp
// (fixed trip counts with a c < C predicate keep p[] in registers)
__device__ __forceinline__ float row_softmax(const float* __restrict__ z, int C, float (&p)[kMaxClasses]) {
  float mx = z[0];
#pragma unroll
  for (int c = 1; c < kMaxClasses; ++c)
    if (c < C) mx = fmaxf(mx, z[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxClasses; ++c) {
    p[c] = c < C ? expf(z[c] - mx) : 0.f;
    s += p[c];
  }
  const float inv = 1.0f / s;
#pragma unroll
  for (int c = 0; c < kMaxClasses; ++c) p[c] *= inv;
  return mx + logf(s);
}

__global__ __launch_bounds__(kThreads) void point_losses_fwd_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ offsets,
    const float* __restrict__ gt_offsets, const int32_t* __restrict__ inst, int64_t M, int C, int64_t ignore_index,
    double* __restrict__ partial /* [blocks][kPartials] */, int64_t* __restrict__ preds /* optional [M] */) {
  __shared__ double scratch[4];
  double focal = 0.0, dice = 0.0, dist = 0.0, dir = 0.0, keep = 0.0, on = 0.0, hit = 0.0, part = 0.0, part_hit = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < M; i += (int64_t)gridDim.x * kThreads) {
    const int64_t lab = labels[i];
    float p[kMaxClasses];
    const float* z = logits + i * C;
    const float lse = row_softmax(z, C, p);
    const bool has_class = lab >= 0 && lab < C;
    if (preds) {  // the predicted class (torch.argmax: the first maximum) and the accuracy counts of model.py:535-541
      int best = 0;
      float bz = z[0];
#pragma unroll
      for (int c = 1; c < kMaxClasses; ++c)
        if (c < C && z[c] > bz) bz = z[c], best = c;
      preds[i] = best;
      const bool ok = (int64_t)best == lab;
      hit += ok ? 1.0 : 0.0;
      part += lab > 0 ? 1.0 : 0.0;
      part_hit += (ok && lab > 0) ? 1.0 : 0.0;
    }
    if (lab != ignore_index) {
      keep += 1.0;
      const float u = (has_class ? z[lab] : 0.f) - lse;  // log p_t
      const float q = 1.0f - expf(u);
      focal += (double)(-(q * q) * u);
    }
    float inter = 0.f, card = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) {
      if (c < C) {
        const float y = ((has_class && c == (int)lab) ? 1.0f : 0.0f) + 1e-6f;
        inter += p[c] * y;
        card += p[c] + y;
      }
    }
    dice += (double)(1.0f - 2.0f * inter / (card + 1e-8f));
    if (lab > 0 && inst[i] >= 0) {
      on += 1.0;
      const float ox = offsets[i * 3], oy = offsets[i * 3 + 1], oz = offsets[i * 3 + 2];
      const float gx = gt_offsets[i * 3], gy = gt_offsets[i * 3 + 1], gz = gt_offsets[i * 3 + 2];
      dist += (double)(fabsf(ox - gx) + fabsf(oy - gy) + fabsf(oz - gz));
      const float no = sqrtf(ox * ox + oy * oy + oz * oz) + 1e-8f, ng = sqrtf(gx * gx + gy * gy + gz * gz) + 1e-8f;
      dir += (double)(-((gx / ng) * (ox / no) + (gy / ng) * (oy / no) + (gz / ng) * (oz / no)));
    }
  }
  double vals[kPartials] = {focal, dice, dist, dir, keep, on, hit, part, part_hit};
#pragma unroll
  for (int k = 0; k < kPartials; ++k) {
    const double s = block_sum(vals[k], scratch);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kPartials + k] = s;
  }
}

__global__ void point_losses_finalize_kernel(const double* __restrict__ partial, int blocks, int64_t M,
                                             float* __restrict__ losses /* [4] */, Stats* __restrict__ stats,
                                             float* __restrict__ accu /* optional [2] */) {
  // one wave: lane l sums the partials of workgroups l, l + 64, ... in that order, then a fixed-order shuffle tree
  // (deterministic; a single thread walking all ~600 partials was a 58 us chain of dependent loads)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s[kPartials] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < blocks; b += 64)
    for (int k = 0; k < kPartials; ++k) s[k] += partial[(int64_t)b * kPartials + k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    for (int k = 0; k < kPartials; ++k) s[k] += __shfl_down(s[k], off, 64);
  if (threadIdx.x != 0) return;
  losses[0] = s[4] > 0 ? (float)(s[0] / s[4]) : 0.f;  // focal: 0 for an all-ignored batch
  losses[1] = (float)(s[1] / (double)M);
  losses[2] = (float)(s[2] / s[5]);                   // NaN when no point is on a part (as x[mask].mean())
  losses[3] = (float)(s[3] / s[5]);
  stats->keep_count = s[4];
  stats->on_count = s[5];
  stats->points = (double)M;
  if (accu) {  // (counts are exact in double; the quotients as torch forms them: both sides to fp32, one division)
    accu[0] = __fdiv_rn((float)s[6], (float)M);     // (sem_preds == sem_labels).sum().float() / M
    accu[1] = __fdiv_rn((float)s[8], (float)s[7]);  // ((sem_preds == sem_labels) & on_part).sum() / on_part.sum()
  }
}

__global__ __launch_bounds__(kThreads) void point_losses_bwd_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ offsets,
    const float* __restrict__ gt_offsets, const int32_t* __restrict__ inst, int64_t M, int C, int64_t ignore_index,
    const Stats* __restrict__ stats, const float* __restrict__ grad /* [4] */, float* __restrict__ d_logits,
    float* __restrict__ d_offsets) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= M) return;
  const float g_focal = stats->keep_count > 0 ? grad[0] / (float)stats->keep_count : 0.f;
  const float g_dice = grad[1] / (float)stats->points;
  const float inv_on = 1.0f / (float)stats->on_count;  // inf when empty: the forward value is NaN as well
  const float g_dist = grad[2] * inv_on, g_dir = grad[3] * inv_on;
  const int64_t lab = labels[i];
  float p[kMaxClasses];
  const float* z = logits + i * C;
  const float lse = row_softmax(z, C, p);
  const bool has_class = lab >= 0 && lab < C;
  // focal: f = -(1 - p_t)^2 u, u = log p_t;  df/du = -(1 - p_t)^2 + 2 (1 - p_t) p_t u;  du/dz_c = [c == t] - p_c
  float dfdu = 0.f;
  if (lab != ignore_index) {
    const float u = (has_class ? z[lab] : 0.f) - lse;
    const float pt = expf(u), q = 1.0f - pt;
    dfdu = g_focal * (-(q * q) + 2.0f * q * pt * u);
  }
  // dice = 1 - 2 I / (K + eps), K = sum_c (p_c + y_c) does not depend on z;  dI/dz_c = p_c (y_c - I)
  float inter = 0.f, card = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxClasses; ++c) {
    if (c < C) {
      const float y = ((has_class && c == (int)lab) ? 1.0f : 0.0f) + 1e-6f;
      inter += p[c] * y;
      card += p[c] + y;
    }
  }
  const float dice_scale = g_dice * (-2.0f / (card + 1e-8f));
#pragma unroll
  for (int c = 0; c < kMaxClasses; ++c) {
    if (c < C) {
      const bool is_t = has_class && c == (int)lab;
      const float y = (is_t ? 1.0f : 0.0f) + 1e-6f;
      float d = dice_scale * p[c] * (y - inter);
      d += dfdu * ((is_t ? 1.0f : 0.0f) - p[c]);
      d_logits[i * C + c] = d;
    }
  }
  float dox = 0.f, doy = 0.f, doz = 0.f;
  if (lab > 0 && inst[i] >= 0) {
    const float ox = offsets[i * 3], oy = offsets[i * 3 + 1], oz = offsets[i * 3 + 2];
    const float gx = gt_offsets[i * 3], gy = gt_offsets[i * 3 + 1], gz = gt_offsets[i * 3 + 2];
    auto sgn = [](float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); };
    dox = g_dist * sgn(ox - gx);
    doy = g_dist * sgn(oy - gy);
    doz = g_dist * sgn(oz - gz);
    // dir = -(ghat . o / (n + eps)):  d/do_b = -[ghat_b / (n + eps) - (ghat . o) o_b / (n (n + eps)^2)]   (n = |o|)
    const float n = sqrtf(ox * ox + oy * oy + oz * oz), ne = n + 1e-8f;
    const float ng = sqrtf(gx * gx + gy * gy + gz * gz) + 1e-8f;
    const float hx = gx / ng, hy = gy / ng, hz = gz / ng;
    const float dot = hx * ox + hy * oy + hz * oz;
    const float k2 = n > 0.f ? dot / (n * ne * ne) : 0.f;  // the norm's subgradient at 0 is 0 (as torch's)
    dox += g_dir * -(hx / ne - k2 * ox);
    doy += g_dir * -(hy / ne - k2 * oy);
    doz += g_dir * -(hz / ne - k2 * oz);
  }
  d_offsets[i * 3] = dox;
  d_offsets[i * 3 + 1] = doy;
  d_offsets[i * 3 + 2] = doz;
}

int fwd_blocks(int64_t M) {
  int64_t b = gpn::cdiv(M, (int64_t)kThreads * 2);
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" size_t gpn_point_losses_ws_bytes(int64_t M) {
  (void)M;
  return gpn::align_up(sizeof(Stats)) + gpn::align_up((size_t)kMaxBlocks * kPartials * sizeof(double));
}

// losses [4] f32 = (focal, dice, offset distance, offset direction); stats_out = opaque 32-byte device block that the
// backward call needs (counts of the selections)
static int point_losses_fwd_impl(const float* logits, const int64_t* labels, const float* offsets, const float* gt_offsets,
                                 const int32_t* instance_labels, int64_t M, int C, int64_t ignore_index, float* losses,
                                 int64_t* preds, float* accu, void* stats_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  GPN_CHECK_ARG(M >= 1 && C >= 1 && C <= kMaxClasses);
  GPN_CHECK_ARG(logits && labels && offsets && gt_offsets && instance_labels && losses && stats_out && ws);
  GPN_CHECK_ARG((preds == nullptr) == (accu == nullptr));
  GPN_CHECK_ARG(ws_bytes >= (size_t)kMaxBlocks * kPartials * sizeof(double));
  const int blocks = fwd_blocks(M);
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(point_losses_fwd_kernel, dim3(blocks), dim3(kThreads), 0, stream, logits, labels, offsets,
                     gt_offsets, instance_labels, M, C, ignore_index, partial, preds);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(point_losses_finalize_kernel, dim3(1), dim3(64), 0, stream, partial, blocks, M, losses,
                     static_cast<Stats*>(stats_out), accu);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_point_losses_fwd(const float* logits, const int64_t* labels, const float* offsets,
                                    const float* gt_offsets, const int32_t* instance_labels, int64_t M, int C,
                                    int64_t ignore_index, float* losses, void* stats_out, void* ws, size_t ws_bytes,
                                    gpn_stream_t stream_) {
  return point_losses_fwd_impl(logits, labels, offsets, gt_offsets, instance_labels, M, C, ignore_index, losses, nullptr, nullptr,
                               stats_out, ws, ws_bytes, (hipStream_t)stream_);
}

// the same pass also emits what the training step derives from the logits for its log and for the proposal stage: the
// predicted class of every point (torch.argmax) and the two accuracies of network/model.py:535-541 - 11 torch launches less
extern "C" int gpn_point_losses_fwd_metrics(const float* logits, const int64_t* labels, const float* offsets,
                                            const float* gt_offsets, const int32_t* instance_labels, int64_t M, int C,
                                            int64_t ignore_index, float* losses, int64_t* preds, float* accu, void* stats_out,
                                            void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  GPN_CHECK_ARG(preds && accu);
  return point_losses_fwd_impl(logits, labels, offsets, gt_offsets, instance_labels, M, C, ignore_index, losses, preds, accu,
                               stats_out, ws, ws_bytes, (hipStream_t)stream_);
}

extern "C" int gpn_point_losses_bwd(const float* logits, const int64_t* labels, const float* offsets,
                                    const float* gt_offsets, const int32_t* instance_labels, int64_t M, int C,
                                    int64_t ignore_index, const void* stats, const float* grad_losses, float* d_logits,
                                    float* d_offsets, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(M >= 1 && C >= 1 && C <= kMaxClasses);
  GPN_CHECK_ARG(logits && labels && offsets && gt_offsets && instance_labels && stats && grad_losses && d_logits &&
                d_offsets);
  hipLaunchKernelGGL(point_losses_bwd_kernel, dim3((unsigned)gpn::cdiv(M, kThreads)), dim3(kThreads), 0, stream, logits,
                     labels, offsets, gt_offsets, instance_labels, M, C, ignore_index, static_cast<const Stats*>(stats),
                     grad_losses, d_logits, d_offsets);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// ====================================================================================================================
// NPCS loss of all proposals (network/model.py:398-462 + compute_npcs_loss, network/grouping_utils.py:14-43).
// Per point: valid = predicted class == label and a non-zero NPCS target; prediction = the 3 of its 3 x (classes - 1)
// logits that belong to the predicted class; the class's symmetry type t selects a list of candidate rotations S_j and the
// loss term (group) it counts in: types 0-2 -> group 0 (2 candidates each), 3 -> group 1 (12), 4 -> group 2 (24).
// cost_j = 5 d2 if d2 <= 0.01 else sqrt(d2) - 0.05 with d2 = |pred - gt S_j - 0.5|^2; per (proposal, group): mean over the
// member points, minimum over j; per group: mean over the proposals that have members; loss = sum over groups.
// The reference (and the torch formulation in network/grouping_utils.py) evaluates this with boolean-mask selections per
// group (twelve host reads) or ~70 element-wise / segment launches forward and ~100 backward; here: one wave per
// proposal forward, one block to finish, one thread per point backward.  Reductions are fixed-order: deterministic.
namespace {

constexpr int kMaxCand = 24;

struct NpcsTypeInfo {  // per symmetry type: first matrix, number of candidates, loss group
  int first[8], count[8], group[8];
  int n_types;
};

__device__ __forceinline__ float npcs_cost(const float pred[3], const float gt[3], const float* __restrict__ S, float* r_out) {
  // target = gt (row vector) @ S; r = pred - target - 0.5
  float d2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = __fadd_rn(__fadd_rn(__fmul_rn(gt[0], S[c]), __fmul_rn(gt[1], S[3 + c])), __fmul_rn(gt[2], S[6 + c]));
    const float r = __fsub_rn(__fsub_rn(pred[c], t), 0.5f);
    if (r_out) r_out[c] = r;
    d2 = __fadd_rn(d2, __fmul_rn(r, r));
  }
  return d2;
}

__global__ __launch_bounds__(256) void npcs_loss_proposal_kernel(
    const float* __restrict__ logits, int n_cls3, const float* __restrict__ gt, const int32_t* __restrict__ sem_preds,
    const int64_t* __restrict__ sem_labels, const int32_t* __restrict__ proposal_offsets, int64_t P,
    const int64_t* __restrict__ sym_of_class, const float* __restrict__ mats, NpcsTypeInfo info,
    float* __restrict__ best /* [P,3] */, int32_t* __restrict__ arg /* [P,3] */, int32_t* __restrict__ cnt /* [P,3] */,
    const int64_t* __restrict__ p_dev) {
  const int lane = threadIdx.x & 63;
  P = gpn::live_rows(p_dev, P);  // (device-counted proposals, gpn::DevRows: a wave walks proposals with a grid stride)
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (int64_t)gridDim.x * 4) {
  const int32_t b = proposal_offsets[p], e = proposal_offsets[p + 1];
  int members[3] = {0, 0, 0};
  for (int32_t m = b + lane; m < e; m += 64) {
    const int32_t cls = sem_preds[m];
    const bool valid = (int64_t)cls == sem_labels[m] && (gt[m * 3] != 0.f || gt[m * 3 + 1] != 0.f || gt[m * 3 + 2] != 0.f);
    if (valid) {
      const int g = info.group[(int)sym_of_class[cls]];
      members[0] += g == 0, members[1] += g == 1, members[2] += g == 2;
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) members[g] += __shfl_xor(members[g], off, 64);
  for (int g = 0; g < 3; ++g) {
    float bestv = 0.f;
    int bestj = 0;
    if (members[g] > 0) {  // wave-uniform
      float acc[kMaxCand];
#pragma unroll
      for (int j = 0; j < kMaxCand; ++j) acc[j] = 0.f;
      int nc = 0;
      for (int32_t m = b + lane; m < e; m += 64) {
        const int32_t cls = sem_preds[m];
        const float gv[3] = {gt[m * 3], gt[m * 3 + 1], gt[m * 3 + 2]};
        const bool valid = (int64_t)cls == sem_labels[m] && (gv[0] != 0.f || gv[1] != 0.f || gv[2] != 0.f);
        if (!valid) continue;
        const int t = (int)sym_of_class[cls];
        if (info.group[t] != g) continue;
        const float* row = logits + (int64_t)m * n_cls3 + 3 * (cls - 1);
        const float pv[3] = {row[0], row[1], row[2]};
        const int first = info.first[t];
        nc = info.count[t];
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
          if (j < nc) {
            const float d2 = npcs_cost(pv, gv, mats + (first + j) * 9, nullptr);
            acc[j] = __fadd_rn(acc[j], d2 <= 0.01f ? __fmul_rn(5.f, d2) : __fsub_rn(sqrtf(d2), 0.05f));
          }
        }
      }
      // candidates per group are the same for every type of the group (2 / 12 / 24): lanes without members report 0
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) nc = max(nc, __shfl_xor(nc, off, 64));
      bestv = INFINITY;
#pragma unroll
      for (int j = 0; j < kMaxCand; ++j) {
        float s = acc[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s = __fadd_rn(s, __shfl_xor(s, off, 64));
        const float mean = __fdiv_rn(s, (float)members[g]);
        if (j < nc && mean < bestv) { bestv = mean; bestj = j; }  // first minimum, like torch.min
      }
    }
    if (lane == 0) {
      best[p * 3 + g] = bestv;
      arg[p * 3 + g] = bestj;
      cnt[p * 3 + g] = members[g];
    }
  }
  }
}

// loss = sum_g (sum over proposals with members of best) / (their number); stats = {n_has[3]} for the backward
__global__ __launch_bounds__(1024) void npcs_loss_finish_kernel(const float* __restrict__ best, const int32_t* __restrict__ cnt,
                                                                int64_t P, float* __restrict__ loss, float* __restrict__ n_has,
                                                                const int64_t* __restrict__ p_dev) {
  P = gpn::live_rows(p_dev, P);  // (no proposal: every group is empty, the loss is 0)
  __shared__ double ssum[3][1024];
  __shared__ int scnt[3][1024];
  double s[3] = {0.0, 0.0, 0.0};
  int c[3] = {0, 0, 0};
  const int64_t per = (P + 1023) / 1024;  // contiguous chunk per thread: fixed order
  for (int64_t p = threadIdx.x * per; p < (threadIdx.x + 1) * per && p < P; ++p)
#pragma unroll
    for (int g = 0; g < 3; ++g)
      if (cnt[p * 3 + g] > 0) { s[g] += (double)best[p * 3 + g]; c[g] += 1; }
#pragma unroll
  for (int g = 0; g < 3; ++g) ssum[g][threadIdx.x] = s[g], scnt[g][threadIdx.x] = c[g];
  __syncthreads();
  for (int stride = 512; stride >= 1; stride >>= 1) {
    if ((int)threadIdx.x < stride)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        ssum[g][threadIdx.x] += ssum[g][threadIdx.x + stride];
        scnt[g][threadIdx.x] += scnt[g][threadIdx.x + stride];
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float total = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      n_has[g] = (float)scnt[g][0];
      if (scnt[g][0] > 0) total += (float)(ssum[g][0] / (double)scnt[g][0]);
    }
    loss[0] = total;
  }
}

__global__ __launch_bounds__(256) void npcs_loss_bwd_kernel(
    const float* __restrict__ logits, int n_cls3, const float* __restrict__ gt, const int32_t* __restrict__ sem_preds,
    const int64_t* __restrict__ sem_labels, const int64_t* __restrict__ proposal_indices, int64_t M,
    const int64_t* __restrict__ sym_of_class, const float* __restrict__ mats, NpcsTypeInfo info,
    const int32_t* __restrict__ arg, const int32_t* __restrict__ cnt, const float* __restrict__ n_has,
    const float* __restrict__ grad_loss, float* __restrict__ d_logits, const int64_t* __restrict__ m_dev) {
  M = gpn::live_rows(m_dev, M);
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float* drow = d_logits + m * n_cls3;
    for (int c = 0; c < n_cls3; ++c) drow[c] = 0.f;
    const int32_t cls = sem_preds[m];
    const float gv[3] = {gt[m * 3], gt[m * 3 + 1], gt[m * 3 + 2]};
    const bool valid = (int64_t)cls == sem_labels[m] && (gv[0] != 0.f || gv[1] != 0.f || gv[2] != 0.f);
    if (!valid) continue;
    const int t = (int)sym_of_class[cls];
    const int g = info.group[t];
    const int64_t p = proposal_indices[m];
    const float* row = logits + m * n_cls3 + 3 * (cls - 1);
    const float pv[3] = {row[0], row[1], row[2]};
    float r[3];
    const float d2 = npcs_cost(pv, gv, mats + (info.first[t] + arg[p * 3 + g]) * 9, r);
    const float coef = grad_loss[0] / (n_has[g] * (float)cnt[p * 3 + g]);
    const float k = d2 <= 0.01f ? 10.f : 1.f / sqrtf(d2);
#pragma unroll
    for (int c = 0; c < 3; ++c) drow[3 * (cls - 1) + c] = coef * k * r[c];
  }
}

}  // namespace

// sym_of_class [n_classes] i64 (symmetry type of every class, gapartnet.yaml symmetry_indices); mats [n_mats,3,3] f32 =
// the candidate rotations of all types back to back; type_first / type_count / type_group [n_types] (host).
// scratch [P * 9 + 4] floats (best [P,3] | arg [P,3] i32 | cnt [P,3] i32 | n_has [3]) is kept by the caller for the backward.
static int npcs_loss_fwd_impl(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                              const int64_t* sem_labels, const int32_t* proposal_offsets, int64_t P, const gpn::DevRows& rows,
                              const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                              const int32_t* type_group, int n_types, float* loss, void* scratch, hipStream_t stream);
extern "C" int gpn_npcs_loss_fwd(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                                 const int64_t* sem_labels, const int32_t* proposal_offsets, int64_t P,
                                 const int64_t* sym_of_class, const float* mats, const int32_t* type_first,
                                 const int32_t* type_count, const int32_t* type_group, int n_types, float* loss, void* scratch,
                                 gpn_stream_t stream_) {
  return npcs_loss_fwd_impl(logits, n_cls3, gt_npcs, sem_preds, sem_labels, proposal_offsets, P, gpn::DevRows(), sym_of_class, mats,
                            type_first, type_count, type_group, n_types, loss, scratch, (hipStream_t)stream_);
}
// proposal count on the device (P = the bound of proposal_offsets and of the scratch layout)
extern "C" int gpn_npcs_loss_fwd_dev(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                                     const int64_t* sem_labels, const int32_t* proposal_offsets, int64_t P, const int64_t* p_dev,
                                     int64_t p_plan, const int64_t* sym_of_class, const float* mats, const int32_t* type_first,
                                     const int32_t* type_count, const int32_t* type_group, int n_types, float* loss, void* scratch,
                                     gpn_stream_t stream_) {
  GPN_CHECK_ARG(p_dev != nullptr);
  return npcs_loss_fwd_impl(logits, n_cls3, gt_npcs, sem_preds, sem_labels, proposal_offsets, P, gpn::DevRows{p_dev, p_plan},
                            sym_of_class, mats, type_first, type_count, type_group, n_types, loss, scratch, (hipStream_t)stream_);
}
static int npcs_loss_fwd_impl(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                              const int64_t* sem_labels, const int32_t* proposal_offsets, int64_t P, const gpn::DevRows& rows,
                              const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                              const int32_t* type_group, int n_types, float* loss, void* scratch, hipStream_t stream) {
  GPN_CHECK_ARG(P >= 1 && n_cls3 >= 3 && n_types >= 1 && n_types <= 8);
  GPN_CHECK_ARG(logits && gt_npcs && sem_preds && sem_labels && proposal_offsets && sym_of_class && mats && loss && scratch);
  NpcsTypeInfo info;
  info.n_types = n_types;
  for (int t = 0; t < n_types; ++t) {
    GPN_CHECK_ARG(type_count[t] >= 1 && type_count[t] <= kMaxCand && type_group[t] >= 0 && type_group[t] < 3);
    info.first[t] = type_first[t], info.count[t] = type_count[t], info.group[t] = type_group[t];
  }
  float* best = static_cast<float*>(scratch);
  int32_t* arg = reinterpret_cast<int32_t*>(best + P * 3);
  int32_t* cnt = arg + P * 3;
  float* n_has = reinterpret_cast<float*>(cnt + P * 3);
  hipLaunchKernelGGL(npcs_loss_proposal_kernel, dim3(gpn::dev_grid(gpn::cdiv(P, 4), gpn::cdiv(gpn::plan_rows(P, rows), 4), rows.dev != nullptr)),
                     dim3(256), 0, stream, logits, n_cls3, gt_npcs, sem_preds, sem_labels, proposal_offsets, P, sym_of_class, mats, info,
                     best, arg, cnt, rows.dev);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(npcs_loss_finish_kernel, dim3(1), dim3(1024), 0, stream, best, cnt, P, loss, n_has, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

static int npcs_loss_bwd_impl(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                              const int64_t* sem_labels, const int64_t* proposal_indices, int64_t M, const gpn::DevRows& rows, int64_t P,
                              const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                              const int32_t* type_group, int n_types, const void* scratch, const float* grad_loss, float* d_logits,
                              hipStream_t stream);
extern "C" int gpn_npcs_loss_bwd(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                                 const int64_t* sem_labels, const int64_t* proposal_indices, int64_t M, int64_t P,
                                 const int64_t* sym_of_class, const float* mats, const int32_t* type_first,
                                 const int32_t* type_count, const int32_t* type_group, int n_types, const void* scratch,
                                 const float* grad_loss, float* d_logits, gpn_stream_t stream_) {
  return npcs_loss_bwd_impl(logits, n_cls3, gt_npcs, sem_preds, sem_labels, proposal_indices, M, gpn::DevRows(), P, sym_of_class, mats,
                            type_first, type_count, type_group, n_types, scratch, grad_loss, d_logits, (hipStream_t)stream_);
}
// point count on the device (M = the bound of the per-point arrays; P = the bound the forward call's scratch was laid out for)
extern "C" int gpn_npcs_loss_bwd_dev(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                                     const int64_t* sem_labels, const int64_t* proposal_indices, int64_t M, const int64_t* m_dev,
                                     int64_t m_plan, int64_t P, const int64_t* sym_of_class, const float* mats,
                                     const int32_t* type_first, const int32_t* type_count, const int32_t* type_group, int n_types,
                                     const void* scratch, const float* grad_loss, float* d_logits, gpn_stream_t stream_) {
  GPN_CHECK_ARG(m_dev != nullptr);
  return npcs_loss_bwd_impl(logits, n_cls3, gt_npcs, sem_preds, sem_labels, proposal_indices, M, gpn::DevRows{m_dev, m_plan}, P,
                            sym_of_class, mats, type_first, type_count, type_group, n_types, scratch, grad_loss, d_logits,
                            (hipStream_t)stream_);
}
static int npcs_loss_bwd_impl(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                              const int64_t* sem_labels, const int64_t* proposal_indices, int64_t M, const gpn::DevRows& rows, int64_t P,
                              const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                              const int32_t* type_group, int n_types, const void* scratch, const float* grad_loss, float* d_logits,
                              hipStream_t stream) {
  GPN_CHECK_ARG(M >= 0 && P >= 1 && n_types >= 1 && n_types <= 8);
  if (M == 0) return GPN_OK;
  GPN_CHECK_ARG(logits && gt_npcs && sem_preds && sem_labels && proposal_indices && scratch && grad_loss && d_logits);
  NpcsTypeInfo info;
  info.n_types = n_types;
  for (int t = 0; t < n_types; ++t) info.first[t] = type_first[t], info.count[t] = type_count[t], info.group[t] = type_group[t];
  const float* best = static_cast<const float*>(scratch);
  const int32_t* arg = reinterpret_cast<const int32_t*>(best + P * 3);
  const int32_t* cnt = arg + P * 3;
  const float* n_has = reinterpret_cast<const float*>(cnt + P * 3);
  hipLaunchKernelGGL(npcs_loss_bwd_kernel, dim3(gpn::dev_grid(gpn::cdiv(M, 256), gpn::cdiv(gpn::plan_rows(M, rows), 256), rows.dev != nullptr)),
                     dim3(256), 0, stream, logits, n_cls3, gt_npcs, sem_preds, sem_labels, proposal_indices, M, sym_of_class, mats, info,
                     arg, cnt, n_has, grad_loss, d_logits, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}


// ================================================================================================ score loss
// Proposal score loss (network/model.py:348-396 of the reference: loss_proposal_score over get_gt_scores,
// grouping_utils.py:144-156, and the class selection of model.py:560-566) in one launch:
//   cls_p   = class of the proposal's first point (sem_labels if given, else sem_preds)
//   l_p     = logits[p, cls_p - 1]
//   t_p     = soft target of max_k ious[p, k]: 0 below bg, 1 above fg, iou * k + b in between (mul, then add: the reference's
//             two roundings)
//   loss    = mean_p ( max(l, 0) - l t + log1p(exp(-|l|)) )        (binary_cross_entropy_with_logits, mean)
// and what the rest of the step wants from the same values: score_preds[p] = sigmoid(l_p) and d loss / d logits (zero except
// column cls_p - 1: (sigmoid(l_p) - t_p) / P), so that backward is one multiplication by the upstream gradient.
// As torch ops this was ~18 launches forward and ~6 backward on a few hundred elements, issued while the GPU waits for the host.
// One workgroup; sums per thread in double, fixed-order tree: deterministic.
namespace {

__global__ __launch_bounds__(kThreads) void score_loss_kernel(const float* __restrict__ logits, int C1,
                                                              const int64_t* __restrict__ cls64, const int32_t* __restrict__ cls32,
                                                              const int32_t* __restrict__ offsets, const float* __restrict__ ious,
                                                              int I, int64_t P, float fg, float bg, float k, float b,
                                                              float* __restrict__ loss, float* __restrict__ score_preds,
                                                              float* __restrict__ d_logits, const int64_t* __restrict__ p_dev) {
  __shared__ double scratch[4];
  double acc = 0.0, bad = 0.0;
  P = gpn::live_rows(p_dev, P);
  if (P == 0) {  // (only with a device-counted P: no proposal survived - the term is absent from the step, model.py:560)
    if (threadIdx.x == 0) loss[0] = 0.f;
    return;
  }
  const float inv_p = 1.0f / (float)P;
  for (int64_t p = threadIdx.x; p < P; p += kThreads) {
    const int32_t first = offsets[p];
    const int64_t cls = cls64 ? cls64[first] : (int64_t)cls32[first];
    const int col = (int)(cls - 1);
    // a proposal whose class is outside 1 .. C1 (e.g. a ground-truth label 0 under its first point): the reference's
    // gather(1, cls - 1) raises an index error there.  No host read here, so the failure is made loud instead: the loss
    // comes out NaN (and the gradient of that proposal stays zero) rather than a silently different number.
    if (col < 0 || col >= C1) bad += 1.0;
    float iou = ious[p * I];
    for (int q = 1; q < I; ++q) iou = fmaxf(iou, ious[p * I + q]);
    const bool is_fg = iou > fg;
    const bool mid = !(is_fg || iou < bg);
    const float t = mid ? __fadd_rn(__fmul_rn(iou, k), b) : (is_fg ? 1.f : 0.f);
    for (int c = 0; c < C1; ++c) d_logits[p * C1 + c] = 0.f;
    float l = 0.f;
    if (col >= 0 && col < C1) l = logits[p * C1 + col];
    const float sig = 1.0f / (1.0f + expf(-l));
    score_preds[p] = sig;
    acc += (double)(fmaxf(l, 0.f) - l * t + log1pf(expf(-fabsf(l))));
    if (col >= 0 && col < C1) d_logits[p * C1 + col] = (sig - t) * inv_p;
  }
  const double total = block_sum(acc, scratch);
  __syncthreads();
  const double n_bad = block_sum(bad, scratch);
  if (threadIdx.x == 0) loss[0] = n_bad > 0.0 ? __builtin_nanf("") : (float)(total / (double)P);
}

}  // namespace

// logits [P, C1] f32; cls_i64 / cls_i32: per proposal-point class (exactly one non-NULL); offsets [P+1] i32 (CSR of the
// proposals); ious [P, I] f32 (gpn_instance_iou).  Outputs: loss [1], score_preds [P], d_logits [P, C1] (fully written).
static int score_loss_impl(const float* logits, int C1, const int64_t* cls_i64, const int32_t* cls_i32, const int32_t* offsets,
                           const float* ious, int I, int64_t P, const int64_t* p_dev, float fg_thresh, float bg_thresh, float* loss,
                           float* score_preds, float* d_logits, hipStream_t stream);
extern "C" int gpn_score_loss(const float* logits, int C1, const int64_t* cls_i64, const int32_t* cls_i32, const int32_t* offsets,
                              const float* ious, int I, int64_t P, float fg_thresh, float bg_thresh, float* loss,
                              float* score_preds, float* d_logits, gpn_stream_t stream_) {
  return score_loss_impl(logits, C1, cls_i64, cls_i32, offsets, ious, I, P, nullptr, fg_thresh, bg_thresh, loss, score_preds, d_logits,
                         (hipStream_t)stream_);
}
// proposal count on the device (P = the bound of the per-proposal arrays); *p_dev == 0 gives loss 0 and writes nothing else
extern "C" int gpn_score_loss_dev(const float* logits, int C1, const int64_t* cls_i64, const int32_t* cls_i32, const int32_t* offsets,
                                  const float* ious, int I, int64_t P, const int64_t* p_dev, float fg_thresh, float bg_thresh,
                                  float* loss, float* score_preds, float* d_logits, gpn_stream_t stream_) {
  GPN_CHECK_ARG(p_dev != nullptr);
  return score_loss_impl(logits, C1, cls_i64, cls_i32, offsets, ious, I, P, p_dev, fg_thresh, bg_thresh, loss, score_preds, d_logits,
                         (hipStream_t)stream_);
}
static int score_loss_impl(const float* logits, int C1, const int64_t* cls_i64, const int32_t* cls_i32, const int32_t* offsets,
                           const float* ious, int I, int64_t P, const int64_t* p_dev, float fg_thresh, float bg_thresh, float* loss,
                           float* score_preds, float* d_logits, hipStream_t stream) {
  GPN_CHECK_ARG(P >= 1 && C1 >= 1 && I >= 1 && fg_thresh > bg_thresh);
  GPN_CHECK_ARG(logits && offsets && ious && loss && score_preds && d_logits && ((cls_i64 != nullptr) != (cls_i32 != nullptr)));
  // (the reference computes k and b in Python doubles and multiplies / adds float tensors by them: rounded to float here)
  const float k = (float)(1.0 / ((double)fg_thresh - (double)bg_thresh));
  const float b = (float)((double)bg_thresh / ((double)bg_thresh - (double)fg_thresh));
  hipLaunchKernelGGL(score_loss_kernel, dim3(1), dim3(kThreads), 0, stream, logits, C1, cls_i64, cls_i32, offsets, ious, I, P,
                     fg_thresh, bg_thresh, k, b, loss, score_preds, d_logits, p_dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
