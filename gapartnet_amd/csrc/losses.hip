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
    double* __restrict__ partial /* [blocks][6] */) {
  __shared__ double scratch[4];
  double focal = 0.0, dice = 0.0, dist = 0.0, dir = 0.0, keep = 0.0, on = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < M; i += (int64_t)gridDim.x * kThreads) {
    const int64_t lab = labels[i];
    float p[kMaxClasses];
    const float* z = logits + i * C;
    const float lse = row_softmax(z, C, p);
    const bool has_class = lab >= 0 && lab < C;
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
  double vals[6] = {focal, dice, dist, dir, keep, on};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double s = block_sum(vals[k], scratch);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 6 + k] = s;
  }
}

__global__ void point_losses_finalize_kernel(const double* __restrict__ partial, int blocks, int64_t M,
                                             float* __restrict__ losses /* [4] */, Stats* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int b = 0; b < blocks; ++b)
    for (int k = 0; k < 6; ++k) s[k] += partial[(int64_t)b * 6 + k];
  losses[0] = s[4] > 0 ? (float)(s[0] / s[4]) : 0.f;  // focal: 0 for an all-ignored batch
  losses[1] = (float)(s[1] / (double)M);
  losses[2] = (float)(s[2] / s[5]);                   // NaN when no point is on a part (as x[mask].mean())
  losses[3] = (float)(s[3] / s[5]);
  stats->keep_count = s[4];
  stats->on_count = s[5];
  stats->points = (double)M;
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
  return gpn::align_up(sizeof(Stats)) + gpn::align_up((size_t)kMaxBlocks * 6 * sizeof(double));
}

// losses [4] f32 = (focal, dice, offset distance, offset direction); stats_out = opaque 32-byte device block that the
// backward call needs (counts of the selections)
extern "C" int gpn_point_losses_fwd(const float* logits, const int64_t* labels, const float* offsets,
                                    const float* gt_offsets, const int32_t* instance_labels, int64_t M, int C,
                                    int64_t ignore_index, float* losses, void* stats_out, void* ws, size_t ws_bytes,
                                    gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(M >= 1 && C >= 1 && C <= kMaxClasses);
  GPN_CHECK_ARG(logits && labels && offsets && gt_offsets && instance_labels && losses && stats_out && ws);
  GPN_CHECK_ARG(ws_bytes >= (size_t)kMaxBlocks * 6 * sizeof(double));
  const int blocks = fwd_blocks(M);
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(point_losses_fwd_kernel, dim3(blocks), dim3(kThreads), 0, stream, logits, labels, offsets,
                     gt_offsets, instance_labels, M, C, ignore_index, partial);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(point_losses_finalize_kernel, dim3(1), dim3(64), 0, stream, partial, blocks, M, losses,
                     static_cast<Stats*>(stats_out));
  GPN_CHECK_LAUNCH();
  return GPN_OK;
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
