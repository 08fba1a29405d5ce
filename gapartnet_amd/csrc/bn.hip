// bn.hip — BatchNorm1d over sparse-tensor feature matrices [N, C], fused with the residual add and ReLU that follow
// it in every block of the reference network (network/backbone.py:40-49: relu(bn(conv(x)) [+ shortcut])).
//
// HBM-streaming kernels: [N, C] row-major with C in 16..224, read as float4.  Training forward = statistics pass
// (<= 128 workgroups x 1024 threads, per-workgroup partial sums in double for a cancellation-safe variance) + one apply
// pass that also adds the residual and applies ReLU.  Backward = one reduction pass (sum g, sum g*xhat with the ReLU
// mask folded in) + one apply pass producing dx (and the residual's gradient).  There is no finalize launch: every
// workgroup of the apply pass folds the <= 128 partials itself (L2-resident, ~1 us) and workgroup 0 also stores mean /
// 1/std / running statistics (forward) or dweight / dbias (backward) - a 64-thread finalize launch between the two
// passes cost 4.5 us of GPU time and one host launch per layer and pass (176 per training step of the default model).  Compared with separate BatchNorm / add / ReLU kernels this removes three full read+write passes per
// layer in forward and two in backward.  All reductions are fixed-order (deterministic).
//
// Inside the network executor (net.hip) most BatchNorms do not run their statistics / reduction pass at all: the conv (or
// dgrad) launch that produces their input (or output gradient) accumulates the column sums in its epilogue as
// order-independent fixed-point integers (bn_stats.h), and only the apply pass of this file runs
// (gpn::bn_fwd_train_fused / bn_bwd_fused: every workgroup folds the <= 32 slot sets, the first batch of elements is
// requested before the fold, predicated batches of 4 loads; a launch can carry a second BatchNorm of the same shape for the
// executor's paired passes).
//
// Small matrices (N <= kSmallRows: the deep levels of the U-Net, where a layer's kernels run at the
// launch-latency floor) take a single-launch form instead: one workgroup per float4 column computes the statistics of
// its four channels and applies them in a second sweep over the (L2-resident) column - one launch instead of three.
#include <cstdlib>

#include "bn_stats.h"
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 128;  // statistics-pass workgroups = partials every apply workgroup folds

// thread layout for column-wise reductions: c4 = tid % C4 (float4 column), r = tid / C4 (row lane), R = T / C4 rows

// ordered sum of red[q][rr * C4 + c/4][c%4] over rr < R for (q, c) pairs, P sub-lanes per pair (strided partial sums, then
// a fixed-order shuffle tree): deterministic.  Returns the sum in sub-lane 0 of each pair; e = pair index of this thread.
template <int T>
__device__ __forceinline__ double column_total(const double (*red)[T][4], int q, int c, int C4, int R, int part, int P) {
  double acc = 0.0;
  const int cc4 = c >> 2, j = c & 3;
  for (int rr = part; rr < R; rr += P) acc += red[q][rr * C4 + cc4][j];
  for (int off = P >> 1; off >= 1; off >>= 1) acc += __shfl_down(acc, off, P);
  return acc;
}

__host__ __device__ __forceinline__ int sub_lanes(int pairs, int T) {
  int P = 16;
  while (P > 1 && pairs * P > T) P >>= 1;
  return P;
}

constexpr int kReduceThreads = 1024;

template <bool BWD>
__global__ __launch_bounds__(kReduceThreads) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ dy, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, int64_t N, int C4, int relu,
                                                                   double* __restrict__ partial /* [blocks][2][C] */,
                                                                   const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);  // (a device-counted row count: every workgroup still writes its - possibly zero - partial)
  extern __shared__ __attribute__((aligned(16))) double red_raw[];
  double (*red)[kReduceThreads][4] = reinterpret_cast<double (*)[kReduceThreads][4]>(red_raw);  // [2][T][4]
  constexpr int T = kReduceThreads;
  const int R = T / C4;
  const int r = threadIdx.x / C4, c4 = threadIdx.x - r * C4;
  const bool lane_ok = r < R;
  const int C = C4 * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
  if (BWD && lane_ok) {
    mu = reinterpret_cast<const f32x4*>(mean)[c4];
    is = reinterpret_cast<const f32x4*>(invstd)[c4];
  }
  // rows are dealt to workgroups in contiguous chunks (fixed assignment => fixed summation order)
  const int64_t rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int64_t row_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row_end = row_begin + rows_per_block < N ? row_begin + rows_per_block : N;
  if (lane_ok) {
#pragma unroll 4
    for (int64_t row = row_begin + r; row < row_end; row += R) {
      const f32x4 xv = reinterpret_cast<const f32x4*>(x)[row * C4 + c4];
      if (!BWD) {
        s0 += xv;
        s1 += xv * xv;
      } else {
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[row * C4 + c4];
        if (relu) {
          const f32x4 yv = reinterpret_cast<const f32x4*>(y)[row * C4 + c4];
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
        }
        s0 += g;
        s1 += g * ((xv - mu) * is);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][threadIdx.x][j] = (double)s0[j];
    red[1][threadIdx.x][j] = (double)s1[j];
  }
  __syncthreads();
  const int P = sub_lanes(2 * C, T);
  const int part = threadIdx.x % P;
  for (int e = threadIdx.x / P; e < 2 * C; e += T / P) {  // whole sub-lane groups iterate together (P divides 64); C > T / 2: several rounds
    const int q = e / C, c = e - q * C;
    const double acc = column_total<T>(red, q, c, C4, R, part, P);
    if (part == 0) partial[((int64_t)blockIdx.x * 2 + q) * C + c] = acc;
  }
}

constexpr int kApplyThreads = 512;
constexpr int kFoldMaxC = 256;

// every workgroup of an apply pass folds the statistics pass's partials [blocks][2][C] itself: row lanes stride over the
// partials, then the same ordered column sum as above.  s / ss (both quantities of channel c) land in sums[0][c] / [1][c].
__device__ __forceinline__ void fold_partials(const double* __restrict__ partial, int blocks, int C4,
                                              double (*red)[kApplyThreads][4], double (*sums)[kFoldMaxC]) {
  constexpr int T = kApplyThreads;
  const int C = C4 * 4, R = T / C4;
  const int r = threadIdx.x / C4, c4 = threadIdx.x - r * C4;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (r < R) {
    for (int b = r; b < blocks; b += R) {
      const double* p = partial + (int64_t)b * 2 * C + c4 * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += p[j], a[4 + j] += p[C + j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[0][threadIdx.x][j] = a[j], red[1][threadIdx.x][j] = a[4 + j];
  __syncthreads();
  const int P = sub_lanes(2 * C, T);
  const int e = threadIdx.x / P, part = threadIdx.x - e * P;
  const int Rb = R < blocks ? R : blocks;  // row lanes past the partial count hold zeros
  if (e < 2 * C) {
    const int q = e / C, c = e - q * C;
    const double acc = column_total<T>(red, q, c, C4, Rb, part, P);
    if (part == 0) sums[q][c] = acc;
  }
  __syncthreads();
}

// the same totals from a slab of fixed-point words a conv launch accumulated (bn_stats.h).  The slab [slots][4][C] is read
// linearly by the whole workgroup (coalesced, every load independent: one memory round trip for C <= 32, the serial
// 32-slot loop per word this replaces was four) and summed per word with LDS integer atomics - integer addition, so the
// order the threads arrive in does not matter; thread c then combines the four exact sums of its channel.
template <bool BWD>
__device__ __forceinline__ void fold_fixed(const unsigned long long* __restrict__ slab, int C, int slots, unsigned long long (*words)[kFoldMaxC],
                                           double (*sums)[kFoldMaxC]) {
  constexpr gpn::StatScale sc = BWD ? gpn::kStatScaleBwd() : gpn::kStatScaleFwd();
  const int W = 4 * C;  // words per slot set
  for (int e = threadIdx.x; e < W; e += kApplyThreads) words[e / C][e % C] = 0ull;
  __syncthreads();
  const int total = W * slots;  // gpn::stat_slot_count(N) sets of the slab are in use
  int w = threadIdx.x % W;              // word of this thread's first element; advances by kApplyThreads % W per step
  const int step = kApplyThreads % W;
#pragma unroll 8
  for (int e = threadIdx.x; e < total; e += kApplyThreads) {
    const unsigned long long v = slab[e];
    if (v) atomicAdd(&words[0][0] + (w / C) * kFoldMaxC + (w % C), v);
    w += step;
    w = w >= W ? w - W : w;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kApplyThreads) {
    sums[0][c] = ldexp((double)(long long)words[0][c], -sc.h0) + ldexp((double)(long long)words[1][c], -sc.l0);
    sums[1][c] = ldexp((double)(long long)words[2][c], -sc.h1) + ldexp((double)(long long)words[3][c], -sc.l1);
  }
  __syncthreads();
}

// one 64-thread workgroup per channel sums the per-workgroup partials: strided per-lane sums, then a fixed-order
// shuffle tree (deterministic); a serial loop over up to 512 partials per channel cost 25 us per layer
__device__ __forceinline__ void sum_partials(const double* __restrict__ partial, int blocks, int C, int c, double& s,
                                             double& ss) {
  s = 0.0;
  ss = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 64) {
    s += partial[((int64_t)b * 2 + 0) * C + c];
    ss += partial[((int64_t)b * 2 + 1) * C + c];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_down(s, off, 64);
    ss += __shfl_down(ss, off, 64);
  }
}

// forward finalize: mean / invstd, running statistics (momentum; unbiased variance as torch.nn.BatchNorm1d)
__global__ __launch_bounds__(64) void bn_finalize_fwd_kernel(const double* __restrict__ partial, int blocks, int64_t N,
                                                             int C, float eps, float momentum, float* __restrict__ mean,
                                                             float* __restrict__ invstd, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var) {
  const int c = blockIdx.x;
  double s, ss;
  sum_partials(partial, blocks, C, c, s, ss);
  if (threadIdx.x != 0) return;
  const double m = s / (double)N;
  double var = ss / (double)N - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = N > 1 ? var * ((double)N / (double)(N - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// y = relu?( (x - mean) * invstd * w + b [+ res] )
__global__ void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ weight, const float* __restrict__ bias, int64_t total4,
                                    int C4, int relu, float* __restrict__ y, const int64_t* __restrict__ n_dev) {
  if (n_dev) total4 = gpn::live_rows(n_dev, total4 / C4) * C4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % C4);
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
    const f32x4 w = reinterpret_cast<const f32x4*>(weight)[c4], b = reinterpret_cast<const f32x4*>(bias)[c4];
    f32x4 v = (reinterpret_cast<const f32x4*>(x)[t] - mu) * is * w + b;
    if (res) v += reinterpret_cast<const f32x4*>(res)[t];
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(y)[t] = v;
  }
}

// eval forward of the executor: the same arithmetic with invstd = 1 / sqrt(running_var + eps) evaluated by the thread for its four
// channels (the value the separate one-workgroup kernel wrote before - 103 launches of 3.7 us per validation step, round 5);
// written to save_invstd by the first workgroup for a backward pass over frozen statistics.  Two pointer sets (blockIdx.y):
// ScoreNet and NPCS-Net in one launch.
__global__ void bn_apply_eval_kernel(const gpn::BnFwdPtrs pa, const gpn::BnFwdPtrs pb, int64_t total4, int C4, float eps, int relu,
                                     const int64_t* __restrict__ n_dev) {
  const gpn::BnFwdPtrs& p = blockIdx.y ? pb : pa;
  if (n_dev) total4 = gpn::live_rows(n_dev, total4 / C4) * C4;
  if (blockIdx.x == 0 && p.invstd)
    for (int c = threadIdx.x; c < 4 * C4; c += blockDim.x) p.invstd[c] = 1.0f / sqrtf(p.running_var[c] + eps);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % C4);
    const f32x4 mu = reinterpret_cast<const f32x4*>(p.running_mean)[c4], var = reinterpret_cast<const f32x4*>(p.running_var)[c4];
    f32x4 is;
#pragma unroll
    for (int j = 0; j < 4; ++j) is[j] = 1.0f / sqrtf(var[j] + eps);
    const f32x4 w = reinterpret_cast<const f32x4*>(p.weight)[c4], b = reinterpret_cast<const f32x4*>(p.bias)[c4];
    f32x4 v = (reinterpret_cast<const f32x4*>(p.x)[t] - mu) * is * w + b;
    if (p.res) v += reinterpret_cast<const f32x4*>(p.res)[t];
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(p.y)[t] = v;
  }
}

// backward finalize: dweight = sum g*xhat, dbias = sum g
__global__ __launch_bounds__(64) void bn_finalize_bwd_kernel(const double* __restrict__ partial, int blocks, int C,
                                                             float* __restrict__ dweight, float* __restrict__ dbias) {
  const int c = blockIdx.x;
  double s, ss;
  sum_partials(partial, blocks, C, c, s, ss);
  if (threadIdx.x != 0) return;
  dbias[c] = (float)s;
  dweight[c] = (float)ss;
}

// dx = invstd * w * (g - dbias/N - xhat * dweight/N)   (training)   |   dx = invstd * w * g   (eval);  dres = g
__global__ void bn_apply_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ weight, const float* __restrict__ dweight,
                                    const float* __restrict__ dbias, int64_t total4, int C4, float inv_n, int relu,
                                    int training, float* __restrict__ dx, float* __restrict__ dres) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % C4);
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
    const f32x4 w = reinterpret_cast<const f32x4*>(weight)[c4];
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[t];
    if (relu) {
      const f32x4 yv = reinterpret_cast<const f32x4*>(y)[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
    }
    if (dres) reinterpret_cast<f32x4*>(dres)[t] = g;
    f32x4 v = g;
    if (training) {
      const f32x4 dw = reinterpret_cast<const f32x4*>(dweight)[c4], db = reinterpret_cast<const f32x4*>(dbias)[c4];
      const f32x4 xhat = (reinterpret_cast<const f32x4*>(x)[t] - mu) * is;
      v = g - db * inv_n - xhat * (dw * inv_n);
    }
    reinterpret_cast<f32x4*>(dx)[t] = v * is * w;
  }
}

// ---- apply passes that fold the finalize (training; C <= kFoldMaxC) ------------------------------------------------------
// thread t walks elements t, t + G*T, ...: its float4 column advances by (G*T) % C4 per step (no 64-bit modulo in the loop)
constexpr int kApplyBatch = 4;  // elements of a thread in flight together

template <bool FIXED>  // FIXED: `partial` is a slab of fixed-point words a conv launch accumulated (bn_stats.h), `blocks` its slot sets in use
__global__ __launch_bounds__(kApplyThreads) void bn_apply_fwd_fold_kernel(gpn::BnFwdPtrs pa, gpn::BnFwdPtrs pb, int blocks, int64_t N,
                                                                         int64_t total4, int C4, float eps, float momentum,
                                                                         int relu, const int64_t* __restrict__ n_dev) {
  if (n_dev) {  // the row count is a device counter (gpn::DevRows); no rows: nothing to normalise, statistics untouched
    N = gpn::live_rows(n_dev, N);
    total4 = N * C4;
    if (N == 0) return;
  }
  // (two BatchNorms of the same shape per launch for the executor's paired passes: blockIdx.y picks the pointer set)
  const gpn::BnFwdPtrs& pp = blockIdx.y ? pb : pa;
  const float* __restrict__ x = pp.x;
  const float* __restrict__ res = pp.res;
  const void* __restrict__ partial = pp.partial;
  const float* __restrict__ weight = pp.weight;
  const float* __restrict__ bias = pp.bias;
  float* __restrict__ y = pp.y;
  float* __restrict__ mean = pp.mean;
  float* __restrict__ invstd = pp.invstd;
  float* __restrict__ running_mean = pp.running_mean;
  float* __restrict__ running_var = pp.running_var;
  // (the fused form needs 8 KB for its integer fold, not the 32 KB of the partial-sum fold: 17 instead of 44 KB of LDS per
  // workgroup - a workgroup then still finds room on a CU whose LDS the weight-gradient stream's workgroups have filled)
  __shared__ double red[2][FIXED ? 1 : kApplyThreads][4];
  __shared__ unsigned long long words[FIXED ? 4 : 1][FIXED ? kFoldMaxC : 1];
  __shared__ double sums[2][kFoldMaxC];
  __shared__ __attribute__((aligned(16))) float stat[4][kFoldMaxC];  // mean, 1/std, weight, bias
  constexpr int U = kApplyBatch;
  const int C = C4 * 4;
  // thread t walks elements t, t + G*T, ... in batches of U whose loads are all in flight together (predicated: no
  // remainder loop that would take one round trip per element); the first batch is requested BEFORE the fold, so its latency
  // overlaps the fold's own round trip - the small layers of the step are one batch per thread and pay this kernel's floor
  const int64_t stride = (int64_t)gridDim.x * kApplyThreads;
  const int step = (int)(stride % C4);
  int64_t t = (int64_t)blockIdx.x * kApplyThreads + threadIdx.x;
  int c4 = (int)(t % C4);
  f32x4 xv[U], rv[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int64_t tt = t + k * stride;
    if (tt < total4) {
      xv[k] = reinterpret_cast<const f32x4*>(x)[tt];
      if (res) rv[k] = reinterpret_cast<const f32x4*>(res)[tt];
    }
  }
  if constexpr (FIXED)
    fold_fixed<false>(static_cast<const unsigned long long*>(partial), C, blocks, reinterpret_cast<unsigned long long (*)[kFoldMaxC]>(&words[0][0]), sums);
  else
    fold_partials(static_cast<const double*>(partial), blocks, C4, reinterpret_cast<double (*)[kApplyThreads][4]>(&red[0][0][0]), sums);
  for (int c = threadIdx.x; c < C; c += kApplyThreads) {
    const double m = sums[0][c] / (double)N;
    double var = sums[1][c] / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    stat[0][c] = mu;
    stat[1][c] = is;
    stat[2][c] = weight[c];
    stat[3][c] = bias[c];
    if (blockIdx.x == 0) {
      mean[c] = mu;
      invstd[c] = is;
      if (running_mean) {
        const double unbiased = N > 1 ? var * ((double)N / (double)(N - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
      }
    }
  }
  __syncthreads();
  while (true) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tt = t + k * stride;
      if (tt < total4) {
        const f32x4 mu = reinterpret_cast<const f32x4*>(stat[0])[c4], is = reinterpret_cast<const f32x4*>(stat[1])[c4];
        const f32x4 w = reinterpret_cast<const f32x4*>(stat[2])[c4], b = reinterpret_cast<const f32x4*>(stat[3])[c4];
        f32x4 v = (xv[k] - mu) * is * w + b;
        if (res) v += rv[k];
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        reinterpret_cast<f32x4*>(y)[tt] = v;
      }
      c4 += step;
      c4 = c4 >= C4 ? c4 - C4 : c4;
    }
    t += U * stride;
    if (t >= total4) break;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tt = t + k * stride;
      if (tt < total4) {
        xv[k] = reinterpret_cast<const f32x4*>(x)[tt];
        if (res) rv[k] = reinterpret_cast<const f32x4*>(res)[tt];
      }
    }
  }
}

template <bool FIXED>
__global__ __launch_bounds__(kApplyThreads) void bn_apply_bwd_fold_kernel(gpn::BnBwdPtrs pa, gpn::BnBwdPtrs pb, int blocks, int64_t total4,
                                                                         int C4, float inv_n, int relu, int training,
                                                                         const int64_t* __restrict__ n_dev) {
  if (n_dev) {  // the row count is a device counter (gpn::DevRows): total4 was its bound x C4
    const int64_t N = gpn::live_rows(n_dev, total4 / C4);
    total4 = N * C4;
    inv_n = 1.0f / (float)N;
    if (N == 0) {  // no rows: zero parameter gradients, nothing else
      if (blockIdx.x == 0) {
        const gpn::BnBwdPtrs& q = blockIdx.y ? pb : pa;
        for (int c = threadIdx.x; c < C4 * 4; c += kApplyThreads) q.dbias[c] = 0.f, q.dweight[c] = 0.f;
      }
      return;
    }
  }
  const gpn::BnBwdPtrs& pp = blockIdx.y ? pb : pa;
  const float* __restrict__ x = pp.x;
  const float* __restrict__ y = pp.y;
  const float* __restrict__ dy = pp.dy;
  const void* __restrict__ partial = pp.partial;
  const float* __restrict__ mean = pp.mean;
  const float* __restrict__ invstd = pp.invstd;
  const float* __restrict__ weight = pp.weight;
  float* __restrict__ dx = pp.dx;
  float* __restrict__ dres = pp.dres;
  float* __restrict__ dweight = pp.dweight;
  float* __restrict__ dbias = pp.dbias;
  __shared__ double red[2][FIXED ? 1 : kApplyThreads][4];
  __shared__ unsigned long long words[FIXED ? 4 : 1][FIXED ? kFoldMaxC : 1];
  __shared__ double sums[2][kFoldMaxC];
  __shared__ __attribute__((aligned(16))) float stat[5][kFoldMaxC];  // dbias, dweight, mean, 1/std, weight
  constexpr int U = kApplyBatch;
  const int C = C4 * 4;
  const int64_t stride = (int64_t)gridDim.x * kApplyThreads;
  const int step = (int)(stride % C4);
  int64_t t = (int64_t)blockIdx.x * kApplyThreads + threadIdx.x;
  int c4 = (int)(t % C4);
  // first batch requested before the fold (see the forward kernel)
  f32x4 gv[U], yv[U], xv[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int64_t tt = t + k * stride;
    if (tt < total4) {
      gv[k] = reinterpret_cast<const f32x4*>(dy)[tt];
      if (relu) yv[k] = reinterpret_cast<const f32x4*>(y)[tt];
      if (training) xv[k] = reinterpret_cast<const f32x4*>(x)[tt];
    }
  }
  if constexpr (FIXED)
    fold_fixed<true>(static_cast<const unsigned long long*>(partial), C, blocks, reinterpret_cast<unsigned long long (*)[kFoldMaxC]>(&words[0][0]), sums);
  else
    fold_partials(static_cast<const double*>(partial), blocks, C4, reinterpret_cast<double (*)[kApplyThreads][4]>(&red[0][0][0]), sums);
  for (int c = threadIdx.x; c < C; c += kApplyThreads) {
    const float db = (float)sums[0][c], dw = (float)sums[1][c];
    stat[0][c] = db;
    stat[1][c] = dw;
    stat[2][c] = mean[c];
    stat[3][c] = invstd[c];
    stat[4][c] = weight[c];
    if (blockIdx.x == 0) {
      dbias[c] = db;
      dweight[c] = dw;
    }
  }
  __syncthreads();
  while (true) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tt = t + k * stride;
      if (tt < total4) {
        const f32x4 is = reinterpret_cast<const f32x4*>(stat[3])[c4], w = reinterpret_cast<const f32x4*>(stat[4])[c4];
        f32x4 g = gv[k];
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = yv[k][j] > 0.f ? g[j] : 0.f;
        }
        if (dres) reinterpret_cast<f32x4*>(dres)[tt] = g;
        f32x4 v = g;
        if (training) {
          const f32x4 db = reinterpret_cast<const f32x4*>(stat[0])[c4], dw = reinterpret_cast<const f32x4*>(stat[1])[c4];
          const f32x4 mu = reinterpret_cast<const f32x4*>(stat[2])[c4];
          const f32x4 xhat = (xv[k] - mu) * is;
          v = g - db * inv_n - xhat * (dw * inv_n);
        }
        reinterpret_cast<f32x4*>(dx)[tt] = v * is * w;
      }
      c4 += step;
      c4 = c4 >= C4 ? c4 - C4 : c4;
    }
    t += U * stride;
    if (t >= total4) break;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t tt = t + k * stride;
      if (tt < total4) {
        gv[k] = reinterpret_cast<const f32x4*>(dy)[tt];
        if (relu) yv[k] = reinterpret_cast<const f32x4*>(y)[tt];
        if (training) xv[k] = reinterpret_cast<const f32x4*>(x)[tt];
      }
    }
  }
}

constexpr int kSmallRows = 1024;

// Small-N layout: a workgroup owns CG adjacent float4 columns (CG = 4: 64 contiguous bytes per row) and 256 / CG row
// lanes; thread t -> column t % CG, row lane t / CG.
// fixed-order sums of eight values per thread over the threads that share a column, all eight in one tree (one pair
// of barriers); results valid in every thread of that column
template <int CG, int T>
__device__ __forceinline__ void column_sum8(double (&v)[8], double (*scratch)[CG][8] /* [T / 64][CG][8] */) {
  constexpr int W = T / 64;
#pragma unroll
  for (int off = 32; off >= CG; off >>= 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] += __shfl_down(v[q], off, 64);
  }
  if ((threadIdx.x & 63) < CG) {
#pragma unroll
    for (int q = 0; q < 8; ++q) scratch[threadIdx.x >> 6][threadIdx.x & 63][q] = v[q];
  }
  __syncthreads();
  const int cg = threadIdx.x % CG;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    double t[W];
#pragma unroll
    for (int w = 0; w < W; ++w) t[w] = scratch[w][cg][q];
#pragma unroll
    for (int step = 1; step < W; step <<= 1) {  // fixed pairwise tree over the waves
#pragma unroll
      for (int w = 0; w + step < W; w += 2 * step) t[w] += t[w + step];
    }
    v[q] = t[0];
  }
}

// single-launch training forward for small N
template <int CG, int T>
__global__ __launch_bounds__(T) void bn_small_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ weight,
    const float* __restrict__ bias, int N, int C4, float eps, float momentum, int relu, float* __restrict__ y,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, const int64_t* __restrict__ n_dev) {
  if (n_dev) {
    N = (int)gpn::live_rows(n_dev, N);
    if (N == 0) return;
  }
  __shared__ double scratch[T / 64][CG][8];
  constexpr int R = T / CG;
  const int c4 = blockIdx.x * CG + threadIdx.x % CG;
  const int rl = threadIdx.x / CG;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int row = rl; row < N; row += R) {
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[(int64_t)row * C4 + c4];
    s0 += xv;
    s1 += xv * xv;
  }
  double sums[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) sums[j] = (double)s0[j], sums[4 + j] = (double)s1[j];
  column_sum8<CG, T>(sums, scratch);
  f32x4 mu, is;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double s = sums[j], ss = sums[4 + j];
    const double m = s / (double)N;
    double var = ss / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    mu[j] = (float)m;
    is[j] = (float)(1.0 / sqrt(var + (double)eps));
    if (rl == 0) {
      const int c = c4 * 4 + j;
      mean[c] = mu[j];
      invstd[c] = is[j];
      if (running_mean) {
        const double unbiased = N > 1 ? var * ((double)N / (double)(N - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
      }
    }
  }
  const f32x4 w = reinterpret_cast<const f32x4*>(weight)[c4], b = reinterpret_cast<const f32x4*>(bias)[c4];
#pragma unroll 8
  for (int row = rl; row < N; row += R) {
    const int64_t t = (int64_t)row * C4 + c4;
    f32x4 v = (reinterpret_cast<const f32x4*>(x)[t] - mu) * is * w + b;
    if (res) v += reinterpret_cast<const f32x4*>(res)[t];
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(y)[t] = v;
  }
}

// single-launch backward for small N (same layout): dweight / dbias, then dx (and dres)
template <int CG, int T>
__global__ __launch_bounds__(T) void bn_small_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ weight, const float* __restrict__ mean, const float* __restrict__ invstd, int N, int C4,
    int relu, int training, float* __restrict__ dx, float* __restrict__ dres, float* __restrict__ dweight,
    float* __restrict__ dbias, const int64_t* __restrict__ n_dev) {
  if (n_dev) N = (int)gpn::live_rows(n_dev, N);  // (N == 0: the sums below are zero, no row is written)
  __shared__ double scratch[T / 64][CG][8];
  constexpr int R = T / CG;
  const int c4 = blockIdx.x * CG + threadIdx.x % CG;
  const int rl = threadIdx.x / CG;
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int row = rl; row < N; row += R) {
    const int64_t t = (int64_t)row * C4 + c4;
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[t];
    if (relu) {
      const f32x4 yv = reinterpret_cast<const f32x4*>(y)[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
    }
    s0 += g;
    s1 += g * ((reinterpret_cast<const f32x4*>(x)[t] - mu) * is);
  }
  double sums[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) sums[j] = (double)s0[j], sums[4 + j] = (double)s1[j];
  column_sum8<CG, T>(sums, scratch);
  f32x4 db, dw;
#pragma unroll
  for (int j = 0; j < 4; ++j) db[j] = (float)sums[j], dw[j] = (float)sums[4 + j];
  if (rl == 0) {
    reinterpret_cast<f32x4*>(dbias)[c4] = db;
    reinterpret_cast<f32x4*>(dweight)[c4] = dw;
  }
  const f32x4 w = reinterpret_cast<const f32x4*>(weight)[c4];
  const float inv_n = 1.0f / (float)N;
#pragma unroll 8
  for (int row = rl; row < N; row += R) {
    const int64_t t = (int64_t)row * C4 + c4;
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[t];
    if (relu) {
      const f32x4 yv = reinterpret_cast<const f32x4*>(y)[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
    }
    if (dres) reinterpret_cast<f32x4*>(dres)[t] = g;
    f32x4 v = g;
    if (training) {
      const f32x4 xhat = (reinterpret_cast<const f32x4*>(x)[t] - mu) * is;
      v = g - db * inv_n - xhat * (dw * inv_n);
    }
    reinterpret_cast<f32x4*>(dx)[t] = v * is * w;
  }
}

int reduce_blocks(int64_t N, int C4) {
  const int R = kReduceThreads / C4;
  // two rows per thread: the mid-size levels are latency-bound (a thread's rows are a serial chain of loads), the large
  // ones need every CU they can get; 8 rows per thread made both ~2x slower (rocprofv3, profiles/r01_kernels_by_grid.csv)
  int64_t b = gpn::cdiv(N, (int64_t)R * 2);
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

int fold_grid(int64_t total4) {
  // one element per thread until every CU has a workgroup; the large levels then walk ~4 elements per thread
  int64_t g = gpn::cdiv(total4, (int64_t)kApplyThreads);
  if (g > 320) g = 320;  // (the two largest levels of the bench then fit one batch of kApplyBatch elements per thread)
  return (int)(g < 1 ? 1 : g);
}

constexpr size_t kReduceLds = (size_t)2 * kReduceThreads * 4 * sizeof(double);

int apply_grid(int64_t total4) {
  int64_t g = gpn::cdiv(total4, kThreads);
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" size_t gpn_bn_ws_bytes(int64_t N, int C) {
  (void)N;
  return gpn::align_up((size_t)kMaxBlocks * 2 * C * sizeof(double));
}

// training forward: batch statistics (saved in mean / invstd for backward), optional running-stat update
extern "C" int gpn_bn_fwd_train(const float* x, const float* res, const float* weight, const float* bias, int64_t N,
                                int C, float eps, float momentum, int relu, float* y, float* mean, float* invstd,
                                float* running_mean, float* running_var, void* ws, size_t ws_bytes,
                                gpn_stream_t stream_) {
  return gpn::bn_fwd_train_rows(x, res, weight, bias, N, gpn::DevRows(), C, eps, momentum, relu, y, mean, invstd, running_mean,
                                running_var, ws, ws_bytes, (hipStream_t)stream_);
}

// (rows.dev: N is the buffers' bound and the live row count a device counter - gpn::DevRows; variants are picked from the plan)
int gpn::bn_fwd_train_rows(const float* x, const float* res, const float* weight, const float* bias, int64_t N,
                           const gpn::DevRows& rows, int C, float eps, float momentum, int relu, float* y, float* mean,
                           float* invstd, float* running_mean, float* running_var, void* ws, size_t ws_bytes,
                           hipStream_t stream) {
  const int64_t Np = gpn::plan_rows(N, rows);
  GPN_CHECK_ARG(N >= 1 && C >= 4 && C % 4 == 0 && C <= 1024);
  GPN_CHECK_ARG(x && weight && bias && y && mean && invstd && ws);
  GPN_CHECK_ARG(ws_bytes >= (size_t)kMaxBlocks * 2 * C * sizeof(double));
  GPN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  const int C4 = C / 4;
  gpn::ProfScope prof(GPN_K_BN, stream, 0.0, 4.0 * (double)N * C * (res ? 4 : 3), rows.dev, N);  // x twice (statistics, apply) [+ res], y
  if (Np <= kSmallRows && N < ((int64_t)1 << 31)) {
    // (the forward kernel stays at 256 threads: at 1024 it measured 51 us instead of 10 - every thread carries the
    // double-precision mean / 1/sqrt epilogue; the backward kernel, three streams and no epilogue, gains from 1024)
    if (C4 % 4 == 0)
      hipLaunchKernelGGL((bn_small_fwd_kernel<4, 256>), dim3(C4 / 4), dim3(256), 0, stream, x, res, weight, bias, (int)N,
                         C4, eps, momentum, relu, y, mean, invstd, running_mean, running_var, rows.dev);
    else
      hipLaunchKernelGGL((bn_small_fwd_kernel<1, 256>), dim3(C4), dim3(256), 0, stream, x, res, weight, bias, (int)N, C4,
                         eps, momentum, relu, y, mean, invstd, running_mean, running_var, rows.dev);
    GPN_CHECK_LAUNCH();
    return GPN_OK;
  }
  const int blocks = reduce_blocks(Np, C4);
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(blocks), dim3(kReduceThreads), kReduceLds, stream, x, nullptr, nullptr,
                     nullptr, nullptr, N, C4, 0, partial, rows.dev);
  GPN_CHECK_LAUNCH();
  const int64_t total4 = N * C4, plan4 = Np * C4;
  if (C <= kFoldMaxC) {
    gpn::BnFwdPtrs pp;
    pp.x = x, pp.res = res, pp.partial = partial, pp.weight = weight, pp.bias = bias, pp.y = y, pp.mean = mean, pp.invstd = invstd,
    pp.running_mean = running_mean, pp.running_var = running_var;
    hipLaunchKernelGGL(bn_apply_fwd_fold_kernel<false>, dim3(fold_grid(plan4)), dim3(kApplyThreads), 0, stream, pp, pp, blocks, N,
                       total4, C4, eps, momentum, relu, rows.dev);
    GPN_CHECK_LAUNCH();
    return GPN_OK;
  }
  if (rows.dev) {
    gpn::set_error("gpn_bn_fwd_train: a device-counted row count needs C <= %d", kFoldMaxC);
    return GPN_ERR_ARG;
  }
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(C), dim3(64), 0, stream, partial, blocks, N, C,
                     eps, momentum, mean, invstd, running_mean, running_var);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(apply_grid(total4)), dim3(kThreads), 0, stream, x, res, mean, invstd,
                     weight, bias, total4, C4, relu, y, nullptr);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// eval forward: y = act((x - mean) * invstd * w + b [+ res]) with caller-provided mean / invstd
extern "C" int gpn_bn_fwd_eval(const float* x, const float* res, const float* weight, const float* bias,
                               const float* mean, const float* invstd, int64_t N, int C, int relu, float* y,
                               gpn_stream_t stream_) {
  return gpn::bn_fwd_eval_rows(x, res, weight, bias, mean, invstd, N, gpn::DevRows(), C, relu, y, (hipStream_t)stream_);
}

int gpn::bn_fwd_eval_rows(const float* x, const float* res, const float* weight, const float* bias, const float* mean,
                          const float* invstd, int64_t N, const gpn::DevRows& rows, int C, int relu, float* y, hipStream_t stream) {
  GPN_CHECK_ARG(N >= 0 && C >= 4 && C % 4 == 0);
  if (N == 0) return GPN_OK;
  GPN_CHECK_ARG(x && weight && bias && mean && invstd && y);
  const int64_t total4 = N * (C / 4);
  hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(apply_grid(gpn::plan_rows(N, rows) * (C / 4))), dim3(kThreads), 0, stream, x, res, mean, invstd,
                     weight, bias, total4, C / 4, relu, y, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// eval forward over the RUNNING statistics of one or two BatchNorms of the same shape (pb == nullptr: one)
int gpn::bn_fwd_eval_running(const gpn::BnFwdPtrs& pa, const gpn::BnFwdPtrs* pb, int64_t N, const gpn::DevRows& rows, int C, float eps,
                             int relu, hipStream_t stream) {
  GPN_CHECK_ARG(N >= 0 && C >= 4 && C % 4 == 0);
  if (N == 0) return GPN_OK;
  GPN_CHECK_ARG(pa.x && pa.weight && pa.bias && pa.running_mean && pa.running_var && pa.y);
  if (pb) GPN_CHECK_ARG(pb->x && pb->weight && pb->bias && pb->running_mean && pb->running_var && pb->y);
  const int64_t total4 = N * (C / 4);
  gpn::ProfScope prof(GPN_K_BN, stream, 0.0, 4.0 * (double)N * C * (pa.res ? 3 : 2) * (pb ? 2 : 1), rows.dev, N);  // x [+ res] read, y written
  hipLaunchKernelGGL(bn_apply_eval_kernel, dim3(apply_grid(gpn::plan_rows(N, rows) * (C / 4)), pb ? 2 : 1), dim3(kThreads), 0, stream,
                     pa, pb ? *pb : pa, total4, C / 4, eps, relu, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// backward of both modes.  y is the forward output (needed for the ReLU mask when relu != 0); dres may be NULL.
extern "C" int gpn_bn_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* mean,
                          const float* invstd, int64_t N, int C, int relu, int training, float* dx, float* dres,
                          float* dweight, float* dbias, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  return gpn::bn_bwd_rows(x, y, dy, weight, mean, invstd, N, gpn::DevRows(), C, relu, training, dx, dres, dweight, dbias, ws,
                          ws_bytes, (hipStream_t)stream_);
}

int gpn::bn_bwd_rows(const float* x, const float* y, const float* dy, const float* weight, const float* mean,
                     const float* invstd, int64_t N, const gpn::DevRows& rows, int C, int relu, int training, float* dx,
                     float* dres, float* dweight, float* dbias, void* ws, size_t ws_bytes, hipStream_t stream) {
  const int64_t Np = gpn::plan_rows(N, rows);
  GPN_CHECK_ARG(N >= 1 && C >= 4 && C % 4 == 0 && C <= 1024);
  GPN_CHECK_ARG(x && dy && weight && mean && invstd && dx && dweight && dbias && ws && (y || !relu));
  GPN_CHECK_ARG(ws_bytes >= (size_t)kMaxBlocks * 2 * C * sizeof(double));
  const int C4 = C / 4;
  gpn::ProfScope prof(GPN_K_BN, stream, 0.0, 4.0 * (double)N * C * (7 + (dres ? 1 : 0)), rows.dev, N);  // x, y, dy twice each; dx [, dres]
  if (Np <= kSmallRows && N < ((int64_t)1 << 31)) {
    // few workgroups, each a serial chain of row loads: 1024 threads per workgroup keep the chain one batch long
    if (C4 % 4 == 0 && Np > 256)
      hipLaunchKernelGGL((bn_small_bwd_kernel<4, 1024>), dim3(C4 / 4), dim3(1024), 0, stream, x, y, dy, weight, mean, invstd,
                         (int)N, C4, relu, training, dx, dres, dweight, dbias, rows.dev);
    else if (C4 % 4 == 0)
      hipLaunchKernelGGL((bn_small_bwd_kernel<4, 256>), dim3(C4 / 4), dim3(256), 0, stream, x, y, dy, weight, mean, invstd,
                         (int)N, C4, relu, training, dx, dres, dweight, dbias, rows.dev);
    else
      hipLaunchKernelGGL((bn_small_bwd_kernel<1, 256>), dim3(C4), dim3(256), 0, stream, x, y, dy, weight, mean, invstd,
                         (int)N, C4, relu, training, dx, dres, dweight, dbias, rows.dev);
    GPN_CHECK_LAUNCH();
    return GPN_OK;
  }
  const int blocks = reduce_blocks(Np, C4);
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(bn_reduce_kernel<true>, dim3(blocks), dim3(kReduceThreads), kReduceLds, stream, x, y, dy, mean, invstd,
                     N, C4, relu, partial, rows.dev);
  GPN_CHECK_LAUNCH();
  const int64_t total4 = N * C4, plan4 = Np * C4;
  if (C <= kFoldMaxC) {
    gpn::BnBwdPtrs pp;
    pp.x = x, pp.y = y, pp.dy = dy, pp.partial = partial, pp.mean = mean, pp.invstd = invstd, pp.weight = weight, pp.dx = dx,
    pp.dres = dres, pp.dweight = dweight, pp.dbias = dbias;
    hipLaunchKernelGGL(bn_apply_bwd_fold_kernel<false>, dim3(fold_grid(plan4)), dim3(kApplyThreads), 0, stream, pp, pp, blocks, total4,
                       C4, 1.0f / (float)N, relu, training, rows.dev);
    GPN_CHECK_LAUNCH();
    return GPN_OK;
  }
  if (rows.dev) {
    gpn::set_error("gpn_bn_bwd: a device-counted row count needs C <= %d", kFoldMaxC);
    return GPN_ERR_ARG;
  }
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, stream, partial, blocks, C,
                     dweight, dbias);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(apply_grid(total4)), dim3(kThreads), 0, stream, x, y, dy, mean, invstd,
                     weight, dweight, dbias, total4, C4, 1.0f / (float)N, relu, training, dx, dres);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// ---- apply passes over sums the producing conv launch accumulated (bn_stats.h; used by the network executor) ----------------
// shapes whose BatchNorm can run as an apply pass over sums a conv launch accumulated.  Any row count since round 3 (the
// small levels too: an apply launch is 4.8 us where the single-launch small-matrix form is 8.7 forward / 14 backward);
// env GPN_BN_FUSE_MIN_ROWS restores a lower bound (1025 = the round-2 behaviour)
bool gpn::bn_two_pass(int64_t N, int C) { return N >= 1 && C % 4 == 0 && C <= kFoldMaxC; }

int gpn::bn_fwd_train_fused(const gpn::BnFwdPtrs& p, const gpn::BnFwdPtrs* twin, int64_t N, int C, float eps, float momentum,
                            int relu, hipStream_t stream, const gpn::DevRows& rows) {
  for (const gpn::BnFwdPtrs* q : {&p, twin}) {
    if (!q) continue;
    GPN_CHECK_ARG(q->x && q->weight && q->bias && q->y && q->mean && q->invstd && q->partial);
    GPN_CHECK_ARG((q->running_mean == nullptr) == (q->running_var == nullptr));
  }
  GPN_CHECK_ARG(gpn::bn_two_pass(N, C));
  GPN_CHECK_ARG(!twin || (twin->res == nullptr) == (p.res == nullptr));
  const int64_t total4 = N * (C / 4);
  gpn::ProfScope prof(GPN_K_BN, stream, 0.0, 4.0 * (double)N * C * (p.res ? 3 : 2) * (twin ? 2 : 1), rows.dev, N);  // x [+ res] read, y written (N = the bound when rows.dev: scaled to the live rows by gpn_prof_get)
  const int64_t Np = gpn::plan_rows(N, rows);  // (slot sets in use: the same function of the plan the producing conv used)
  hipLaunchKernelGGL(bn_apply_fwd_fold_kernel<true>, dim3(fold_grid(Np * (C / 4)), twin ? 2 : 1), dim3(kApplyThreads), 0, stream, p,
                     twin ? *twin : p, gpn::stat_slot_count(Np), N, total4, C / 4, eps, momentum, relu, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

int gpn::bn_bwd_fused(const gpn::BnBwdPtrs& p, const gpn::BnBwdPtrs* twin, int64_t N, int C, int relu, int training,
                      hipStream_t stream, const gpn::DevRows& rows) {
  for (const gpn::BnBwdPtrs* q : {&p, twin}) {
    if (!q) continue;
    GPN_CHECK_ARG(q->x && q->dy && q->weight && q->mean && q->invstd && q->dx && q->dweight && q->dbias && q->partial && (q->y || !relu));
  }
  GPN_CHECK_ARG(gpn::bn_two_pass(N, C));
  GPN_CHECK_ARG(!twin || (twin->dres == nullptr) == (p.dres == nullptr));
  const int64_t total4 = N * (C / 4);
  gpn::ProfScope prof(GPN_K_BN, stream, 0.0, 4.0 * (double)N * C * (4 + (p.dres ? 1 : 0)) * (twin ? 2 : 1), rows.dev, N);  // x, y, dy read; dx [, dres] written
  const int64_t Np = gpn::plan_rows(N, rows);
  hipLaunchKernelGGL(bn_apply_bwd_fold_kernel<true>, dim3(fold_grid(Np * (C / 4)), twin ? 2 : 1), dim3(kApplyThreads), 0, stream, p,
                     twin ? *twin : p, gpn::stat_slot_count(Np), total4, C / 4, 1.0f / (float)N, relu, training, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
