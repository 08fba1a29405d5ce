// pose.hip — NPCS -> camera-frame similarity fit (5-point RANSAC + Umeyama + oriented box) for ALL proposals of a batch
// (SURVEY.md §8f rank 3).  Reference: gapartnet/misc/pose_fitting.py:4-147 (estimate_similarity_umeyama :4-43,
// evaluate_model :46-51, get_RANSAC_inliers :54-80, estimate_similarity_transform :83-118, estimate_pose_from_npcs :121-147),
// called per proposal from network/model.py:966-980: a Python loop of up to 100 iterations per proposal on CPU numpy.
//
// Two launches, float64 throughout (the reference is numpy float64):
//  * pose_hypotheses_kernel: one thread per (proposal, RANSAC iteration): the five picked correspondences, their Umeyama
//    fit with the 3x3 SVD done in registers (one-sided Jacobi: accurate for the small singular value of nearly coplanar
//    picks), transform [sR | t] to the workspace.
//  * pose_select_kernel: one workgroup per proposal: residual of every hypothesis over the proposal's points (thread =
//    hypothesis, points staged through LDS), the reference's sequential choice evaluated after the fact (running minimum,
//    first iteration below stop_thrsh, first arg-min up to it; NaN / >= 1e10 residuals never win), inliers of the chosen
//    hypothesis, Umeyama on the inliers, box corners.  Reductions over points are per-thread strided sums + a fixed LDS
//    tree: deterministic.
// Quirks of the reference that are kept (tests/test_pose_fitting_batched.py compares with the sequential function fed the
// same picks): a hypothesis is scored with transform @ source (column convention) although the fit is
// target = source @ (sR) + t; the pass threshold is max(|src|/|tgt|, |tgt|/|src|) of the MEAN point norms; the inlier ratio
// counts non-zero inlier INDICES; a single-point proposal is duplicated, which makes every hypothesis NaN: no pose.
#include "gpn_common.h"

namespace {

constexpr int kSelThreads = 128;
constexpr int kChunk = 128;  // points staged per pass of the residual loop

struct M3 {
  double a[3][3];
};

__device__ __forceinline__ double det3(const M3& m) {
  return m.a[0][0] * (m.a[1][1] * m.a[2][2] - m.a[1][2] * m.a[2][1]) - m.a[0][1] * (m.a[1][0] * m.a[2][2] - m.a[1][2] * m.a[2][0]) +
         m.a[0][2] * (m.a[1][0] * m.a[2][1] - m.a[1][1] * m.a[2][0]);
}

// A = U diag(S) V^T with S descending (one-sided Jacobi on the columns of A; zero singular values get unit vectors that
// complete the basis).  NaN input: NaN output.
__device__ void svd3(const M3& A, M3& U, double (&S)[3], M3& V) {
  double g[3][3], v[3][3];  // columns: g[col][row]
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) g[c][r] = A.a[r][c], v[c][r] = r == c ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double al = g[p][0] * g[p][0] + g[p][1] * g[p][1] + g[p][2] * g[p][2];
      const double be = g[q][0] * g[q][0] + g[q][1] * g[q][1] + g[q][2] * g[q][2];
      const double ga = g[p][0] * g[q][0] + g[p][1] * g[q][1] + g[p][2] * g[q][2];
      if (fabs(ga) > 1e-17 * sqrt(al * be) && fabs(ga) > 0.0) {
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double gp = g[p][r], gq = g[q][r];
          g[p][r] = c * gp - s * gq;
          g[q][r] = s * gp + c * gq;
          const double vp = v[p][r], vq = v[q][r];
          v[p][r] = c * vp - s * vq;
          v[q][r] = s * vp + c * vq;
        }
      }
    }
    if (!rotated) break;
  }
  double sig[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) sig[c] = sqrt(g[c][0] * g[c][0] + g[c][1] * g[c][1] + g[c][2] * g[c][2]);
  // order: descending singular values (stable for ties)
  int ord[3] = {0, 1, 2};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2 - i; ++j)
      if (sig[ord[j]] < sig[ord[j + 1]]) {
        const int t = ord[j];
        ord[j] = ord[j + 1];
        ord[j + 1] = t;
      }
  double u[3][3];
  const double tiny = 1e-300;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = ord[k];
    S[k] = sig[c];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      V.a[r][k] = v[c][r];
      u[k][r] = sig[c] > tiny ? g[c][r] / sig[c] : 0.0;
    }
  }
  // complete U where singular values vanished (rank 2: cross product; rank <= 1: any orthonormal completion)
  if (!(S[2] > tiny)) {
    if (!(S[1] > tiny)) {
      if (!(S[0] > tiny)) u[0][0] = 1.0, u[0][1] = 0.0, u[0][2] = 0.0;
      // a unit vector orthogonal to u0
      const int m = fabs(u[0][0]) <= fabs(u[0][1]) ? (fabs(u[0][0]) <= fabs(u[0][2]) ? 0 : 2) : (fabs(u[0][1]) <= fabs(u[0][2]) ? 1 : 2);
      double e[3] = {0.0, 0.0, 0.0};
      e[m] = 1.0;
      const double d = u[0][m];
      double w[3] = {e[0] - d * u[0][0], e[1] - d * u[0][1], e[2] - d * u[0][2]};
      const double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      u[1][0] = w[0] / n, u[1][1] = w[1] / n, u[1][2] = w[2] / n;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int r = 0; r < 3; ++r) U.a[r][k] = u[k][r];
}

// Umeyama from the moments of a point set: mu_s, mu_d, cov = sum (d - mu_d)(s - mu_s)^T / n, var = sum |s - mu_s|^2 / n
// -> T = [sR | t] (12 doubles, row-major 3x4), scale, R (pose_fitting.py:4-43)
__device__ void umeyama_from_moments(const double (&mu_s)[3], const double (&mu_d)[3], const M3& cov, double var, double (&T)[12],
                                     double& scale, M3& R) {
  M3 U, V;
  double S[3];
  svd3(cov, U, S, V);
  M3 Vh;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Vh.a[i][j] = V.a[j][i];
  if (det3(U) * det3(Vh) < 0.0) {  // reflection: flip the weakest axis
    S[2] = -S[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) U.a[r][2] = -U.a[r][2];
  }
  scale = (S[0] + S[1] + S[2]) / var;
  // rotation = (U Vh)^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += U.a[j][k] * Vh.a[k][i];
      R.a[i][j] = acc;
    }
  bool finite = true;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) finite = finite && isfinite(cov.a[i][j]);
  if (!finite) scale = nan("");
  // translation = mu_d - mu_s . (s R)   (row vector times matrix)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) acc += mu_s[i] * (scale * R.a[i][j]);
    T[j * 4 + 3] = mu_d[j] - acc;
#pragma unroll
    for (int i = 0; i < 3; ++i) T[j * 4 + i] = scale * R.a[j][i];  // diag(s) @ R
  }
}

__global__ __launch_bounds__(128) void pose_hypotheses_kernel(const double* __restrict__ xyz, const double* __restrict__ npcs,
                                                              const int64_t* __restrict__ offsets,
                                                              const int64_t* __restrict__ picks, int64_t P, int H,
                                                              double* __restrict__ hyp /* [P][H][12] */) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * H) return;
  const int64_t p = t / H;
  const int64_t o0 = offsets[p], n = offsets[p + 1] - o0;
  double s[5][3], d[5][3];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int64_t pick = picks[t * 5 + k];
    if (n == 1) pick = 0;  // a duplicated single point: picks 0 and 1 both address it
    const int64_t row = o0 + pick;
#pragma unroll
    for (int c = 0; c < 3; ++c) s[k][c] = npcs[row * 3 + c], d[k][c] = xyz[row * 3 + c];
  }
  double mu_s[3], mu_d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    mu_s[c] = (s[0][c] + s[1][c] + s[2][c] + s[3][c] + s[4][c]) / 5.0;
    mu_d[c] = (d[0][c] + d[1][c] + d[2][c] + d[3][c] + d[4][c]) / 5.0;
  }
  M3 cov;
  double var = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) acc += (d[k][i] - mu_d[i]) * (s[k][j] - mu_s[j]);
      cov.a[i][j] = acc / 5.0;
    }
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int c = 0; c < 3; ++c) var += (s[k][c] - mu_s[c]) * (s[k][c] - mu_s[c]);
  var /= 5.0;
  double T[12], scale;
  M3 R;
  umeyama_from_moments(mu_s, mu_d, cov, var, T, scale, R);
#pragma unroll
  for (int q = 0; q < 12; ++q) hyp[t * 12 + q] = isfinite(scale) ? T[q] : nan("");
}

// fixed-order sum of one double per thread over the workgroup (result in every thread)
__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = kSelThreads / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_max(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = kSelThreads / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kSelThreads) void pose_select_kernel(
    const double* __restrict__ xyz, const double* __restrict__ npcs, const int64_t* __restrict__ offsets,
    const double* __restrict__ hyp, int64_t P, int H, double stop_thrsh, uint8_t* __restrict__ valid,
    double* __restrict__ scale_out, double* __restrict__ rot_out, double* __restrict__ trans_out,
    double* __restrict__ transform_out, double* __restrict__ bbox_out, uint8_t* __restrict__ inlier_mask,
    int64_t* __restrict__ best_out, double* __restrict__ residual_out) {
  __shared__ double red[kSelThreads];
  __shared__ double pts[kChunk][6];
  __shared__ double res_sh[kSelThreads];
  __shared__ double bestT[12];
  __shared__ int best_sh, never_sh;
  const int64_t p = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t o0 = offsets[p], n = offsets[p + 1] - o0;
  const bool single = n == 1;
  // ---- pass threshold: max(|src| / |tgt|, |tgt| / |src|) of the mean point norms --------------------------------------
  double sn = 0.0, tn = 0.0;
  for (int64_t i = tid; i < n; i += kSelThreads) {
    const double* s = npcs + (o0 + i) * 3;
    const double* d = xyz + (o0 + i) * 3;
    sn += sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    tn += sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  }
  sn = block_sum(sn, red) / (double)n;
  tn = block_sum(tn, red) / (double)n;
  const double pass_thrsh = fmax(sn / tn, tn / sn);
  // ---- residual of hypothesis h = tid over all points (points staged through LDS) --------------------------------------
  for (int h0 = 0; h0 < H; h0 += kSelThreads) {
    const int h = h0 + tid;
    double T[12];
    if (h < H) {
#pragma unroll
      for (int q = 0; q < 12; ++q) T[q] = hyp[(p * H + h) * 12 + q];
    }
    double sq = 0.0;
    for (int64_t c0 = 0; c0 < n; c0 += kChunk) {
      const int cnt = (int)(n - c0 < kChunk ? n - c0 : kChunk);
      for (int e = tid; e < cnt * 6; e += kSelThreads) {
        const int i = e / 6, c = e - i * 6;
        pts[i][c] = c < 3 ? npcs[(o0 + c0 + i) * 3 + c] : xyz[(o0 + c0 + i) * 3 + (c - 3)];
      }
      __syncthreads();
      if (h < H) {
        for (int i = 0; i < cnt; ++i) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const double e = pts[i][3 + r] - (T[r * 4 + 0] * pts[i][0] + T[r * 4 + 1] * pts[i][1] + T[r * 4 + 2] * pts[i][2] + T[r * 4 + 3]);
            sq += e * e;
          }
        }
      }
      __syncthreads();
    }
    if (h < H) {
      const double r = sqrt(sq);
      residual_out[p * H + h] = r;
      if (H <= kSelThreads) res_sh[h] = r;
    }
  }
  __syncthreads();
  // ---- the reference's sequential choice, after the fact -----------------------------------------------------------------
  if (tid == 0) {
    double run = INFINITY;
    int stop = H - 1;
    for (int h = 0; h < H; ++h) {
      double r = H <= kSelThreads ? res_sh[h] : residual_out[p * H + h];
      if (isnan(r) || r >= 1e10) r = INFINITY;
      run = fmin(run, r);
      if (run < stop_thrsh) {
        stop = h;
        break;
      }
    }
    double bv = INFINITY;
    int best = 0;
    for (int h = 0; h <= stop; ++h) {
      double r = H <= kSelThreads ? res_sh[h] : residual_out[p * H + h];
      if (isnan(r) || r >= 1e10) r = INFINITY;
      if (r < bv) bv = r, best = h;
    }
    best_sh = best;
    never_sh = isfinite(bv) ? 0 : 1;
    best_out[p] = best;
  }
  __syncthreads();
  const int best = best_sh;
  const bool never = never_sh != 0;
  if (tid < 12) bestT[tid] = hyp[(p * H + best) * 12 + tid];
  __syncthreads();
  // ---- inliers of the chosen hypothesis; counts -----------------------------------------------------------------------------
  double counted = 0.0, n_in = 0.0;
  double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
  for (int64_t i = tid; i < n; i += kSelThreads) {
    const double* s = npcs + (o0 + i) * 3;
    const double* d = xyz + (o0 + i) * 3;
    double e2 = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double e = d[r] - (bestT[r * 4 + 0] * s[0] + bestT[r * 4 + 1] * s[1] + bestT[r * 4 + 2] * s[2] + bestT[r * 4 + 3]);
      e2 += e * e;
    }
    const bool in = (sqrt(e2) < pass_thrsh) && !never;
    inlier_mask[o0 + i] = in ? 1 : 0;  // (masked by `valid` at the end)
    if (in) {
      n_in += 1.0;
      if (i != 0) counted += 1.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) ms[c] += s[c], md[c] += d[c];
    }
  }
  counted = block_sum(counted, red);
  n_in = block_sum(n_in, red);
  const double ratio = counted / (double)n;
  bool ok = ratio >= 0.01 && n_in > 0.0 && !never && !single;
  // ---- Umeyama on the inliers ------------------------------------------------------------------------------------------------
  double mu_s[3], mu_d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    mu_s[c] = block_sum(ms[c], red) / n_in;
    mu_d[c] = block_sum(md[c], red) / n_in;
  }
  double cv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, var = 0.0;
  for (int64_t i = tid; i < n; i += kSelThreads) {
    if (!inlier_mask[o0 + i]) continue;
    const double* s = npcs + (o0 + i) * 3;
    const double* d = xyz + (o0 + i) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) cv[a * 3 + b] += (d[a] - mu_d[a]) * (s[b] - mu_s[b]);
      var += (s[a] - mu_s[a]) * (s[a] - mu_s[a]);
    }
  }
  M3 cov;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) cov.a[a][b] = block_sum(cv[a * 3 + b], red) / n_in;
  var = block_sum(var, red) / n_in;
  double T[12], scale;
  M3 R;
  umeyama_from_moments(mu_s, mu_d, cov, var, T, scale, R);  // (every thread: the box needs R, scale and t below)
  ok = ok && isfinite(scale);
  // ---- half extents of the NPCS-aligned box over the inliers: (xyz - t) @ pinv(R) / s, pinv(R) = R^T for a rotation ----------
  double hm[3] = {0, 0, 0};
  for (int64_t i = tid; i < n; i += kSelThreads) {
    if (!inlier_mask[o0 + i]) continue;
    const double* d = xyz + (o0 + i) * 3;
    const double q[3] = {d[0] - T[3], d[1] - T[7], d[2] - T[11]};
#pragma unroll
    for (int j = 0; j < 3; ++j) hm[j] = fmax(hm[j], fabs((q[0] * R.a[j][0] + q[1] * R.a[j][1] + q[2] * R.a[j][2]) / scale));
  }
  double half[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) half[j] = block_max(hm[j], red);
  if (!ok) {  // no pose: the inlier mask of this proposal is cleared, outputs are NaN
    for (int64_t i = tid; i < n; i += kSelThreads) inlier_mask[o0 + i] = 0;
  }
  if (tid == 0) {
    const double qn = nan("");
    valid[p] = ok ? 1 : 0;
    scale_out[p] = ok ? scale : qn;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      trans_out[p * 3 + i] = ok ? T[i * 4 + 3] : qn;
#pragma unroll
      for (int j = 0; j < 3; ++j) rot_out[p * 9 + i * 3 + j] = ok ? R.a[i][j] : qn;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) transform_out[p * 16 + i * 4 + j] = !ok ? qn : (i < 3 ? T[i * 4 + j] : (j == 3 ? 1.0 : 0.0));
    const int sg[8][3] = {{-1, -1, -1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc += (double)sg[k][i] * half[i] * scale * R.a[i][j];
        bbox_out[p * 24 + k * 3 + j] = ok ? acc + T[j * 4 + 3] : qn;
      }
  }
}

}  // namespace

extern "C" size_t gpn_pose_fit_ws_bytes(int64_t P, int H) { return gpn::align_up((size_t)(P > 0 ? P : 1) * H * 12 * sizeof(double)); }

// All proposals' poses in two launches.  xyz / npcs [M,3] f64 (points of all proposals, proposal p = rows
// offsets[p]:offsets[p+1], every proposal non-empty), picks [P,H,5] i64 (sample indices inside the proposal).
// Outputs (caller-allocated): valid [P] u8, scale [P], rotation [P,3,3], translation [P,3], transform [P,4,4],
// bbox [P,8,3] (NaN where not valid), inlier_mask [M] u8, best_iteration [P] i64, residual [P,H].
extern "C" int gpn_pose_fit(const double* xyz, const double* npcs, const int64_t* offsets, const int64_t* picks, int64_t P,
                            int64_t M, int H, double stop_thrsh, uint8_t* valid, double* scale, double* rotation,
                            double* translation, double* transform, double* bbox, uint8_t* inlier_mask,
                            int64_t* best_iteration, double* residual, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(P >= 0 && M >= 0 && H >= 1 && H <= 4096);
  if (P == 0) return GPN_OK;
  GPN_CHECK_ARG(xyz && npcs && offsets && picks && valid && scale && rotation && translation && transform && bbox && inlier_mask &&
                best_iteration && residual);
  if (!ws || ws_bytes < (size_t)P * H * 12 * sizeof(double)) {
    gpn::set_error("gpn_pose_fit: workspace too small");
    return GPN_ERR_WS;
  }
  double* hyp = static_cast<double*>(ws);
  hipLaunchKernelGGL(pose_hypotheses_kernel, dim3((unsigned)gpn::cdiv(P * H, 128)), dim3(128), 0, stream, xyz, npcs, offsets,
                     picks, P, H, hyp);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pose_select_kernel, dim3((unsigned)P), dim3(kSelThreads), 0, stream, xyz, npcs, offsets, hyp, P, H, stop_thrsh,
                     valid, scale, rotation, translation, transform, bbox, inlier_mask, best_iteration, residual);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
