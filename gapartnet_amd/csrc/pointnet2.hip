// pointnet2.hip — kernel family F: the PointNet++ point ops behind the reference's
// pybind module `pointnet2_cuda` (dataset/process_tools/utils/pointnet_lib/src/pointnet2_api.cpp:10-25).
// Same argument meaning and results as the vendored kernels; written for wave64 / LDS broadcast.
#include <cmath>

#include "gpn_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float dist2_nofma(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- ball_query_gpu.cu:9-45 : thread per query, candidates broadcast from LDS tiles ----------------
__global__ __launch_bounds__(kThreads) void pn2_ball_query_kernel(int n, int m, float radius2, int nsample,
                                                                  const float* __restrict__ new_xyz,
                                                                  const float* __restrict__ xyz,
                                                                  int32_t* __restrict__ idx) {
  __shared__ float tile[kThreads * 3];
  const int bs = blockIdx.y;
  const int pt = blockIdx.x * kThreads + threadIdx.x;
  const bool active = pt < m;
  const float* base = xyz + (int64_t)bs * n * 3;
  float qx = 0, qy = 0, qz = 0;
  if (active) {
    const float* q = new_xyz + ((int64_t)bs * m + pt) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  int32_t* out = idx + ((int64_t)bs * m + pt) * nsample;
  int cnt = 0;
  for (int k0 = 0; k0 < n; k0 += kThreads) {
    const int n_here = (n - k0 < kThreads) ? (n - k0) : kThreads;
    for (int e = threadIdx.x; e < n_here * 3; e += kThreads) tile[e] = base[(int64_t)k0 * 3 + e];
    __syncthreads();
    if (active && cnt < nsample) {
      for (int t = 0; t < n_here; ++t) {
        const float d2 = dist2_nofma(qx, qy, qz, tile[t * 3], tile[t * 3 + 1], tile[t * 3 + 2]);
        if (d2 < radius2) {
          const int k = k0 + t;
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) out[l] = k;
          out[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
    if (__syncthreads_and((!active) || cnt >= nsample)) break;
  }
}

// ---- group_points_gpu.cu:47-66 / :8-25 ---------------------------------------------------------------
__global__ void pn2_group_points_kernel(int c, int n, int npoints, int nsample, const float* __restrict__ points,
                                        const int32_t* __restrict__ idx, float* __restrict__ out) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= npoints * nsample) return;
  const int32_t j = idx[(int64_t)bs * npoints * nsample + index];
  out[((int64_t)bs * c + ch) * npoints * nsample + index] = points[((int64_t)bs * c + ch) * n + j];
}
__global__ void pn2_group_points_grad_kernel(int c, int n, int npoints, int nsample,
                                             const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
                                             float* __restrict__ grad_points) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= npoints * nsample) return;
  const int32_t j = idx[(int64_t)bs * npoints * nsample + index];
  atomicAdd(grad_points + ((int64_t)bs * c + ch) * n + j,
            grad_out[((int64_t)bs * c + ch) * npoints * nsample + index]);
}

// ---- sampling_gpu.cu:8-24 / :46-63 -------------------------------------------------------------------
__global__ void pn2_gather_points_kernel(int c, int n, int m, const float* __restrict__ points,
                                         const int32_t* __restrict__ idx, float* __restrict__ out) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= m) return;
  out[((int64_t)bs * c + ch) * m + pt] = points[((int64_t)bs * c + ch) * n + idx[(int64_t)bs * m + pt]];
}
__global__ void pn2_gather_points_grad_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                                              const int32_t* __restrict__ idx, float* __restrict__ grad_points) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= m) return;
  atomicAdd(grad_points + ((int64_t)bs * c + ch) * n + idx[(int64_t)bs * m + pt],
            grad_out[((int64_t)bs * c + ch) * m + pt]);
}

// ---- sampling_gpu.cu:93-209 : furthest point sampling ----------------------------------------------
// One 1024-thread workgroup per cloud.  The arg-max reproduces the reference's result for ITS block size
// Bref = opt_n_threads(n): highest distance, ties to the lowest reference thread id (k mod Bref), then
// to the lowest k.  Wave-level reduction by shuffles, cross-wave through LDS.
struct FpsCand {
  float v;
  int k;
};
__device__ __forceinline__ bool fps_better(float av, int ak, float bv, int bk, int bmask) {
  if (av != bv) return av > bv;
  // the reference's tree (sampling_gpu.cu:143-200) merges slot j with j+s for s = B/2 ... 1 and keeps slot j on
  // ties, so two tied candidates are decided at the lowest bit where their thread ids differ, the 0-bit winning:
  // ascending order of the bit-reversed thread id
  const unsigned ta = __brev((unsigned)(ak & bmask)), tb = __brev((unsigned)(bk & bmask));
  if (ta != tb) return ta < tb;
  return ak < bk;
}

__global__ __launch_bounds__(1024) void pn2_fps_kernel(int n, int m, int bmask, const float* __restrict__ dataset,
                                                       float* __restrict__ temp, int32_t* __restrict__ idxs) {
  __shared__ float wv[16];
  __shared__ int wk[16];
  __shared__ int s_old;
  const int bs = blockIdx.x;
  const float* d = dataset + (int64_t)bs * n * 3;
  float* t = temp + (int64_t)bs * n;
  int32_t* out = idxs + (int64_t)bs * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
    float best = -1.f;
    int besti = 0;
    for (int k = tid; k < n; k += 1024) {
      const float dd = dist2_nofma(d[k * 3], d[k * 3 + 1], d[k * 3 + 2], x1, y1, z1);
      const float tk = t[k];
      const float d2 = dd < tk ? dd : tk;
      t[k] = d2;
      if (d2 > best) { best = d2; besti = k; }
    }
    // threads with no point (tid >= n) carry (-1, tid): never better than a real candidate
    if (tid >= n) besti = tid;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_down(best, off, 64);
      const int ok = __shfl_down(besti, off, 64);
      if (lane + off < 64 && fps_better(ov, ok, best, besti, bmask)) { best = ov; besti = ok; }
    }
    if (lane == 0) { wv[wave] = best; wk[wave] = besti; }
    __syncthreads();
    if (wave == 0) {
      float v = lane < 16 ? wv[lane] : -2.f;
      int k = lane < 16 ? wk[lane] : 0x7fffffff;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const float ov = __shfl_down(v, off, 64);
        const int ok = __shfl_down(k, off, 64);
        if (fps_better(ov, ok, v, k, bmask)) { v = ov; k = ok; }
      }
      if (lane == 0) { s_old = k; out[j] = k; }
    }
    __syncthreads();
    old = s_old;
  }
}

// ---- interpolate_gpu.cu:81-124 : three_nn ------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void pn2_three_nn_kernel(int n, int m, const float* __restrict__ unknown,
                                                                const float* __restrict__ known,
                                                                float* __restrict__ dist2,
                                                                int32_t* __restrict__ idx) {
  __shared__ float tile[kThreads * 3];
  const int bs = blockIdx.y;
  const int pt = blockIdx.x * kThreads + threadIdx.x;
  const bool active = pt < n;
  const float* kn = known + (int64_t)bs * m * 3;
  float ux = 0, uy = 0, uz = 0;
  if (active) {
    const float* u = unknown + ((int64_t)bs * n + pt) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += kThreads) {
    const int n_here = (m - k0 < kThreads) ? (m - k0) : kThreads;
    for (int e = threadIdx.x; e < n_here * 3; e += kThreads) tile[e] = kn[(int64_t)k0 * 3 + e];
    __syncthreads();
    if (active) {
      for (int t = 0; t < n_here; ++t) {
        const float d = dist2_nofma(ux, uy, uz, tile[t * 3], tile[t * 3 + 1], tile[t * 3 + 2]);
        const int k = k0 + t;
        if (d < best1) { best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1; best1 = d; besti1 = k; }
        else if (d < best2) { best3 = best2; besti3 = besti2; best2 = d; besti2 = k; }
        else if (d < best3) { best3 = d; besti3 = k; }
      }
    }
    __syncthreads();
  }
  if (active) {
    float* d2 = dist2 + ((int64_t)bs * n + pt) * 3;
    int32_t* id = idx + ((int64_t)bs * n + pt) * 3;
    d2[0] = (float)best1; d2[1] = (float)best2; d2[2] = (float)best3;
    id[0] = besti1; id[1] = besti2; id[2] = besti3;
  }
}

// ---- interpolate_gpu.cu:9-57 : knn (k <= 200), insertion into a sorted per-thread list ---------------
__global__ __launch_bounds__(64) void pn2_knn_kernel(int n, int m, int k, const float* __restrict__ unknown,
                                                     const float* __restrict__ known, float* __restrict__ dist2,
                                                     int32_t* __restrict__ idx) {
  const int bs = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= n) return;
  const float* u = unknown + ((int64_t)bs * n + pt) * 3;
  const float* kn = known + (int64_t)bs * m * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  double best[200];
  int besti[200];
  for (int i = 0; i < k; ++i) { best[i] = 1e40; besti[i] = 0; }
  for (int i = 0; i < m; ++i) {
    const float d = dist2_nofma(ux, uy, uz, kn[i * 3], kn[i * 3 + 1], kn[i * 3 + 2]);
    for (int j = 0; j < k; ++j) {
      if (d < best[j]) {
        for (int l = k - 1; l > j; --l) { best[l] = best[l - 1]; besti[l] = besti[l - 1]; }
        best[j] = d; besti[j] = i;
        break;
      }
    }
  }
  for (int i = 0; i < k; ++i) {
    idx[((int64_t)bs * n + pt) * k + i] = besti[i];
    dist2[((int64_t)bs * n + pt) * k + i] = (float)best[i];
  }
}

// ---- interpolate_gpu.cu:149-169 / :192-214 -----------------------------------------------------------
__global__ void pn2_three_interpolate_kernel(int c, int m, int n, const float* __restrict__ points,
                                             const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                             float* __restrict__ out) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= n) return;
  const float* w = weight + ((int64_t)bs * n + pt) * 3;
  const int32_t* id = idx + ((int64_t)bs * n + pt) * 3;
  const float* p = points + ((int64_t)bs * c + ch) * m;
  out[((int64_t)bs * c + ch) * n + pt] =
      __fadd_rn(__fadd_rn(__fmul_rn(w[0], p[id[0]]), __fmul_rn(w[1], p[id[1]])), __fmul_rn(w[2], p[id[2]]));
}
__global__ void pn2_three_interpolate_grad_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                                                  const int32_t* __restrict__ idx,
                                                  const float* __restrict__ weight,
                                                  float* __restrict__ grad_points) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= n) return;
  const float g = grad_out[((int64_t)bs * c + ch) * n + pt];
  const float* w = weight + ((int64_t)bs * n + pt) * 3;
  const int32_t* id = idx + ((int64_t)bs * n + pt) * 3;
  float* gp = grad_points + ((int64_t)bs * c + ch) * m;
  atomicAdd(gp + id[0], __fmul_rn(g, w[0]));
  atomicAdd(gp + id[1], __fmul_rn(g, w[1]));
  atomicAdd(gp + id[2], __fmul_rn(g, w[2]));
}

int opt_n_threads(int work_size) {  // cuda_utils.h:10-14
  const int pow_2 = (int)(std::log((double)work_size) / std::log(2.0));
  int v = 1 << pow_2;
  if (v > 1024) v = 1024;
  if (v < 1) v = 1;
  return v;
}

}  // namespace

extern "C" int gpn_pn2_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                                  const float* xyz, int32_t* idx, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample >= 1);
  if (b == 0 || m == 0) return GPN_OK;
  GPN_CHECK_ARG(new_xyz && xyz && idx);
  hipLaunchKernelGGL(pn2_ball_query_kernel, dim3((unsigned)gpn::cdiv(m, kThreads), b), dim3(kThreads), 0, stream, n,
                     m, radius * radius, nsample, new_xyz, xyz, idx);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_pn2_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                                    const int32_t* idx, float* out, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
  if (b == 0 || c == 0 || npoints * nsample == 0) return GPN_OK;
  GPN_CHECK_ARG(points && idx && out);
  hipLaunchKernelGGL(pn2_group_points_kernel, dim3((unsigned)gpn::cdiv((int64_t)npoints * nsample, kThreads), c, b),
                     dim3(kThreads), 0, stream, c, n, npoints, nsample, points, idx, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
extern "C" int gpn_pn2_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                                         const int32_t* idx, float* grad_points, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
  if (b == 0 || c == 0 || npoints * nsample == 0) return GPN_OK;
  GPN_CHECK_ARG(grad_out && idx && grad_points);
  hipLaunchKernelGGL(pn2_group_points_grad_kernel,
                     dim3((unsigned)gpn::cdiv((int64_t)npoints * nsample, kThreads), c, b), dim3(kThreads), 0, stream,
                     c, n, npoints, nsample, grad_out, idx, grad_points);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_pn2_gather_points(int b, int c, int n, int npoints, const float* points, const int32_t* idx,
                                     float* out, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0);
  if (b == 0 || c == 0 || npoints == 0) return GPN_OK;
  GPN_CHECK_ARG(points && idx && out);
  hipLaunchKernelGGL(pn2_gather_points_kernel, dim3((unsigned)gpn::cdiv(npoints, kThreads), c, b), dim3(kThreads),
                     0, stream, c, n, npoints, points, idx, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
extern "C" int gpn_pn2_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out,
                                          const int32_t* idx, float* grad_points, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0);
  if (b == 0 || c == 0 || npoints == 0) return GPN_OK;
  GPN_CHECK_ARG(grad_out && idx && grad_points);
  hipLaunchKernelGGL(pn2_gather_points_grad_kernel, dim3((unsigned)gpn::cdiv(npoints, kThreads), c, b),
                     dim3(kThreads), 0, stream, c, n, npoints, grad_out, idx, grad_points);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// ---- the same sampling for big clouds (pre-processing: N ~ 1e5..1e6 -> 20 000 samples) -------------------------------
// One workgroup streams all N points through one CU every iteration (16 MB per iteration at N = 1e6: ~200 us, x 20 000).
// Here G workgroups share a cloud: each owns a contiguous chunk of the points (and of `temp`), reduces its chunk to one
// candidate, publishes it, and all G meet at a counter barrier; every workgroup then reduces the G candidates itself, so
// the winner never has to be broadcast.  The candidate order is the single-workgroup kernel's (fps_better: distance, then
// the reference's thread-id tie-break, then index), so the samples are identical.  All b * G workgroups must be resident
// at once (b * G <= number of CUs; the host picks G accordingly).
__global__ __launch_bounds__(1024) void pn2_fps_multi_kernel(int n, int m, int bmask, int G,
                                                             const float* __restrict__ dataset, float* __restrict__ temp,
                                                             int32_t* __restrict__ idxs, float* cand_v /* [b][2][G] */,
                                                             int* cand_k /* [b][2][G] */, unsigned* arrived /* [b] */) {
  __shared__ float wv[16];
  __shared__ int wk[16];
  __shared__ int s_old;
  const int bs = blockIdx.x / G, w = blockIdx.x - bs * G;
  const float* d = dataset + (int64_t)bs * n * 3;
  float* t = temp + (int64_t)bs * n;
  int32_t* out = idxs + (int64_t)bs * m;
  float* cv = cand_v + (int64_t)bs * 2 * G;
  int* ck = cand_k + (int64_t)bs * 2 * G;
  unsigned* counter = arrived + bs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (n + G - 1) / G;
  const int lo = w * chunk, hi = lo + chunk < n ? lo + chunk : n;
  int old = 0;
  if (w == 0 && tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
    float best = -1.f;
    int besti = 0x7fffffff;
    for (int k = lo + tid; k < hi; k += 1024) {
      const float dd = dist2_nofma(d[k * 3], d[k * 3 + 1], d[k * 3 + 2], x1, y1, z1);
      const float tk = t[k];
      const float d2 = dd < tk ? dd : tk;
      t[k] = d2;
      if (fps_better(d2, k, best, besti, bmask)) { best = d2; besti = k; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_down(best, off, 64);
      const int ok = __shfl_down(besti, off, 64);
      if (lane + off < 64 && fps_better(ov, ok, best, besti, bmask)) { best = ov; besti = ok; }
    }
    if (lane == 0) { wv[wave] = best; wk[wave] = besti; }
    __syncthreads();
    const int slot = (j & 1) * G;
    if (wave == 0) {
      float v = lane < 16 ? wv[lane] : -2.f;
      int k = lane < 16 ? wk[lane] : 0x7fffffff;
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const float ov = __shfl_down(v, off, 64);
        const int ok = __shfl_down(k, off, 64);
        if (fps_better(ov, ok, v, k, bmask)) { v = ov; k = ok; }
      }
      if (lane == 0) {
        __hip_atomic_store(cv + slot + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ck + slot + w, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)G * (unsigned)j;  // j-th meeting of G workgroups (m * G < 2^32: host-checked)
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        __threadfence();
      }
      // lane 0 has passed the barrier; the wave re-converges here and reads every workgroup's candidate (G <= 64)
      __threadfence();  // every lane: the candidate reads below stay behind lane 0's acquire
      float gv = -2.f;
      int gk = 0x7fffffff;
      if (lane < G) {
        gv = __hip_atomic_load(cv + slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gk = __hip_atomic_load(ck + slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_down(gv, off, 64);
        const int ok = __shfl_down(gk, off, 64);
        if (lane + off < 64 && fps_better(ov, ok, gv, gk, bmask)) { gv = ov; gk = ok; }
      }
      if (lane == 0) {
        s_old = gk;
        if (w == 0) out[j] = gk;
      }
    }
    __syncthreads();
    old = s_old;
  }
}

extern "C" int gpn_pn2_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp,
                                               int32_t* idxs, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && n >= 1 && m >= 0);
  if (b == 0 || m == 0) return GPN_OK;
  GPN_CHECK_ARG(dataset && temp && idxs);
  const int bref = opt_n_threads(n);
  hipLaunchKernelGGL(pn2_fps_kernel, dim3(b), dim3(1024), 0, stream, n, m, bref - 1, dataset, temp, idxs);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// workgroups per cloud of the multi-workgroup form: as many as are guaranteed to be resident together (occupancy query x
// CU count; the kernel meets at a counter barrier, so every workgroup of a cloud must be running), at most 64 (one wave
// reduces the candidates), none for clouds a single workgroup handles faster than a chip-wide barrier per sample costs
static int fps_groups(int b, int n) {
  if (n < 65536 || b < 1) return 1;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pn2_fps_multi_kernel, 1024, 0) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return 1;
  }
  int g = cus / b;  // one workgroup per CU: the point chunks stream through a CU's own L1/LDS path
  if (g > 64) g = 64;
  return g < 2 ? 1 : g;
}

extern "C" size_t gpn_pn2_furthest_point_sampling_ws_bytes(int b, int n) {
  const int G = fps_groups(b, n);
  if (G == 1) return 0;
  return gpn::align_up((size_t)b * 2 * G * sizeof(float)) + gpn::align_up((size_t)b * 2 * G * sizeof(int)) +
         gpn::align_up((size_t)b * sizeof(unsigned));
}

extern "C" int gpn_pn2_furthest_point_sampling_ws(int b, int n, int m, const float* dataset, float* temp, int32_t* idxs,
                                                  void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && n >= 1 && m >= 0);
  if (b == 0 || m == 0) return GPN_OK;
  const int G = fps_groups(b, n);
  if (G == 1) return gpn_pn2_furthest_point_sampling(b, n, m, dataset, temp, idxs, stream_);
  GPN_CHECK_ARG(dataset && temp && idxs);
  GPN_CHECK_ARG((int64_t)m * G < (int64_t)0x7fffffff);
  gpn::WsCarver carve(ws, ws_bytes);
  float* cand_v = carve.take<float>((size_t)b * 2 * G);
  int* cand_k = carve.take<int>((size_t)b * 2 * G);
  unsigned* arrived = carve.take<unsigned>((size_t)b);
  GPN_CHECK_WS(carve);
  GPN_CHECK_HIP(hipMemsetAsync(arrived, 0, (size_t)b * sizeof(unsigned), stream));
  const int bref = opt_n_threads(n);
  // cooperative launch: the runtime starts the grid only when ALL its workgroups can be resident at once (and refuses a
  // grid that cannot be), which is what the in-kernel counter barrier needs - a plain launch next to other streams' kernels,
  // other ranks sharing the device or a CU mask could leave some workgroups unscheduled while the resident ones spin.
  int bmask = bref - 1, groups = G;
  void* args[] = {&n, &m, &bmask, &groups, &dataset, &temp, &idxs, &cand_v, &cand_k, &arrived};
  const hipError_t err = hipLaunchCooperativeKernel((const void*)pn2_fps_multi_kernel, dim3(b * G), dim3(1024), args, 0, stream);
  if (err != hipSuccess) {  // not co-schedulable here (too large for this device / partition): single-workgroup form
    (void)hipGetLastError();
    return gpn_pn2_furthest_point_sampling(b, n, m, dataset, temp, idxs, stream_);
  }
  return GPN_OK;
}

extern "C" int gpn_pn2_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                                int32_t* idx, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && n >= 0 && m >= 0);
  if (b == 0 || n == 0) return GPN_OK;
  GPN_CHECK_ARG(unknown && known && dist2 && idx);
  hipLaunchKernelGGL(pn2_three_nn_kernel, dim3((unsigned)gpn::cdiv(n, kThreads), b), dim3(kThreads), 0, stream, n,
                     m, unknown, known, dist2, idx);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_pn2_knn(int b, int n, int m, int k, const float* unknown, const float* known, float* dist2,
                           int32_t* idx, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && k >= 1 && k <= 200);
  if (b == 0 || n == 0) return GPN_OK;
  GPN_CHECK_ARG(unknown && known && dist2 && idx);
  hipLaunchKernelGGL(pn2_knn_kernel, dim3((unsigned)gpn::cdiv(n, 64), b), dim3(64), 0, stream, n, m, k, unknown,
                     known, dist2, idx);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_pn2_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx,
                                         const float* weight, float* out, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && c >= 0 && m >= 0 && n >= 0);
  if (b == 0 || c == 0 || n == 0) return GPN_OK;
  GPN_CHECK_ARG(points && idx && weight && out);
  hipLaunchKernelGGL(pn2_three_interpolate_kernel, dim3((unsigned)gpn::cdiv(n, kThreads), c, b), dim3(kThreads), 0,
                     stream, c, m, n, points, idx, weight, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
extern "C" int gpn_pn2_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                              const int32_t* idx, const float* weight, float* grad_points,
                                              gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(b >= 0 && c >= 0 && m >= 0 && n >= 0);
  if (b == 0 || c == 0 || n == 0) return GPN_OK;
  GPN_CHECK_ARG(grad_out && idx && weight && grad_points);
  hipLaunchKernelGGL(pn2_three_interpolate_grad_kernel, dim3((unsigned)gpn::cdiv(n, kThreads), c, b),
                     dim3(kThreads), 0, stream, c, n, m, grad_out, idx, weight, grad_points);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
