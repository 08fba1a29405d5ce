// linear.hip — the dense heads of the path: y = x W^T + b on [N, cin] rows with cin, cout <= 64 (include/gpn.h section H).
//
// Reference: network/model.py:114-120, 160-175, 322-337 — sem_seg_head Linear(16, classes), offset_head Linear(16, 16) ->
// BatchNorm -> ReLU -> Linear(16, 3), score_head Linear(16, classes - 1), npcs_head Linear(16, 3 (classes - 1)): five tiny
// GEMMs over 10^5 rows, forward and backward, per training step.  They are HBM-streaming work (160k x (16 + 27) floats), not
// matrix work; until round 3 they ran as K = 1 cases of the sparse-conv kernels, which cost ~16 launches per layer and pass
// pair (channel padding to 16, an identity rulebook, weight packing, the conv, the bias add; padded dgrad, wgrad + slice
// reduce, bias sum).  Here: ONE forward launch, and three for backward (dx; per-workgroup partial dW / db; their
// fixed-order sum) - deterministic, fp32 accumulation in a fixed order (ci ascending; rows ascending, then workgroups).
#include "gpn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxC = 64;        // cin, cout <= 64
constexpr int kRowsPerWg = 128;  // rows a workgroup of the dW pass reduces (round 4: 512 -> 128 - a pass over 10k rows had 20 workgroups
                                 // for 256 CUs: 23.9 -> 14.6 us for 16 -> 27 channels, 104 -> 30 us for 4096 x 64 -> 64; unchanged at 160k rows)
constexpr int kTileRows = 128;

// thread = (row, j): outputs 4j .. 4j + 3 of that row.  W^T is staged in LDS ([ci][cout padded to 4]: the four outputs of a
// thread are one 16-byte LDS read per ci); the threads of a row read the same x row (one L1 line, broadcast)
__global__ __launch_bounds__(kThreads) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ b, int64_t N, int cin, int cout,
                                                              float* __restrict__ y, const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);  // (device-counted rows, gpn::DevRows)
  __shared__ __attribute__((aligned(16))) float Wt[kMaxC][kMaxC];  // [ci][o]
  __shared__ __attribute__((aligned(16))) float bs[kMaxC];
  const int Q = (cout + 3) >> 2, cp = Q * 4;
  for (int e = threadIdx.x; e < cin * cp; e += kThreads) {
    const int ci = e / cp, o = e - ci * cp;
    Wt[ci][o] = o < cout ? W[(int64_t)o * cin + ci] : 0.f;
  }
  for (int o = threadIdx.x; o < cp; o += kThreads) bs[o] = (b && o < cout) ? b[o] : 0.f;
  __syncthreads();
  const int64_t total = N * Q;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (int64_t)gridDim.x * kThreads) {
    const int64_t row = t / Q;
    const int j = (int)(t - row * Q);
    const float4* __restrict__ xr = reinterpret_cast<const float4*>(x + row * cin);
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = 0; c4 < (cin >> 2); ++c4) {
      const float4 xv = xr[c4];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 w = *reinterpret_cast<const float4*>(&Wt[4 * c4 + u][4 * j]);
        acc.x = fmaf(xs[u], w.x, acc.x);
        acc.y = fmaf(xs[u], w.y, acc.y);
        acc.z = fmaf(xs[u], w.z, acc.z);
        acc.w = fmaf(xs[u], w.w, acc.w);
      }
    }
    const float4 bv = *reinterpret_cast<const float4*>(&bs[4 * j]);
    acc.x += bv.x, acc.y += bv.y, acc.z += bv.z, acc.w += bv.w;
    float* __restrict__ yr = y + row * cout + 4 * j;
    if ((cout & 3) == 0) {
      *reinterpret_cast<float4*>(yr) = acc;
    } else {
      const float a[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * j + u < cout) yr[u] = a[u];
    }
  }
}

// dx[row, 4c .. 4c + 3] = sum_o dy[row, o] W[o, 4c .. 4c + 3]: thread = (row, c)
__global__ __launch_bounds__(kThreads) void linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W, int64_t N,
                                                             int cin, int cout, float* __restrict__ dx,
                                                             const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);
  __shared__ __attribute__((aligned(16))) float Ws[kMaxC][kMaxC];  // [o][ci]
  for (int e = threadIdx.x; e < cout * cin; e += kThreads) Ws[e / cin][e % cin] = W[e];
  __syncthreads();
  const int C4 = cin >> 2;
  const int64_t total = N * C4;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (int64_t)gridDim.x * kThreads) {
    const int64_t row = t / C4;
    const int c = (int)(t - row * C4);
    const float* __restrict__ g = dy + row * cout;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int o = 0; o < cout; ++o) {
      const float gv = g[o];
      const float4 w = *reinterpret_cast<const float4*>(&Ws[o][4 * c]);
      acc.x = fmaf(gv, w.x, acc.x);
      acc.y = fmaf(gv, w.y, acc.y);
      acc.z = fmaf(gv, w.z, acc.z);
      acc.w = fmaf(gv, w.w, acc.w);
    }
    reinterpret_cast<float4*>(dx)[t] = acc;
  }
}

// partial[wg][o][ci] = sum over the workgroup's rows of dy[row, o] x[row, ci];  partial[wg][cout * cin + o] = sum of dy[row, o].
// Rows go through LDS in tiles of 128 rows (64 for the widest layers: 64 KB of dynamic LDS at most) (coalesced loads; dynamic LDS sized to the layer: a 32-row tile made the kernel a
// chain of 16 load round trips per workgroup, 36 us per layer); thread (o, c) owns four adjacent ci of one o.
__global__ __launch_bounds__(kThreads) void linear_dw_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     int64_t N, int cin, int cout, int tile_rows,
                                                                     float* __restrict__ partial, const int64_t* __restrict__ n_dev) {
  N = gpn::live_rows(n_dev, N);  // (workgroups past the live rows write zero partials: the sum below runs over the bound's)
  extern __shared__ __attribute__((aligned(16))) float lin_smem[];
  float* xs = lin_smem;                       // [tile_rows][cin]
  const int gp = cout + 1;                    // (odd pitch: the o-th column of consecutive rows in different banks)
  float* gs = lin_smem + tile_rows * cin;     // [tile_rows][cout + 1]
  const int C4 = cin >> 2;
  const int owners = cout * C4;  // (o, c) pairs; a thread takes pairs t, t + 256, ... (<= 4 of them at 64 x 64)
  // the heads have 12-108 pairs: G = 256 / owners row groups share a tile's rows (group g takes rows g, g + G, ...) and are
  // summed in group order at the end - with one group 40 threads of 256 did all the arithmetic of a 16 -> 10 layer
  const int G = owners <= kThreads / 2 ? kThreads / owners : 1;
  const int g = G > 1 ? (int)threadIdx.x / owners : 0;
  const int pg = G > 1 ? (int)threadIdx.x - g * owners : (int)threadIdx.x;  // pair of this thread in the grouped form
  float4 acc[4];
  float bacc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = float4{0.f, 0.f, 0.f, 0.f}, bacc[k] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerWg;
  const int64_t r1 = r0 + kRowsPerWg < N ? r0 + kRowsPerWg : N;
  for (int64_t base = r0; base < r1; base += tile_rows) {
    const int rows = (int)(r1 - base < tile_rows ? r1 - base : tile_rows);
    __syncthreads();
    // the tile's rows are contiguous in both arrays: straight float4 / float copies
    const float4* __restrict__ xsrc = reinterpret_cast<const float4*>(x + base * cin);
    for (int e = threadIdx.x; e < rows * C4; e += kThreads) reinterpret_cast<float4*>(xs)[e] = xsrc[e];
    const float* __restrict__ gsrc = dy + base * cout;
    for (int e = threadIdx.x; e < rows * cout; e += kThreads) {
      const int r = e / cout, o = e - r * cout;
      gs[r * gp + o] = gsrc[e];
    }
    __syncthreads();
    if (G > 1) {
      if (g < G) {
        const int o = pg / C4, c = pg - o * C4;
        for (int r = g; r < rows; r += G) {
          const float gv = gs[r * gp + o];
          const float4 xv = *reinterpret_cast<const float4*>(&xs[r * cin + 4 * c]);
          acc[0].x = fmaf(gv, xv.x, acc[0].x);
          acc[0].y = fmaf(gv, xv.y, acc[0].y);
          acc[0].z = fmaf(gv, xv.z, acc[0].z);
          acc[0].w = fmaf(gv, xv.w, acc[0].w);
          if (c == 0) bacc[0] += gv;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = threadIdx.x + k * kThreads;
        if (p < owners) {
          const int o = p / C4, c = p - o * C4;
          for (int r = 0; r < rows; ++r) {
            const float gv = gs[r * gp + o];
            const float4 xv = *reinterpret_cast<const float4*>(&xs[r * cin + 4 * c]);
            acc[k].x = fmaf(gv, xv.x, acc[k].x);
            acc[k].y = fmaf(gv, xv.y, acc[k].y);
            acc[k].z = fmaf(gv, xv.z, acc[k].z);
            acc[k].w = fmaf(gv, xv.w, acc[k].w);
            if (c == 0) bacc[k] += gv;
          }
        }
      }
    }
  }
  float* __restrict__ out = partial + (int64_t)blockIdx.x * (cout * cin + cout);
  if (G > 1) {  // the groups' sums through LDS (the tiles are done with), added in group order by group 0
    __syncthreads();
    float* red = lin_smem;  // [G][owners][5]
    if (g < G) {
      float* q = red + ((size_t)g * owners + pg) * 5;
      q[0] = acc[0].x, q[1] = acc[0].y, q[2] = acc[0].z, q[3] = acc[0].w, q[4] = bacc[0];
    }
    __syncthreads();
    if (g == 0) {
      float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      for (int gg = 0; gg < G; ++gg)
#pragma unroll
        for (int u = 0; u < 5; ++u) v[u] += red[((size_t)gg * owners + pg) * 5 + u];
      const int o = pg / C4, c = pg - o * C4;
      *reinterpret_cast<float4*>(out + o * cin + 4 * c) = float4{v[0], v[1], v[2], v[3]};
      if (c == 0) out[cout * cin + o] = v[4];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = threadIdx.x + k * kThreads;
    if (p < owners) {
      const int o = p / C4, c = p - o * C4;
      *reinterpret_cast<float4*>(out + o * cin + 4 * c) = acc[k];
      if (c == 0) out[cout * cin + o] = bacc[k];
    }
  }
}

// dW / db = the partials summed over the workgroups: 16 lanes per element stride over them (8 loads in flight each), then a
// fixed-order shuffle tree - deterministic; one thread per element walking all 313 partials of a 160k-row layer was a
// chain of 40 round trips (10 us)
__global__ __launch_bounds__(kThreads) void linear_dw_sum_kernel(const float* __restrict__ partial, int blocks, int cin, int cout,
                                                                 float* __restrict__ dW, float* __restrict__ db) {
  const int per = cout * cin + cout;
  const int part = threadIdx.x & 15;
  const int e = blockIdx.x * (kThreads / 16) + (threadIdx.x >> 4);
  float acc = 0.f;
  if (e < per) {
    int b = part;
    for (; b + 7 * 16 < blocks; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 16 * u) * per + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < blocks; b += 16) acc += partial[(int64_t)b * per + e];
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 16);
  if (part != 0 || e >= per) return;
  if (e < cout * cin) {
    if (dW) dW[e] = acc;
  } else if (db) {
    db[e - cout * cin] = acc;
  }
}

bool shape_ok(int64_t N, int cin, int cout) { return N >= 0 && cin >= 4 && cin % 4 == 0 && cin <= kMaxC && cout >= 1 && cout <= kMaxC; }

inline int grid_for(int64_t total) {
  const int64_t g = gpn::cdiv(total, kThreads);
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

// 1 if gpn_linear_fwd / gpn_linear_bwd take this shape (cin a multiple of 4, both widths <= 64)
extern "C" int gpn_linear_supported(int cin, int cout) { return shape_ok(0, cin, cout) ? 1 : 0; }

// y [N, cout] = x [N, cin] W^T + b;  W [cout, cin] (torch.nn.Linear's layout), b [cout] or NULL
static int linear_fwd_impl(const float* x, const float* W, const float* b, int64_t N, const gpn::DevRows& rows, int cin, int cout,
                           float* y, hipStream_t stream) {
  GPN_CHECK_ARG(shape_ok(N, cin, cout));
  if (N == 0) return GPN_OK;
  GPN_CHECK_ARG(x && W && y);
  gpn::ProfScope prof(GPN_K_LINEAR, stream, 2.0 * (double)N * cin * cout, 4.0 * (double)N * (cin + cout), rows.dev, N);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(grid_for(gpn::plan_rows(N, rows) * ((cout + 3) / 4))), dim3(kThreads), 0, stream, x, W, b, N,
                     cin, cout, y, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
extern "C" int gpn_linear_fwd(const float* x, const float* W, const float* b, int64_t N, int cin, int cout, float* y,
                              gpn_stream_t stream_) {
  return linear_fwd_impl(x, W, b, N, gpn::DevRows(), cin, cout, y, (hipStream_t)stream_);
}
// row count on the device (N = the bound of x / y)
extern "C" int gpn_linear_fwd_dev(const float* x, const float* W, const float* b, int64_t N, const int64_t* n_dev, int64_t n_plan,
                                  int cin, int cout, float* y, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev != nullptr);
  return linear_fwd_impl(x, W, b, N, gpn::DevRows{n_dev, n_plan}, cin, cout, y, (hipStream_t)stream_);
}

extern "C" size_t gpn_linear_bwd_ws_bytes(int64_t N, int cin, int cout) {
  return gpn::align_up((size_t)gpn::cdiv(N > 0 ? N : 1, (int64_t)kRowsPerWg) * (size_t)(cout * cin + cout) * sizeof(float));
}

// dx [N, cin] = dy W, dW [cout, cin] = dy^T x, db [cout] = column sums of dy; any of the three outputs may be NULL (skipped)
static int linear_bwd_impl(const float* x, const float* W, const float* dy, int64_t N, const gpn::DevRows& rows, int cin, int cout,
                           float* dx, float* dW, float* db, void* ws, size_t ws_bytes, hipStream_t stream);
extern "C" int gpn_linear_bwd(const float* x, const float* W, const float* dy, int64_t N, int cin, int cout, float* dx, float* dW,
                              float* db, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  return linear_bwd_impl(x, W, dy, N, gpn::DevRows(), cin, cout, dx, dW, db, ws, ws_bytes, (hipStream_t)stream_);
}
extern "C" int gpn_linear_bwd_dev(const float* x, const float* W, const float* dy, int64_t N, const int64_t* n_dev, int64_t n_plan,
                                  int cin, int cout, float* dx, float* dW, float* db, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev != nullptr);
  return linear_bwd_impl(x, W, dy, N, gpn::DevRows{n_dev, n_plan}, cin, cout, dx, dW, db, ws, ws_bytes, (hipStream_t)stream_);
}
static int linear_bwd_impl(const float* x, const float* W, const float* dy, int64_t N, const gpn::DevRows& rows, int cin, int cout,
                           float* dx, float* dW, float* db, void* ws, size_t ws_bytes, hipStream_t stream) {
  GPN_CHECK_ARG(shape_ok(N, cin, cout));
  if (N == 0) {
    if (dW) GPN_CHECK_HIP(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)cout * cin, stream));
    if (db) GPN_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * (size_t)cout, stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(dy && (!dx || W) && (!dW || x));
  gpn::ProfScope prof(GPN_K_LINEAR, stream, (dx ? 2.0 : 0.0) * (double)N * cin * cout + (dW ? 2.0 : 0.0) * (double)N * cin * cout,
                      4.0 * (double)N * ((dx ? cin + cout : 0) + (dW || db ? cin + cout : 0)), rows.dev, N);
  if (dx) {
    hipLaunchKernelGGL(linear_dx_kernel, dim3(grid_for(gpn::plan_rows(N, rows) * (cin / 4))), dim3(kThreads), 0, stream, dy, W, N, cin, cout,
                       dx, rows.dev);
    GPN_CHECK_LAUNCH();
  }
  if (dW || db) {
    const int blocks = (int)gpn::cdiv(N, (int64_t)kRowsPerWg);
    if (!ws || ws_bytes < (size_t)blocks * (size_t)(cout * cin + cout) * sizeof(float)) {
      gpn::set_error("gpn_linear_bwd: workspace too small");
      return GPN_ERR_WS;
    }
    GPN_CHECK_ARG(x);
    float* partial = static_cast<float*>(ws);
    const int tile_rows = (size_t)kTileRows * (cin + cout + 1) * sizeof(float) <= 65536 ? kTileRows : kTileRows / 2;
    const size_t lds = (size_t)tile_rows * (cin + cout + 1) * sizeof(float);  // 22 KB for a 16 -> 27 head
    hipLaunchKernelGGL(linear_dw_partial_kernel, dim3(blocks), dim3(kThreads), lds, stream, x, dy, N, cin, cout, tile_rows, partial,
                       rows.dev);
    GPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(linear_dw_sum_kernel, dim3((cout * cin + cout + kThreads / 16 - 1) / (kThreads / 16)), dim3(kThreads), 0, stream,
                       (const float*)partial, blocks, cin, cout, dW, db);
    GPN_CHECK_LAUNCH();
  }
  return GPN_OK;
}
