// What does a workgroup that reads a device-side row count and returns cost?  (sizing of upper-bound launches whose live
// extent is a device counter: DESIGN "proposal stage without the host read")
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/empty_wg_probe tools/probes/empty_wg_probe.hip && gpurun_out/empty_wg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void probe(const int64_t* n_dev, int64_t per_wg, float* out) {
  const int64_t n = *n_dev;
  const int64_t row0 = (int64_t)blockIdx.x * per_wg;
  if (row0 >= n) return;
  out[row0 + threadIdx.x % per_wg] = 1.f;
}
__global__ __launch_bounds__(256) void probe_remap(const int64_t* n_dev, int64_t per_wg, float* out) {
  const int64_t n = *n_dev;
  const int64_t wg = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = wg * per_wg;
  if (row0 >= n) return;
  out[row0 + threadIdx.x % per_wg] = 1.f;
}
int main() {
  int64_t* n_dev; float* out;
  hipMalloc(&n_dev, 8); hipMalloc(&out, 64 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 200;
  for (int variant = 0; variant < 2; ++variant)
  for (int64_t wgs : {256, 2048, 8192, 20000, 40000, 80000, 160000}) {
    for (double live_frac : {1.0, 0.5, 0.1, 0.0}) {
      int64_t n = (int64_t)(wgs * 16 * live_frac);
      hipMemcpy(n_dev, &n, 8, hipMemcpyHostToDevice);
      for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, n_dev, 16, out);
      hipDeviceSynchronize();
      hipEventRecord(a, 0);
      for (int r = 0; r < reps; ++r) {
        if (variant == 0) hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, n_dev, 16, out);
        else hipLaunchKernelGGL(probe_remap, dim3(wgs), dim3(256), 0, 0, n_dev, 16, out);
      }
      hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%s wgs %7lld live %.2f : %.2f us per launch\n", variant ? "xcd-remap" : "linear   ", (long long)wgs, live_frac, ms * 1000 / reps);
    }
  }
  return 0;
}
