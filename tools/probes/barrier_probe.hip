// What does a layer boundary cost INSIDE one persistent launch (a grid barrier + the hand-over of the layer's output to every
// other workgroup) against the same boundary as a dependent kernel launch?  ("collapse the small shapes into
// persistent kernels ... if a barrier costs more than a launch floor, commit the trace that shows it" -> profiles/r05_barrier_probe.txt)
//
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/barrier_probe tools/probes/barrier_probe.hip && gpurun_out/barrier_probe
//
// Every variant runs the same chain of N dependent "layers": workgroup b of layer i reads the 1 KiB (or `bytes`) record that
// workgroup (b + 37) % G wrote in layer i - 1, adds one to every word and writes its own record; the final records must all equal N
// (a stale read anywhere shows up as a wrong count).  Variants:
//   launches      one kernel launch per layer (the product's structure: hipLaunchKernelGGL back to back on one stream)
//   flat          one launch; layers separated by a single device-scope counter barrier, records written / read with sc1 accesses
//   xcd-sc1       one launch; hierarchical barrier (per-XCD arrival counter -> top counter -> generation word), sc1 records
//   xcd-fence     the same barrier; plain stores, the XCD's last arriver writes the L2 back (release fence), every workgroup
//                 invalidates (acquire fence) after the barrier - what a kernel boundary does, once per XCD instead of per launch
//   one-xcd       only the workgroups that landed on XCD 0 take part (32 of 256): barrier on that XCD's counter alone, sc1 records
//                 (the "XCD-resident sub-network" of the review: one L2, no cross-XCD traffic - and 1/8 of the chip's CUs)
// Spin loops give up after 20 ms of s_memrealtime (a flag is set and the run reports FAILED) so that a mistake cannot hang the box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

struct Bar {
  unsigned reg[8 * 16];   // workgroups registered per XCD (one 64-byte line each)
  unsigned xcc[8 * 16];   // arrivals per XCD (monotonic)
  unsigned total, top, gen, failed;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }  // HW_REG_XCC_ID, 4 bits

// wait until *p >= want; false after ~20 ms
__device__ __forceinline__ bool spin_ge(const unsigned* p, unsigned want, Bar* b) {
  const unsigned long long t0 = wall_clock64();
  while (ld_sc1(p) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 2000000ull || ld_sc1(&b->failed)) {  // 100 MHz
      st_sc1(&b->failed, 1u);
      return false;
    }
  }
  return true;
}

enum { FLAT = 0, XCD_SC1 = 1, XCD_FENCE = 2, ONE_XCD = 3 };

template <int MODE>
__global__ __launch_bounds__(256) void persistent(Bar* b, unsigned* recs, int words, int layers) {
  __shared__ unsigned s_n, s_nx, s_ok;
  const unsigned G = gridDim.x;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) {
    atomicAdd(&b->reg[x * 16], 1u);
    __builtin_amdgcn_s_waitcnt(0);
    atomicAdd(&b->total, 1u);
    s_ok = spin_ge(&b->total, G, b) ? 1u : 0u;  // every workgroup of the grid is resident and has registered
    s_n = ld_sc1(&b->reg[x * 16]);
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += ld_sc1(&b->reg[i * 16]) != 0u;
    s_nx = nx;
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned n_here = s_n, n_xcds = s_nx;
  if (MODE == ONE_XCD && x != 0) return;
  // my index among the participants: for ONE_XCD a ticket on XCD 0; otherwise blockIdx
  unsigned me = blockIdx.x, P = G;
  if (MODE == ONE_XCD) {
    __shared__ unsigned s_me;
    if (threadIdx.x == 0) s_me = atomicAdd(&b->top, 1u);  // (top is unused as a barrier word in this mode)
    __syncthreads();
    me = s_me, P = n_here;
  }
  unsigned* mine = recs + (size_t)me * words;
  const unsigned* theirs = recs + (size_t)((me + 37u) % P) * words;
  const size_t half = (size_t)G * words;  // ping-pong halves
  for (int l = 1; l <= layers; ++l) {
    const unsigned* src = theirs + ((l - 1) & 1) * half;
    unsigned* dst = mine + (l & 1) * half;
    for (int w = threadIdx.x; w < words; w += 256) {
      if (MODE == XCD_FENCE) dst[w] = src[w] + 1u;
      else st_sc1(dst + w, ld_sc1(src + w) + 1u);
    }
    // ---- barrier ----
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
      bool ok = true;
      if (MODE == FLAT) {
        atomicAdd(&b->top, 1u);
        ok = spin_ge(&b->top, G * (unsigned)l, b);
      } else if (MODE == ONE_XCD) {
        atomicAdd(&b->xcc[0], 1u);
        ok = spin_ge(&b->xcc[0], n_here * (unsigned)l, b);
      } else {
        const unsigned old = atomicAdd(&b->xcc[x * 16], 1u);
        if (old == n_here * (unsigned)l - 1u) {  // this XCD's last arriver
          if (MODE == XCD_FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          const unsigned old2 = atomicAdd(&b->top, 1u);
          if (old2 == n_xcds * (unsigned)l - 1u) st_sc1(&b->gen, (unsigned)l);
        }
        ok = spin_ge(&b->gen, (unsigned)l, b);
        if (MODE == XCD_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    if (!s_ok) return;
  }
}

__global__ __launch_bounds__(256) void layer_kernel(const unsigned* src_half, unsigned* dst_half, int words) {
  const unsigned me = blockIdx.x, G = gridDim.x;
  const unsigned* src = src_half + (size_t)((me + 37u) % G) * words;
  unsigned* dst = dst_half + (size_t)me * words;
  for (int w = threadIdx.x; w < words; w += 256) dst[w] = src[w] + 1u;
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 200;
  const int G = 256;
  Bar* bar;
  unsigned* recs;
  const int max_words = 16384;  // 64 KiB per workgroup
  CHECK(hipMalloc(&bar, sizeof(Bar)));
  CHECK(hipMalloc(&recs, (size_t)2 * G * max_words * 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  std::vector<unsigned> host((size_t)G * max_words);
  printf("# %d workgroups x 256 threads, %d dependent layers per run; us per layer (best of 5 runs); errors = records != %d\n", G, layers, layers);
  for (int words : {256, 4096, 16384}) {
    for (int variant = -1; variant < 4; ++variant) {
      double best = 1e30;
      long errors = 0;
      unsigned failed = 0;
      unsigned participants = G;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipMemset(bar, 0, sizeof(Bar)));
        CHECK(hipMemset(recs, 0, (size_t)2 * G * max_words * 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, 0));
        if (variant < 0) {
          for (int l = 1; l <= layers; ++l)
            hipLaunchKernelGGL(layer_kernel, dim3(G), dim3(256), 0, 0, recs + ((l - 1) & 1) * (size_t)G * words, recs + (l & 1) * (size_t)G * words, words);
        } else if (variant == FLAT) {
          hipLaunchKernelGGL(persistent<FLAT>, dim3(G), dim3(256), 0, 0, bar, recs, words, layers);
        } else if (variant == XCD_SC1) {
          hipLaunchKernelGGL(persistent<XCD_SC1>, dim3(G), dim3(256), 0, 0, bar, recs, words, layers);
        } else if (variant == XCD_FENCE) {
          hipLaunchKernelGGL(persistent<XCD_FENCE>, dim3(G), dim3(256), 0, 0, bar, recs, words, layers);
        } else {
          hipLaunchKernelGGL(persistent<ONE_XCD>, dim3(G), dim3(256), 0, 0, bar, recs, words, layers);
        }
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms * 1000.0 / layers < best) best = ms * 1000.0 / layers;
        Bar hb;
        CHECK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        failed |= hb.failed;
        if (variant == ONE_XCD) participants = hb.reg[0];
        CHECK(hipMemcpy(host.data(), recs + (layers & 1) * (size_t)G * words, (size_t)participants * words * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)participants * words; ++i) errors += host[i] != (unsigned)layers;
      }
      const char* names[] = {"launches ", "flat     ", "xcd-sc1  ", "xcd-fence", "one-xcd  "};
      printf("record %6d B  %s  %7.2f us/layer  participants %3u  errors %ld%s\n", words * 4, names[variant + 1], best, participants, errors,
             failed ? "  FAILED (a spin loop gave up)" : "");
      fflush(stdout);
    }
  }
  return 0;
}
