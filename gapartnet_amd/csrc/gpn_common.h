// gpn_common.h — shared host/device helpers for libgpn_hip.so (gfx950 only).
#pragma once
#include <cstring>  // must precede rocprim (texture_cache_iterator.hpp calls memset on the host)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/gpn.h"

namespace gpn {

void set_error(const char* fmt, ...);

// profiler hooks (gpn_prof.cpp): bracket a launch with hipEvents when profiling is on
struct ProfScope {
  int id;
  hipStream_t stream;
  hipEvent_t start = nullptr;
  ProfScope(int kernel_id, hipStream_t s, double flops, double bytes);
  ~ProfScope();
};

// per-device caches (occupancy, side streams, helper threads) are arrays indexed by the HIP device ordinal
constexpr int kMaxDevices = 64;

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// carve sub-buffers out of the caller's workspace
struct WsCarver {
  char* base;
  size_t used = 0, cap;
  WsCarver(void* ws, size_t bytes) : base(static_cast<char*>(ws)), cap(bytes) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    T* p = reinterpret_cast<T*>(base + used);
    used += bytes;
    return p;
  }
  bool ok() const { return used <= cap && (base != nullptr || used == 0); }
};

// gpn_spconv_fwd_ordered with an "add to out" mode (spconv_fwd.hip); used by the network executor's backward (net.hip)
int spconv_fwd_into(const float* in, const float* packed_w, const int32_t* nbr, const int32_t* nbr_p, const int32_t* perm, int K,
                    int64_t n_dst, int cin, int cout, float* out, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

// the masked-tile kernel (spconv_tiles.hip): which shapes it takes, and its launch
bool spconv_tiles_supported(int K, int64_t n_dst, int cin, int cout);
int spconv_tiles_launch(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                        int cin, int cout, int accumulate, float* out, hipStream_t stream);

}  // namespace gpn

#define GPN_CHECK_ARG(cond)                                                     \
  do {                                                                          \
    if (!(cond)) {                                                              \
      gpn::set_error("%s: bad argument: %s", __func__, #cond);                  \
      return GPN_ERR_ARG;                                                       \
    }                                                                           \
  } while (0)

#define GPN_CHECK_HIP(expr)                                                     \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      gpn::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
      return GPN_ERR_HIP;                                                       \
    }                                                                           \
  } while (0)

#define GPN_CHECK_LAUNCH()                                                      \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      gpn::set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e_)); \
      return GPN_ERR_HIP;                                                       \
    }                                                                           \
  } while (0)

#define GPN_CHECK_WS(carver)                                                    \
  do {                                                                          \
    if (!(carver).ok()) {                                                       \
      gpn::set_error("%s: workspace too small (%zu needed, %zu given)", __func__, (carver).used, (carver).cap); \
      return GPN_ERR_WS;                                                        \
    }                                                                           \
  } while (0)
