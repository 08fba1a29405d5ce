// gpn_common.h — shared host/device helpers for libgpn_hip.so (gfx950 only).
#pragma once
#include <cstdlib>
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
  int64_t tag = 0;
  ProfScope(int kernel_id, hipStream_t s, double flops, double bytes, int64_t tag = 0);
  // the same for a launch whose row count is a device counter: `bytes` / `flops` were computed for `bound` rows; gpn_prof_get reads
  // the counter and scales them to the live rows (round 5: the BatchNorm passes of the device-counted proposal networks were
  // accounted at their 2 N bound - a family fraction above 1)
  ProfScope(int kernel_id, hipStream_t s, double flops, double bytes, const int64_t* rows_dev, int64_t bound, int64_t tag = 0);
  ~ProfScope();
};
// shape of a conv launch as a profiler tag (gpn_prof_get_launches): rows of the launch's output, taps, channel blocks, and
// whether the launch carries a second problem (paired pass)
inline int64_t prof_shape_tag(int K, int64_t n_dst, int cin, int cout, bool twin) {
  return ((int64_t)(twin ? 1 : 0) << 62) | ((int64_t)(K & 63) << 48) | ((int64_t)((cin / 16) & 255) << 40) |
         ((int64_t)((cout / 16) & 255) << 32) | (n_dst & 0xffffffffll);
}

// ---- launches whose extent is a DEVICE counter (round 4: the proposal stage without a host read) -----------------------------
// A tensor of a data-dependent size lives in a buffer of its upper bound; its live row count is an int64 on the device that the
// producing kernels write.  Consumers take (n = the bound, DevRows{dev, plan}): kernels read the live count themselves
// (live_rows), walk their work items with a grid-stride loop - so correctness never depends on the host's guess - and the
// host sizes grids and picks kernel variants from `plan`, its estimate of the count (the previous step's value; <= 0: none,
// the bound is used).  dev == nullptr is the ordinary case: n is the exact count, every loop runs once, same code path.
struct DevRows {
  const int64_t* dev = nullptr;
  int64_t plan = 0;
};
inline int64_t plan_rows(int64_t n, const DevRows& r) {
  if (!r.dev || r.plan <= 0) return n;
  return r.plan < n ? r.plan : n;
}
__device__ __forceinline__ int64_t live_rows(const int64_t* dev, int64_t n) {
  if (!dev) return n;
  const int64_t v = __builtin_nontemporal_load(dev);
  return v < n ? (v > 0 ? v : 0) : n;
}
// workgroups of a launch whose work items are counted by a device counter: enough for 1.5 x the planned count (at least
// `floor_wgs`, so that a stale plan still fills the chip), never more than the bound needs; a multiple of `mult`
inline unsigned dev_grid(int64_t wgs_bound, int64_t wgs_plan, bool dev, int mult = 1, int64_t floor_wgs = 1024) {
  int64_t g = wgs_bound;
  if (dev) {
    int64_t want = wgs_plan + wgs_plan / 2 + mult;
    if (want < floor_wgs) want = floor_wgs;
    if (want < g) g = want;
  }
  if (g < 1) g = 1;
  g = (g + mult - 1) / mult * mult;
  return (unsigned)g;
}

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

// BatchNorm column sums accumulated by the producing conv launch (bn_stats.h): what a conv's epilogue adds to.
// slab == nullptr: nothing.  x == nullptr: statistics of the conv's OUTPUT (sum out, sum out^2; forward).  x != nullptr: the
// launch is a dgrad that writes the final gradient g of a BatchNorm's output y = act(bn(x) [+ res]): sum g', sum g' xhat with
// g' = g masked by the ReLU (y > 0) and xhat = (x - mean) invstd (backward).
constexpr int kStatSlots = 32;
// slot sets a launch over `rows` output rows spreads its contributions over (a power of two <= kStatSlots): enough that an
// address takes <= 64 of the launch's atomics, few enough that the apply pass's fold of a small layer is one or two words
// per thread - the deep levels (large C, few tiles) would otherwise fold 4 C x 32 words in every workgroup
inline int stat_slot_count(int64_t rows) {
  const int64_t tiles = (rows + 15) / 16;
  int s = 1;
  while (s < kStatSlots && tiles > 64 * (int64_t)s) s *= 2;
  return s;
}
// A second, independent problem of the same shape over the same rulebook, computed by the workgroups with blockIdx.y == 1
// of the SAME launch (the executor's paired passes over two structurally identical networks, net.hip): its operand /
// result / BatchNorm-sum pointers.  in == nullptr: none.
// An eval-mode BatchNorm (+ residual add, + ReLU) applied by the conv launch that produces its input, in the epilogue (round 6:
// an inference pass - running statistics, no backward pass to follow - has no BatchNorm launches at all).  Per output element
// exactly the arithmetic of the stand-alone pass (bn.hip, bn_apply_eval_kernel): ((v - mean) * (1 / sqrt(var + eps))) * weight +
// bias, then + res, then max(0, .) - the same bits.  mean == nullptr: none.
struct ConvAffine {
  const float* mean = nullptr;
  const float* var = nullptr;
  const float* weight = nullptr;
  const float* bias = nullptr;
  const float* res = nullptr;  // [rows, cout] residual, or nullptr
  float eps = 0.f;
  int relu = 0;
};
struct ConvTwin {
  const float* in = nullptr;
  const float* packed = nullptr;
  float* out = nullptr;
  unsigned long long* slab = nullptr;
  const float* x = nullptr;
  const float* y = nullptr;
  const float* mean = nullptr;
  const float* invstd = nullptr;
  ConvAffine ep;  // the second problem's BatchNorm (eps and relu are the first one's)
};
struct ConvStats {
  unsigned long long* slab = nullptr;  // [kStatSlots][4][C] fixed-point words (the first slot_mask + 1 sets used), zeroed by the caller
  int slot_mask = kStatSlots - 1;      // stat_slot_count(rows of the tensor the sums are over) - 1
  const float* x = nullptr;
  const float* y = nullptr;
  const float* mean = nullptr;
  const float* invstd = nullptr;
  int relu = 0;
  ConvAffine ep;  // an inference pass's BatchNorm on the conv's output (never together with a slab)
  ConvTwin twin;  // (rides with the sums: both are "what else this launch does", and every launch path already carries this struct)
};
#ifdef __HIPCC__
// the four per-column constants of a ConvAffine, and its application to one output element e = row * cout + col
struct AffineCol {
  float mu, is, w, b;
};
__device__ __forceinline__ AffineCol affine_col(const ConvAffine& ep, uint32_t col) {
  return AffineCol{ep.mean[col], 1.0f / sqrtf(ep.var[col] + ep.eps), ep.weight[col], ep.bias[col]};
}
__device__ __forceinline__ float affine_apply(const ConvAffine& ep, const AffineCol& c, float v, uint32_t e) {
  v = (v - c.mu) * c.is * c.w + c.b;
  if (ep.res) v += ep.res[e];
  if (ep.relu) v = v > 0.f ? v : 0.f;
  return v;
}
#endif
// pointer sets of the BatchNorm apply passes; a launch takes two and its workgroups pick by blockIdx.y (twin launches as above)
struct BnFwdPtrs {
  const float* x = nullptr;
  const float* res = nullptr;
  const void* partial = nullptr;
  const float* weight = nullptr;
  const float* bias = nullptr;
  float* y = nullptr;
  float* mean = nullptr;
  float* invstd = nullptr;
  float* running_mean = nullptr;
  float* running_var = nullptr;
};
struct BnBwdPtrs {
  const float* x = nullptr;
  const float* y = nullptr;
  const float* dy = nullptr;
  const void* partial = nullptr;
  const float* mean = nullptr;
  const float* invstd = nullptr;
  const float* weight = nullptr;
  float* dx = nullptr;
  float* dres = nullptr;
  float* dweight = nullptr;
  float* dbias = nullptr;
};
inline size_t stat_slab_bytes(int C) { return align_up((size_t)kStatSlots * 4 * C * sizeof(unsigned long long)); }

// gpn_spconv_fwd_ordered with an "add to out" mode and optional BatchNorm sums (spconv_fwd.hip); used by the network executor
int spconv_fwd_into(const float* in, const float* packed_w, const int32_t* nbr, const int32_t* nbr_p, const int32_t* perm, int K,
                    int64_t n_dst, int cin, int cout, float* out, int accumulate, const ConvStats& stats, void* ws,
                    size_t ws_bytes, hipStream_t stream, const DevRows& rows = DevRows());
// true if a conv of this shape runs on a kernel whose epilogue can accumulate ConvStats (masked-tile or direct kernel)
bool spconv_fwd_accumulates_stats(int K, int64_t n_dst, int cin, int cout);
bool spconv_fwd_accumulates_stats(int K, int64_t n_dst, int cin, int cout, const DevRows& rows);
// true if a conv of this shape runs on a kernel whose epilogue can apply a ConvAffine (an inference pass's BatchNorm)
bool spconv_fwd_applies_affine(int K, int64_t n_dst, int cin, int cout, const DevRows& rows);
// the masked-tile kernel (spconv_tiles.hip): which shapes it takes, and its launch
bool spconv_tiles_supported(int K, int64_t n_dst, int cin, int cout);
int spconv_tiles_launch(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                        int cin, int cout, int accumulate, const ConvStats& stats, float* out, hipStream_t stream,
                        const DevRows& rows = DevRows());
// the masked tap-split kernel (spconv_msplit.hip, round 6): k = 27 / 8 layers below the masked-tile kernel's size
bool spconv_msplit_supported(int K, int64_t n_dst, int cin, int cout);
int spconv_msplit_launch(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                         int cin, int cout, int accumulate, const ConvStats& stats, float* out, hipStream_t stream,
                         const DevRows& rows = DevRows());
// weight-gradient contraction and its (batched) slice sums (spconv.hip); used by gpn_spconv_wgrad and the network executor
constexpr int kWgradReduceJobs = 24;
constexpr int kWgradSets = 4;
struct WgradSet {  // one layer's operands
  const float* in;
  const float* dout;
  const int32_t *pair_src, *pair_dst, *tile_off;
  float* partial;
};
struct WgradSets {  // layers of one shape contracted by one launch (kernel argument)
  int n = 0;
  WgradSet s[kWgradSets];
};
struct WgradReduceJob {
  const float* partial;
  float* dW;
  int64_t elems;
  int S, K, cin, cout, oki, few;
};
int wgrad_slices(int K, int cin, int cout, int64_t n_dst);
int wgrad_contract(const WgradSets& sets, int K, int64_t n_dst, int cin, int cout, int S, hipStream_t stream,
                   const int64_t* n_dst_dev = nullptr);
WgradReduceJob wgrad_reduce_job(const float* partial, int S, int K, int cin, int cout, int flags, float* dW);
int wgrad_reduce_many(const WgradReduceJob* jobs, int n, hipStream_t stream);
// gpn_rulebook_level_counts with row count and level-0 extent on the device (rulebook.hip; used by gpn_voxelize_scenes)
int rulebook_level_counts_dev(const int32_t* indices, int64_t n_max, const int64_t* n_dev, int64_t batch_size,
                              const int64_t* max_coord_dev, int n_levels, int64_t* counts, void* ws, size_t ws_bytes,
                              hipStream_t stream);
// BatchNorm apply passes over sums a conv launch accumulated (bn.hip); bn_two_pass: the shapes that take them
bool bn_two_pass(int64_t N, int C);
// (partial = the slab; `twin`: a second BatchNorm of the same shape in the same launch, or nullptr)
int bn_fwd_train_fused(const BnFwdPtrs& p, const BnFwdPtrs* twin, int64_t N, int C, float eps, float momentum, int relu,
                       hipStream_t stream, const DevRows& rows = DevRows());
int bn_bwd_fused(const BnBwdPtrs& p, const BnBwdPtrs* twin, int64_t N, int C, int relu, int training, hipStream_t stream,
                 const DevRows& rows = DevRows());
// gpn_bn_fwd_train / gpn_bn_fwd_eval / gpn_bn_bwd with the row count optionally on the device (DevRows)
int bn_fwd_train_rows(const float* x, const float* res, const float* weight, const float* bias, int64_t N, const DevRows& rows,
                      int C, float eps, float momentum, int relu, float* y, float* mean, float* invstd, float* running_mean,
                      float* running_var, void* ws, size_t ws_bytes, hipStream_t stream);
int bn_fwd_eval_rows(const float* x, const float* res, const float* weight, const float* bias, const float* mean,
                     const float* invstd, int64_t N, const DevRows& rows, int C, int relu, float* y, hipStream_t stream);
int bn_fwd_eval_running(const BnFwdPtrs& pa, const BnFwdPtrs* pb, int64_t N, const DevRows& rows, int C, float eps, int relu,
                        hipStream_t stream);
int bn_bwd_rows(const float* x, const float* y, const float* dy, const float* weight, const float* mean, const float* invstd,
                int64_t N, const DevRows& rows, int C, int relu, int training, float* dx, float* dres, float* dweight,
                float* dbias, void* ws, size_t ws_bytes, hipStream_t stream);

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
