// optim.hip — Adam update of every parameter of the model in ONE launch (include/gpn.h section O).
//
// Reference: GAPartNet.configure_optimizers (network/model.py:1051-1055) = torch.optim.Adam(lr); the update rule below is
// torch's single-tensor Adam, operation by operation in fp32 (exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g,
// 1 - b2); denom = sqrt(exp_avg_sq) / sqrt(1 - b2^t) + eps; p.addcdiv_(exp_avg, denom, -lr / (1 - b1^t))).
// torch's own fused / foreach implementations spend ~1 ms of host time per step grouping the model's ~330 tensors and issue
// 9-15 launches; here the tensors are described ONCE by a device-resident table (pointers do not change from step to step:
// the executor's gradient buffer is persistent), a step is one launch and a few scalars.
#include "gpn_common.h"

#include <algorithm>

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 4096;  // elements per workgroup

// `gate` (optional): a device counter deciding whether this launch's tensors take a step at all - the proposal networks of a
// training step whose proposal count stayed on the device (include/gpn.h section DEV): with *gate == 0 the reference never ran
// those networks, their gradients are None and torch.optim.Adam leaves value, moments and step count alone.  Here the networks ran
// over zero rows and produced zero gradients; a step with them would still decay the moments and move the parameters by the
// momentum term.  So: *gate == 0 -> nothing is touched and *skipped += 1; later launches take their step number as
// step - *skipped (bias corrections recomputed from it in double, as the host does).
__global__ __launch_bounds__(kThreads) void adam_kernel(const gpn_adam_tensor_t* __restrict__ table,
                                                        const int32_t* __restrict__ block_first, int n_tensors, float b1w,
                                                        float b2, float b2w, float bc2_sqrt, float eps, float neg_step,
                                                        const int64_t* __restrict__ gate, int64_t* __restrict__ skipped,
                                                        double lr, double beta1, double beta2, int64_t step) {
  if (gate) {
    if (*gate == 0) {
      if (blockIdx.x == 0 && threadIdx.x == 0) *skipped += 1;  // (nobody reads it in this launch)
      return;
    }
    const int64_t sk = *skipped;
    if (sk > 0) {
      const double n = (double)(step - sk > 1 ? step - sk : 1);
      bc2_sqrt = (float)sqrt(1.0 - pow(beta2, n));
      neg_step = (float)(-lr / (1.0 - pow(beta1, n)));
    }
  }
  // tensor of this block: last t with block_first[t] <= blockIdx.x
  int lo = 0, hi = n_tensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const gpn_adam_tensor_t t = table[lo];
  const int64_t base = (int64_t)((int)blockIdx.x - block_first[lo]) * kChunk;
  float* __restrict__ p = static_cast<float*>(t.param);
  const float* __restrict__ g = static_cast<const float*>(t.grad);
  float* __restrict__ m = static_cast<float*>(t.exp_avg);
  float* __restrict__ v = static_cast<float*>(t.exp_avg_sq);
  for (int64_t i = base + threadIdx.x; i < base + kChunk && i < t.numel; i += kThreads) {
    const float gi = g[i];
    float mi = m[i], vi = v[i];
    mi = __fadd_rn(mi, __fmul_rn(b1w, __fsub_rn(gi, mi)));
    vi = __fadd_rn(__fmul_rn(vi, b2), __fmul_rn(b2w, __fmul_rn(gi, gi)));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), bc2_sqrt), eps);
    p[i] = __fadd_rn(p[i], __fmul_rn(neg_step, __fdiv_rn(mi, denom)));
    m[i] = mi;
    v[i] = vi;
  }
}

// up to kCopySegs (src, dst, n) fp32 segments copied by ONE launch, the segment table as a kernel argument: the gradients
// autograd allocates afresh every step (the dense heads' ~57 small tensors) go to the persistent buffers the Adam table points
// at.  torch._foreach_copy_ took its per-tensor path for them: 57 hipMemcpyAsync of 3.6 us each on the training stream.
constexpr int kCopySegs = 96;
struct CopyBatch {
  int n;
  gpn_copy_seg_t seg[kCopySegs];
};
__global__ __launch_bounds__(kThreads) void copy_many_kernel(const CopyBatch b) {
  const gpn_copy_seg_t s = b.seg[blockIdx.y];
  const float* __restrict__ src = static_cast<const float*>(s.src);
  float* __restrict__ dst = static_cast<float*>(s.dst);
  // 16 bytes per lane where both ends are aligned (round 6: bench.py puts the model's 31.6 MB of parameters back through this
  // kernel every step - 4-byte copies moved them at 1.8 TB/s)
  const bool wide = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  const int64_t n4 = wide ? s.numel >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads)
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < s.numel; i += (int64_t)gridDim.x * kThreads) dst[i] = src[i];
}

}  // namespace

extern "C" int gpn_copy_many(const gpn_copy_seg_t* segs_host, int n_segs, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(n_segs >= 0 && (n_segs == 0 || segs_host));
  for (int first = 0; first < n_segs; first += kCopySegs) {
    CopyBatch b;
    b.n = std::min(kCopySegs, n_segs - first);
    int64_t longest = 1;
    for (int i = 0; i < b.n; ++i) {
      b.seg[i] = segs_host[first + i];
      GPN_CHECK_ARG(b.seg[i].numel >= 0 && (b.seg[i].numel == 0 || (b.seg[i].src && b.seg[i].dst)));
      longest = std::max(longest, b.seg[i].numel);
    }
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)std::min<int64_t>(gpn::cdiv(longest, 4 * kThreads), 64), b.n), dim3(kThreads), 0, stream, b);
    GPN_CHECK_LAUNCH();
  }
  return GPN_OK;
}

extern "C" int gpn_adam_blocks(int64_t numel) { return (int)gpn::cdiv(numel > 0 ? numel : 1, kChunk); }

extern "C" int gpn_adam_step(const gpn_adam_tensor_t* table_dev, const int32_t* block_first_dev, int n_tensors, int n_blocks,
                             double lr, double beta1, double beta2, double eps, int64_t step, gpn_stream_t stream_) {
  return gpn_adam_step_gated(table_dev, block_first_dev, n_tensors, n_blocks, lr, beta1, beta2, eps, step, nullptr, nullptr, stream_);
}

extern "C" int gpn_adam_step_gated(const gpn_adam_tensor_t* table_dev, const int32_t* block_first_dev, int n_tensors, int n_blocks,
                                   double lr, double beta1, double beta2, double eps, int64_t step, const int64_t* gate_dev,
                                   int64_t* skipped_dev, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(n_tensors >= 0 && n_blocks >= 0 && step >= 1);
  GPN_CHECK_ARG((gate_dev == nullptr) == (skipped_dev == nullptr));
  if (n_tensors == 0 || n_blocks == 0) return GPN_OK;
  GPN_CHECK_ARG(table_dev && block_first_dev);
  // scalars in double like torch's Python-side arithmetic, rounded to fp32 once (1 - 0.999f would be off by 1.3e-5)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(n_blocks), dim3(kThreads), 0, stream, table_dev, block_first_dev, n_tensors,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(-lr / bc1),
                     gate_dev, skipped_dev, lr, beta1, beta2, step);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
