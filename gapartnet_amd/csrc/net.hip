// net.hip — kernel family U: layer-program executor for the sparse residual U-Net (include/gpn.h section U).
//
// Reference behaviour: network/backbone.py:40-49 (ResBlock.forward), :126-141 (UBlock.forward), :150-155
// (SparseUNet.forward) — a Python walk over ~200 spconv / BatchNorm1d modules per forward, and autograd's walk back.
// Here the host describes that walk once as a flat list of CONV / BN / CONCAT ops over numbered activation slots and
// this file issues every launch of the forward (or of the backward, in reverse program order) from one native loop.
// The arithmetic is exactly that of the per-layer entry points (gpn_spconv_fwd_w, gpn_spconv_wgrad, gpn_bn_*), which
// this file calls; the kernels of its own are the column concat / split, the gradient accumulation and a batched
// weight packer (every weight of the program packed by a few launches at the start of a pass instead of one launch
// per layer).  In backward the weight-gradient contractions run on a second (lower-priority) stream: they are off
// the critical path (nothing in the backward pass consumes dW), so they fill the CUs the dependent dgrad / BN chain
// of the small, deep levels leaves idle.  Round 3: BatchNorm sums ride in the conv / dgrad epilogues (bn_stats.h), and two
// structurally identical networks over the same rulebooks run as PAIRED passes - one launch per layer for both (NetSet below).
#include <unistd.h>

#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gpn_common.h"
#include "spconv_pack.h"

namespace {

constexpr int kThreads = 256;

// BatchNorm in the conv launches - training: the sums in the conv / dgrad epilogues (bn_stats.h); inference: the whole BatchNorm
// (gpn::ConvAffine) - on / off: gpn_net_bn_fusion(0) restores the separate launches (A/B measurements, and the tests that compare
// the two forms).  (Round 6: no environment readers are left in the library - every switch is an entry point that a test flips.
// Round 3's trap, for the record: hipcc numbers the lambdas of namespace-scope initialisers per anonymous-namespace block, so a
// lambda in a second block got the mangled name - and, at link time, the body - of the first block's: a knob that silently read
// another knob's variable.)
std::atomic<int> g_bn_fusion{1};

// dst[r, 0:ca] = a[r, :], dst[r, ca:ca+cb] = b[r, :]   (float4 granularity; channel counts are multiples of 4).  Two pointer
// sets per launch, picked by blockIdx.y (paired passes, see NetSet below)
struct ConcatPtrs {
  const float4* a;
  const float4* b;
  float4* dst;
};
__global__ __launch_bounds__(kThreads) void concat_kernel(ConcatPtrs pa, ConcatPtrs pb, int64_t rows, int ca4, int cb4,
                                                          const int64_t* __restrict__ rows_dev) {
  rows = gpn::live_rows(rows_dev, rows);
  const ConcatPtrs& p = blockIdx.y ? pb : pa;
  const float4* __restrict__ a = p.a;
  const float4* __restrict__ b = p.b;
  float4* __restrict__ dst = p.dst;
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t r = i / c4;
    const int c = (int)(i - r * c4);
    dst[i] = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
  }
}

// the transpose: da (+)= dsrc[:, 0:ca], db (+)= dsrc[:, ca:]; acc_a / acc_b select overwrite (0) or accumulate (1)
struct SplitPtrs {
  const float4* dsrc;
  float4* da;
  float4* db;
};
__global__ __launch_bounds__(kThreads) void split_kernel(SplitPtrs pa, SplitPtrs pb, int64_t rows, int ca4, int cb4, int acc_a,
                                                         int acc_b, const int64_t* __restrict__ rows_dev) {
  rows = gpn::live_rows(rows_dev, rows);
  const SplitPtrs& pp = blockIdx.y ? pb : pa;
  const float4* __restrict__ dsrc = pp.dsrc;
  float4* __restrict__ da = pp.da;
  float4* __restrict__ db = pp.db;
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t r = i / c4;
    const int c = (int)(i - r * c4);
    const float4 g = dsrc[i];
    float4* p = c < ca4 ? da + (r * ca4 + c) : db + (r * cb4 + (c - ca4));
    if (c < ca4 ? acc_a : acc_b) {
      float4 o = *p;
      o.x += g.x, o.y += g.y, o.z += g.z, o.w += g.w;
      *p = o;
    } else {
      *p = g;
    }
  }
}

__global__ __launch_bounds__(kThreads) void accumulate_kernel(float4* __restrict__ dst, const float4* __restrict__ src,
                                                              int64_t total4, int c4, const int64_t* __restrict__ rows_dev) {
  if (rows_dev) total4 = gpn::live_rows(rows_dev, total4 / c4) * c4;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total4; i += (int64_t)gridDim.x * kThreads) {
    float4 o = dst[i];
    const float4 g = src[i];
    o.x += g.x, o.y += g.y, o.z += g.z, o.w += g.w;
    dst[i] = o;
  }
}

// batched weight packing: up to kPackBatch weights per launch, descriptors passed by value in the kernel arguments
constexpr int kPackBatch = 24;
struct PackDesc {
  const float* W;
  float* packed;
  int K, cin_w, cout_w, flags;
};
struct PackBatch {
  PackDesc d[kPackBatch];
};

__global__ __launch_bounds__(kThreads) void pack_many_kernel(PackBatch batch) {
  const PackDesc d = batch.d[blockIdx.y];
  const int64_t total = (int64_t)d.K * d.cin_w * d.cout_w;
  for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (int64_t)gridDim.x * kThreads)
    d.packed[t] = gpn::packed_weight_element(d.W, d.K, d.cin_w, d.cout_w, d.flags, t);
}

inline int grid_for(int64_t total) {
  int64_t g = gpn::cdiv(total, kThreads);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

inline size_t slot_bytes(const gpn_net_slot_t& s) { return (size_t)s.rows * s.channels * sizeof(float); }
inline gpn::DevRows dev_rows(const gpn_net_slot_t& s) { return gpn::DevRows{s.rows_dev, s.rows_plan}; }
inline int64_t plan_of(const gpn_net_slot_t& s) { return gpn::plan_rows(s.rows, dev_rows(s)); }

// partials of the layers whose slice sums are batched into one launch: at most this much (they are written and read back
// within a few launches - the bound keeps them inside the 256 MB memory-side cache)
constexpr size_t kWgradBatchBytes = (size_t)96 << 20;

struct Need {
  size_t tmp = 0;     // gradient staging buffer (largest slot that can receive a second gradient)
  size_t op = 0;      // largest per-op workspace of the main chain (conv tap-split partials, BN partials)
  size_t wgrad = 0;   // weight-gradient partials (side stream): room for a batch of layers whose slice sums run in one launch
  size_t packed = 0;  // all packed weights of the program
  size_t stats = 0;   // BatchNorm sum slabs the conv epilogues add to (bn_stats.h), one per BN op
};

Need workspace_need(const gpn_net_op_t* ops, int n_ops, const gpn_net_slot_t* slots, const gpn_net_rulebook_t* rbs,
                    const gpn_net_conv_t* convs) {
  Need n;
  size_t wgrad_sum = 0;
  for (int i = 0; i < n_ops; ++i) {
    const gpn_net_op_t& op = ops[i];
    const gpn_net_slot_t& s0 = slots[op.src0];
    if (op.kind == GPN_NET_CONV) {
      const gpn_net_rulebook_t& rb = rbs[op.rulebook];
      const gpn_net_conv_t& cv = convs[op.param];
      size_t w = gpn_spconv_fwd_ws_bytes(rb.K, rb.n_dst, cv.cin, cv.cout);
      size_t wt = gpn_spconv_fwd_ws_bytes(rb.K, rb.n_src, cv.cout, cv.cin);
      n.op = std::max(n.op, std::max(w, wt));
      const size_t wg = gpn_spconv_wgrad_ws_bytes(rb.K, cv.cin, cv.cout, rb.n_dst);
      n.wgrad = std::max(n.wgrad, wg);
      wgrad_sum += wg;
      n.packed += gpn::align_up((size_t)rb.K * cv.cin * cv.cout * sizeof(float));
      n.tmp = std::max(n.tmp, slot_bytes(s0));
    } else if (op.kind == GPN_NET_BN) {
      n.op = std::max(n.op, gpn_bn_ws_bytes(s0.rows, s0.channels));
      n.stats += gpn::stat_slab_bytes(s0.channels);
      if (op.src1 >= 0) n.tmp = std::max(n.tmp, slot_bytes(slots[op.src1]));
    }
  }
  n.tmp = gpn::align_up(n.tmp);
  n.op = gpn::align_up(n.op);
  n.wgrad = gpn::align_up(std::max(n.wgrad, std::min(wgrad_sum, kWgradBatchBytes)));
  return n;
}

// pack the weight of every CONV op into `area` (transposed [+ tap-reversed] for the backward pass); packed_of[i] is the
// packed weight of op i.  ceil(n_conv / kPackBatch) launches.
int pack_program(const gpn_net_op_t* ops, int n_ops, const gpn_net_rulebook_t* rbs, const gpn_net_conv_t* convs,
                 bool transposed, char* area, std::vector<const float*>& packed_of, hipStream_t stream) {
  packed_of.assign(n_ops, nullptr);
  PackBatch batch;
  int fill = 0;
  int64_t max_total = 0;
  size_t off = 0;
  auto flush = [&]() -> int {
    if (fill == 0) return GPN_OK;
    int gx = (int)std::min<int64_t>(gpn::cdiv(max_total, kThreads), 256);
    hipLaunchKernelGGL(pack_many_kernel, dim3(gx, fill), dim3(kThreads), 0, stream, batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      gpn::set_error("gpn_net: weight packing launch failed: %s", hipGetErrorString(e));
      return GPN_ERR_HIP;
    }
    fill = 0;
    max_total = 0;
    return GPN_OK;
  };
  for (int i = 0; i < n_ops; ++i) {
    if (ops[i].kind != GPN_NET_CONV) continue;
    const gpn_net_rulebook_t& rb = rbs[ops[i].rulebook];
    const gpn_net_conv_t& cv = convs[ops[i].param];
    float* dst = reinterpret_cast<float*>(area + off);
    off += gpn::align_up((size_t)rb.K * cv.cin * cv.cout * sizeof(float));
    packed_of[i] = dst;
    int flags = GPN_LAYOUT_OKI;
    if (transposed) flags |= GPN_PACK_TRANSPOSE | (rb.reverse_taps ? GPN_PACK_REVERSE : 0);
    batch.d[fill++] = PackDesc{cv.W, dst, rb.K, cv.cin, cv.cout, flags};
    max_total = std::max<int64_t>(max_total, (int64_t)rb.K * cv.cin * cv.cout);
    if (fill == kPackBatch) {
      int rc = flush();
      if (rc) return rc;
    }
  }
  return flush();
}

// the second stream (and the events that order it against the caller's stream), one set per device
constexpr int kForkEvents = 64;
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork[kForkEvents] = {};
  hipEvent_t join = nullptr;
  int next = 0;
};

SideStream* side_stream() {
  static SideStream per_device[gpn::kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= gpn::kMaxDevices) return nullptr;
  SideStream& s = per_device[dev];
  if (!s.stream) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = numerically largest = lowest priority
    // (confining this stream to a subset of the CUs - hipExtStreamCreateWithCUMask, rounds 3 and 4 - changed nothing at any mask
    // that left it >= 128 CUs and cost 20 % at 64: profiles/r04_cu_mask.txt; removed)
    if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
    for (auto& e : s.fork)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &s;
}

int check_program(const char* who, const gpn_net_op_t* ops, int n_ops, const gpn_net_slot_t* slots, int n_slots,
                  const gpn_net_rulebook_t* rbs, int n_rbs, const gpn_net_conv_t* convs, int n_convs,
                  const gpn_net_bn_t* bns, int n_bns) {
  if (!ops || !slots || n_ops < 0 || n_slots < 1 || (n_rbs && !rbs) || (n_convs && !convs) || (n_bns && !bns)) {
    gpn::set_error("%s: null table", who);
    return GPN_ERR_ARG;
  }
  for (int i = 0; i < n_ops; ++i) {
    const gpn_net_op_t& op = ops[i];
    auto bad = [&](const char* what) {
      gpn::set_error("%s: op %d (kind %d): %s", who, i, op.kind, what);
      return GPN_ERR_ARG;
    };
    if (op.src0 < 0 || op.src0 >= n_slots || op.dst < 0 || op.dst >= n_slots || op.src1 >= n_slots) return bad("slot index out of range");
    const gpn_net_slot_t &s0 = slots[op.src0], &d = slots[op.dst];
    if (s0.rows < 1 || d.rows < 1) return bad("empty slot");
    if (op.kind == GPN_NET_CONV) {
      if (op.rulebook < 0 || op.rulebook >= n_rbs || op.param < 0 || op.param >= n_convs) return bad("table index out of range");
      const gpn_net_rulebook_t& rb = rbs[op.rulebook];
      const gpn_net_conv_t& cv = convs[op.param];
      if (rb.n_src != s0.rows || rb.n_dst != d.rows || cv.cin != s0.channels || cv.cout != d.channels)
        return bad("slot shape does not match rulebook / weight");
      if (!rb.nbr || !cv.W) return bad("null rulebook / weight pointer");
    } else if (op.kind == GPN_NET_BN) {
      if (op.param < 0 || op.param >= n_bns) return bad("table index out of range");
      if (bns[op.param].C != s0.channels || d.channels != s0.channels || d.rows != s0.rows) return bad("slot shape does not match BatchNorm");
      if (op.src1 >= 0 && (slots[op.src1].rows != s0.rows || slots[op.src1].channels != s0.channels)) return bad("residual shape");
    } else if (op.kind == GPN_NET_CONCAT) {
      if (op.src1 < 0) return bad("concat needs two inputs");
      const gpn_net_slot_t& s1 = slots[op.src1];
      if (s1.rows != s0.rows || d.rows != s0.rows || d.channels != s0.channels + s1.channels || s0.channels % 4 || s1.channels % 4)
        return bad("concat shapes");
    } else {
      return bad("unknown op kind");
    }
  }
  return GPN_OK;
}

}  // namespace

// conv-epilogue BatchNorm sums on (1, default) / off (0); on < 0 queries.  Returns the previous setting.
extern "C" int gpn_net_bn_fusion(int on) {
  return on < 0 ? g_bn_fusion.load(std::memory_order_relaxed) : g_bn_fusion.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

namespace {
std::atomic<int>& wgrad_group_setting();
}
extern "C" int gpn_net_wgrad_group(int layers) {
  std::atomic<int>& g = wgrad_group_setting();
  if (layers < 1) return g.load(std::memory_order_relaxed);
  return g.exchange(layers > gpn::kWgradSets ? gpn::kWgradSets : layers, std::memory_order_relaxed);
}

extern "C" size_t gpn_net_ws_bytes(const gpn_net_op_t* ops, int n_ops, const gpn_net_slot_t* slots, int n_slots,
                                   const gpn_net_rulebook_t* rulebooks, const gpn_net_conv_t* convs) {
  (void)n_slots;
  if (!ops || !slots) return 0;
  const Need n = workspace_need(ops, n_ops, slots, rulebooks, convs);
  return n.tmp + n.op + n.wgrad + n.packed + n.stats;
}

namespace {

// One or two networks per pass.  Two ("paired" passes): structurally identical programs over the SAME rulebooks - the
// ScoreNet and NPCS-Net U-Nets of network/model.py:116-118, which read the same proposal grid - whose layers are launched
// TOGETHER: every conv / BatchNorm-apply / concat launch computes layer i of both networks (blockIdx.y picks the network's
// pointer set: gpn::ConvTwin, gpn::BnFwdPtrs), so a pass over two small networks costs the launches of one.  The arithmetic
// of each network is that of its own single pass (same kernels, same summation orders).
struct NetSet {
  gpn_net_slot_t* slots;
  const gpn_net_conv_t* convs;
  const gpn_net_bn_t* bns;
};

int net_forward_impl(const char* who, const gpn_net_op_t* ops, int n_ops, const NetSet* nets, int n_nets, int n_slots,
                     const gpn_net_rulebook_t* rbs, int n_rbs, int n_convs, int n_bns, int mode, void* ws, size_t ws_bytes,
                     hipStream_t stream) {
  // mode: 0 = eval (running statistics), 1 = training (batch statistics), GPN_NET_INFERENCE = eval and no backward pass follows
  const int training = mode == 1 ? 1 : 0;
  const bool inference = mode == GPN_NET_INFERENCE;
  if (mode != 0 && mode != 1 && mode != GPN_NET_INFERENCE) {
    gpn::set_error("%s: training must be 0, 1 or GPN_NET_INFERENCE", who);
    return GPN_ERR_ARG;
  }
  int rc = GPN_OK;
  for (int t = 0; t < n_nets; ++t) {
    rc = check_program(who, ops, n_ops, nets[t].slots, n_slots, rbs, n_rbs, nets[t].convs, n_convs, nets[t].bns, n_bns);
    if (rc) return rc;
  }
  const Need need = workspace_need(ops, n_ops, nets[0].slots, rbs, nets[0].convs);
  const size_t per_net = need.packed + need.stats;
  if (!ws || ws_bytes < n_nets * per_net + need.op) {
    gpn::set_error("%s: workspace too small (%zu needed, %zu given)", who, n_nets * per_net + need.op, ws_bytes);
    return GPN_ERR_WS;
  }
  std::vector<const float*> packed_of[2];
  for (int t = 0; t < n_nets; ++t) {
    rc = pack_program(ops, n_ops, rbs, nets[t].convs, false, static_cast<char*>(ws) + t * per_net, packed_of[t], stream);
    if (rc) return rc;
  }
  void* op_ws = static_cast<char*>(ws) + n_nets * per_net;
  const size_t op_ws_bytes = ws_bytes - n_nets * per_net;
  // training: a BatchNorm that directly follows a conv whose kernel has the sum epilogue gets its statistics from that
  // launch (bn_stats.h) and keeps only its apply pass
  std::vector<unsigned long long*> slab_of[2];  // by BN op: where its sums are accumulated; by CONV op: same slab
  for (int t = 0; t < n_nets; ++t) slab_of[t].assign(n_ops, nullptr);
  if (training && g_bn_fusion.load(std::memory_order_relaxed)) {
    std::vector<int> readers(n_slots, 0);
    for (int i = 0; i < n_ops; ++i) {
      readers[ops[i].src0]++;
      if (ops[i].src1 >= 0) readers[ops[i].src1]++;
    }
    for (int t = 0; t < n_nets; ++t) {
      char* area = static_cast<char*>(ws) + t * per_net + need.packed;
      size_t off = 0;
      for (int i = 0; i + 1 < n_ops; ++i) {
        const gpn_net_op_t &cv_op = ops[i], &bn_op = ops[i + 1];
        if (cv_op.kind != GPN_NET_CONV || bn_op.kind != GPN_NET_BN || bn_op.src0 != cv_op.dst || readers[cv_op.dst] != 1) continue;
        const gpn_net_rulebook_t& rb = rbs[cv_op.rulebook];
        const gpn_net_conv_t& cv = nets[t].convs[cv_op.param];
        const gpn_net_slot_t& out_slot = nets[0].slots[cv_op.dst];
        if (!gpn::bn_two_pass(plan_of(out_slot), cv.cout) ||
            !gpn::spconv_fwd_accumulates_stats(rb.K, rb.n_dst, cv.cin, cv.cout, dev_rows(out_slot)))
          continue;
        slab_of[t][i] = slab_of[t][i + 1] = reinterpret_cast<unsigned long long*>(area + off);
        off += gpn::stat_slab_bytes(cv.cout);
      }
      if (off) GPN_CHECK_HIP(hipMemsetAsync(area, 0, off, stream));
    }
  }
  const bool pair = n_nets == 2;
  // inference (running statistics, no backward pass to follow): a BatchNorm that directly follows a conv - the only reader of the
  // conv's output, the conv on a kernel with the affine epilogue - is applied by that conv's launch (gpn::ConvAffine: the same
  // arithmetic per element, the same bits) and has no launch of its own; the conv's own output slot stays unwritten.
  // folded[i] != 0: BN op i was applied by CONV op i - 1.
  std::vector<char> folded(n_ops, 0);
  if (inference && g_bn_fusion.load(std::memory_order_relaxed)) {
    std::vector<int> readers(n_slots, 0);
    for (int i = 0; i < n_ops; ++i) {
      readers[ops[i].src0]++;
      if (ops[i].src1 >= 0) readers[ops[i].src1]++;
    }
    for (int i = 0; i + 1 < n_ops; ++i) {
      const gpn_net_op_t &cv_op = ops[i], &bn_op = ops[i + 1];
      if (cv_op.kind != GPN_NET_CONV || bn_op.kind != GPN_NET_BN || bn_op.src0 != cv_op.dst || readers[cv_op.dst] != 1) continue;
      if (bn_op.src1 == cv_op.dst || bn_op.dst == cv_op.src0) continue;
      const gpn_net_rulebook_t& rb = rbs[cv_op.rulebook];
      const gpn_net_conv_t& cv = nets[0].convs[cv_op.param];
      bool ok = gpn::spconv_fwd_applies_affine(rb.K, rb.n_dst, cv.cin, cv.cout, dev_rows(nets[0].slots[cv_op.dst]));
      for (int t = 0; t < n_nets; ++t) {
        const gpn_net_bn_t& bn = nets[t].bns[bn_op.param];
        ok = ok && bn.running_mean && bn.running_var && bn.eps == nets[0].bns[bn_op.param].eps;
      }
      if (ok) folded[i + 1] = 1;
    }
  }
  for (int i = 0; i < n_ops; ++i) {
    const gpn_net_op_t& op = ops[i];
    if (folded[i]) continue;  // (applied by the conv before it)
    for (int t = 0; t < n_nets; ++t)
      if (!nets[t].slots[op.src0].data || !nets[t].slots[op.dst].data) {
        gpn::set_error("%s: op %d: null activation pointer", who, i);
        return GPN_ERR_ARG;
      }
    const gpn_net_slot_t &s0 = nets[0].slots[op.src0], &d = nets[0].slots[op.dst];
    if (op.kind == GPN_NET_CONV) {
      const gpn_net_rulebook_t& rb = rbs[op.rulebook];
      const gpn_net_conv_t& cv = nets[0].convs[op.param];
      gpn::ConvStats st;
      st.slab = slab_of[0][i];
      st.slot_mask = gpn::stat_slot_count(plan_of(d)) - 1;
      if (pair) {
        st.twin.in = nets[1].slots[op.src0].data;
        st.twin.packed = packed_of[1][i];
        st.twin.out = nets[1].slots[op.dst].data;
        st.twin.slab = slab_of[1][i];
      }
      float* out = d.data;
      if (i + 1 < n_ops && folded[i + 1]) {  // this launch also applies the BatchNorm behind it, into that BatchNorm's output slot
        const gpn_net_op_t& bo = ops[i + 1];
        auto affine = [&](int t) {
          const gpn_net_bn_t& bn = nets[t].bns[bo.param];
          gpn::ConvAffine a;
          a.mean = bn.running_mean, a.var = bn.running_var, a.weight = bn.weight, a.bias = bn.bias;
          a.res = bo.src1 >= 0 ? nets[t].slots[bo.src1].data : nullptr;
          a.eps = bn.eps, a.relu = (bo.flags & GPN_NET_RELU) ? 1 : 0;
          return a;
        };
        for (int t = 0; t < n_nets; ++t)
          if (!nets[t].slots[bo.dst].data || (bo.src1 >= 0 && !nets[t].slots[bo.src1].data)) {
            gpn::set_error("%s: op %d: null activation pointer", who, i + 1);
            return GPN_ERR_ARG;
          }
        st.ep = affine(0);
        out = nets[0].slots[bo.dst].data;
        if (pair) {
          st.twin.ep = affine(1);
          st.twin.out = nets[1].slots[bo.dst].data;
        }
      }
      rc = gpn::spconv_fwd_into(s0.data, packed_of[0][i], rb.nbr, rb.nbr_p, rb.perm, rb.K, rb.n_dst, cv.cin, cv.cout, out, 0, st,
                                op_ws, op_ws_bytes, stream, dev_rows(d));
    } else if (op.kind == GPN_NET_BN) {
      const int relu = (op.flags & GPN_NET_RELU) ? 1 : 0;
      gpn::BnFwdPtrs pp[2];
      for (int t = 0; t < n_nets; ++t) {
        const gpn_net_bn_t& bn = nets[t].bns[op.param];
        pp[t].x = nets[t].slots[op.src0].data;
        pp[t].res = op.src1 >= 0 ? nets[t].slots[op.src1].data : nullptr;
        pp[t].partial = slab_of[t][i];
        pp[t].weight = bn.weight, pp[t].bias = bn.bias, pp[t].y = nets[t].slots[op.dst].data;
        pp[t].mean = bn.save_mean, pp[t].invstd = bn.save_invstd;
        pp[t].running_mean = bn.running_mean, pp[t].running_var = bn.running_var;
      }
      const gpn_net_bn_t& bn0 = nets[0].bns[op.param];
      const bool together = pair && nets[1].bns[op.param].eps == bn0.eps && nets[1].bns[op.param].momentum == bn0.momentum;
      if (training && slab_of[0][i] && together) {
        rc = gpn::bn_fwd_train_fused(pp[0], &pp[1], s0.rows, bn0.C, bn0.eps, bn0.momentum, relu, stream, dev_rows(s0));
      } else if (!training) {  // running statistics: one launch per BatchNorm, or per pair of them
        for (int t = 0; t < n_nets; ++t)
          if (!nets[t].bns[op.param].running_mean || !nets[t].bns[op.param].running_var) {
            gpn::set_error("%s: op %d: eval-mode BatchNorm needs running statistics", who, i);
            return GPN_ERR_ARG;
          }
        if (together) {
          rc = gpn::bn_fwd_eval_running(pp[0], &pp[1], s0.rows, dev_rows(s0), bn0.C, bn0.eps, relu, stream);
        } else {
          for (int t = 0; t < n_nets && rc == GPN_OK; ++t)
            rc = gpn::bn_fwd_eval_running(pp[t], nullptr, s0.rows, dev_rows(s0), nets[t].bns[op.param].C, nets[t].bns[op.param].eps, relu, stream);
        }
      } else {
        for (int t = 0; t < n_nets && rc == GPN_OK; ++t) {
          const gpn_net_bn_t& bn = nets[t].bns[op.param];
          if (training && slab_of[t][i]) {
            rc = gpn::bn_fwd_train_fused(pp[t], nullptr, s0.rows, bn.C, bn.eps, bn.momentum, relu, stream, dev_rows(s0));
          } else {
            rc = gpn::bn_fwd_train_rows(pp[t].x, pp[t].res, bn.weight, bn.bias, s0.rows, dev_rows(s0), bn.C, bn.eps, bn.momentum, relu,
                                        pp[t].y, bn.save_mean, bn.save_invstd, bn.running_mean, bn.running_var, op_ws, op_ws_bytes, stream);
          }
        }
      }
    } else {
      const gpn_net_slot_t& s1 = nets[0].slots[op.src1];
      ConcatPtrs ca{(const float4*)s0.data, (const float4*)s1.data, (float4*)d.data}, cb = ca;
      if (pair)
        cb = ConcatPtrs{(const float4*)nets[1].slots[op.src0].data, (const float4*)nets[1].slots[op.src1].data,
                        (float4*)nets[1].slots[op.dst].data};
      hipLaunchKernelGGL(concat_kernel, dim3(grid_for(plan_of(d) * (d.channels / 4)), n_nets), dim3(kThreads), 0, stream, ca, cb,
                         d.rows, s0.channels / 4, s1.channels / 4, d.rows_dev);
      GPN_CHECK_LAUNCH();
      rc = GPN_OK;
    }
    if (rc) return rc;
  }
  return GPN_OK;
}

}  // namespace

extern "C" int gpn_net_forward(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots, int n_slots,
                               const gpn_net_rulebook_t* rbs, int n_rbs, const gpn_net_conv_t* convs, int n_convs,
                               const gpn_net_bn_t* bns, int n_bns, int training, void* ws, size_t ws_bytes,
                               gpn_stream_t stream_) {
  const NetSet one{slots, convs, bns};
  return net_forward_impl(__func__, ops, n_ops, &one, 1, n_slots, rbs, n_rbs, n_convs, n_bns, training, ws, ws_bytes,
                          (hipStream_t)stream_);
}

// two structurally identical networks over the same rulebooks in one pass (see NetSet); workspace: 2 x gpn_net_ws_bytes
extern "C" int gpn_net_forward_pair(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots_a, gpn_net_slot_t* slots_b,
                                    int n_slots, const gpn_net_rulebook_t* rbs, int n_rbs, const gpn_net_conv_t* convs_a,
                                    const gpn_net_conv_t* convs_b, int n_convs, const gpn_net_bn_t* bns_a,
                                    const gpn_net_bn_t* bns_b, int n_bns, int training, void* ws, size_t ws_bytes,
                                    gpn_stream_t stream_) {
  const NetSet two[2] = {{slots_a, convs_a, bns_a}, {slots_b, convs_b, bns_b}};
  return net_forward_impl(__func__, ops, n_ops, two, 2, n_slots, rbs, n_rbs, n_convs, n_bns, training, ws, ws_bytes,
                          (hipStream_t)stream_);
}

namespace {

// The weight-gradient launches of a backward pass are issued by a helper thread: they go to the second stream, nothing
// on the main chain waits for them, and a HIP launch costs the issuing thread ~2 us - taking them (and their stream
// waits) off the thread that issues the dgrad / BatchNorm chain shortens the host side of the pass by about a third.
// Protocol: the caller publishes jobs (plain structs) through an atomic counter while the pass runs; the worker spins
// on the counter during a pass and sleeps on a condition variable between passes.
// layers per weight-gradient contraction launch (1 ... gpn::kWgradSets; gpn_net_wgrad_group()).  1 = only
// the two networks of a paired pass share a launch (what every measurement of round 3 ran: the knob was unreadable, see the
// note at env_bn_fusion).  First measurement, round 4 (tools/wgrad_group_ab.sh, profiles/r04_wgrad_group_ab.txt): the grouped
// path is bit-equal (48 executor / paired-pass / golden tests at 4), 90 -> 67 launches per step on the weight-gradient stream,
// step time 8.79 / 8.78 / 8.72 ms mean of three interleaved runs at 1 / 2 / 4 - inside the noise, fewer launches for the helper
// thread to issue: default 4.
std::atomic<int> g_wgrad_group{4};
std::atomic<int>& wgrad_group_setting() { return g_wgrad_group; }

constexpr int64_t g_wgrad_group_rows = 16384;  // only layers with fewer rows than this share a launch

struct WgradJob {
  const float* in;
  const float* dout;
  const int32_t *pair_src, *pair_dst, *tile_off;
  int K;
  int64_t n_dst;
  gpn::DevRows rows;  // n_dst is a bound and the live count of destination rows a device counter (or {nullptr, 0})
  int cin, cout;
  float* dW;
  hipEvent_t after;  // recorded on the caller's stream once dout is final
  // the same layer of the second network of a paired pass (one launch contracts both), or nulls
  const float* in2;
  const float* dout2;
  float* dW2;
};

class WgradWorker {
 public:
  // start a pass: jobs will be issued to `side` of device `dev` with workspace `ws`
  void begin(int dev, hipStream_t side, void* ws, size_t ws_bytes, size_t max_jobs) {
    ensure_thread();
    jobs_.resize(max_jobs);
    dev_ = dev, side_ = side, ws_ = ws, ws_bytes_ = ws_bytes;
    published_.store(0, std::memory_order_relaxed);
    closed_.store(false, std::memory_order_relaxed);
    finished_.store(false, std::memory_order_relaxed);
    rc_ = GPN_OK;
    {
      std::lock_guard<std::mutex> lk(mu_);
      active_ = true;
    }
    cv_.notify_one();
  }
  void push(const WgradJob& j) {
    const size_t i = published_.load(std::memory_order_relaxed);
    jobs_[i] = j;
    published_.store(i + 1, std::memory_order_release);
  }
  // end of pass: wait until every published job has been issued; returns the first error (message in `err`)
  int finish(std::string& err) {
    closed_.store(true, std::memory_order_release);
    while (!finished_.load(std::memory_order_acquire)) std::this_thread::yield();
    err = err_;
    return rc_;
  }

 private:
  void ensure_thread() {
    const pid_t pid = getpid();
    if (started_ && pid_ == pid) return;
    started_ = true;  // (after a fork the child starts its own worker; the parent's thread does not exist there)
    pid_ = pid;
    std::thread([this] { run(); }).detach();
  }
  void run() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return active_; });
        active_ = false;
      }
      (void)hipSetDevice(dev_);
      size_t done = 0;
      // slice sums are deferred: every contraction gets its own piece of the workspace, and the sums of a batch of layers
      // (up to kWgradReduceJobs of them / kWgradBatchBytes of partials) run as ONE launch (gpn::wgrad_reduce_many)
      gpn::WgradReduceJob pending[gpn::kWgradReduceJobs];
      int n_pending = 0;
      size_t used = 0;
      const size_t room = std::min(ws_bytes_, kWgradBatchBytes);
      auto fail = [this](int rc) {
        if (rc_ == GPN_OK) {
          rc_ = rc;
          err_ = gpn_last_error();
        }
      };
      auto flush = [&] {
        if (n_pending && rc_ == GPN_OK) {
          const int rc = gpn::wgrad_reduce_many(pending, n_pending, side_);
          if (rc != GPN_OK) fail(rc);
        }
        n_pending = 0;
        used = 0;
      };
      // consecutive jobs of one shape (the convs of a level's residual blocks; the two networks of a paired pass) are
      // contracted by ONE launch of up to kWgradSets layers: held back until a job of another shape arrives or the pass ends
      // (on the deep levels a contraction is a 15-40 us latency chain over a few hundred workgroups: together they overlap)
      WgradJob group[gpn::kWgradSets];
      int n_group = 0, group_sets = 0;
      auto same_shape = [](const WgradJob& a, const WgradJob& b) {
        return a.K == b.K && a.n_dst == b.n_dst && a.cin == b.cin && a.cout == b.cout && a.rows.dev == b.rows.dev;
      };
      auto launch_group = [&]() -> int {
        if (!n_group) return GPN_OK;
        const WgradJob& j0 = group[0];
        const int n = n_group, n_sets = group_sets;
        n_group = 0, group_sets = 0;
        // the events were recorded in job order on one stream: the last one covers the others
        if (hipStreamWaitEvent(side_, group[n - 1].after, 0) != hipSuccess) {
          gpn::set_error("gpn_net_backward: hipStreamWaitEvent failed on the weight-gradient stream");
          return GPN_ERR_HIP;
        }
        const size_t elems = (size_t)j0.K * j0.cin * j0.cout;
        if (j0.n_dst == 0) {
          for (int g = 0; g < n; ++g) {
            GPN_CHECK_HIP(hipMemsetAsync(group[g].dW, 0, sizeof(float) * elems, side_));
            if (group[g].dW2) GPN_CHECK_HIP(hipMemsetAsync(group[g].dW2, 0, sizeof(float) * elems, side_));
          }
          return GPN_OK;
        }
        const int S = gpn::wgrad_slices(j0.K, j0.cin, j0.cout, gpn::plan_rows(j0.n_dst, j0.rows));
        const size_t bytes = gpn::align_up((size_t)S * elems * sizeof(float));
        if (used + n_sets * bytes > room || n_pending + n_sets > gpn::kWgradReduceJobs) flush();
        if (n_sets * bytes > ws_bytes_ || !ws_) {
          gpn::set_error("gpn_net_backward: weight-gradient workspace too small");
          return GPN_ERR_WS;
        }
        gpn::WgradSets sets;
        float* dW_of[gpn::kWgradSets];
        for (int g = 0; g < n; ++g) {
          const WgradJob& j = group[g];
          float* partial = reinterpret_cast<float*>(static_cast<char*>(ws_) + used + sets.n * bytes);
          dW_of[sets.n] = j.dW;
          sets.s[sets.n++] = gpn::WgradSet{j.in, j.dout, j.pair_src, j.pair_dst, j.tile_off, partial};
          if (j.dW2) {
            partial = reinterpret_cast<float*>(static_cast<char*>(ws_) + used + sets.n * bytes);
            dW_of[sets.n] = j.dW2;
            sets.s[sets.n++] = gpn::WgradSet{j.in2, j.dout2, j.pair_src, j.pair_dst, j.tile_off, partial};
          }
        }
        const int rc = gpn::wgrad_contract(sets, j0.K, j0.n_dst, j0.cin, j0.cout, S, side_, j0.rows.dev);
        if (rc != GPN_OK) return rc;
        for (int q = 0; q < sets.n; ++q)
          pending[n_pending++] = gpn::wgrad_reduce_job(sets.s[q].partial, S, j0.K, j0.cin, j0.cout, GPN_LAYOUT_OKI, dW_of[q]);
        used += sets.n * bytes;
        return GPN_OK;
      };
      auto issue = [&](const WgradJob& j) -> int {
        const int n_sets = j.dW2 ? 2 : 1;
        // (a layer with many rows fills the chip on its own: holding it back only delays it)
        const int limit = gpn::plan_rows(j.n_dst, j.rows) < g_wgrad_group_rows ? g_wgrad_group.load(std::memory_order_relaxed) : 1;
        if (n_group && (!same_shape(group[0], j) || group_sets + n_sets > limit)) {
          const int rc = launch_group();
          if (rc != GPN_OK) return rc;
        }
        group[n_group++] = j;
        group_sets += n_sets;
        if (group_sets >= limit) return launch_group();
        return GPN_OK;
      };
      for (;;) {
        const size_t avail = published_.load(std::memory_order_acquire);
        if (done < avail) {
          const WgradJob& j = jobs_[done++];
          if (rc_ != GPN_OK) continue;  // drain after an error
          const int rc = issue(j);
          if (rc != GPN_OK) fail(rc);
        } else if (closed_.load(std::memory_order_acquire) && done == published_.load(std::memory_order_acquire)) {
          break;
        } else {
          std::this_thread::yield();  // nothing published yet: give the core away (8 ranks share the host's cores)
        }
      }
      if (rc_ == GPN_OK) {
        const int rc = launch_group();
        if (rc != GPN_OK) fail(rc);
      }
      flush();
      finished_.store(true, std::memory_order_release);
    }
  }

  std::mutex mu_;
  std::condition_variable cv_;
  bool active_ = false, started_ = false;
  pid_t pid_ = 0;
  std::vector<WgradJob> jobs_;
  std::atomic<size_t> published_{0};
  std::atomic<bool> closed_{false}, finished_{false};
  int dev_ = 0;
  hipStream_t side_ = nullptr;
  void* ws_ = nullptr;
  size_t ws_bytes_ = 0;
  int rc_ = GPN_OK;
  std::string err_;
};

// one worker (thread + pass lock) per device: passes on different devices of one process do not serialise, and nothing
// here is keyed by "the" device of the process
struct DeviceWorker {
  std::mutex pass_mu;  // one backward pass at a time per device (a worker serves a single pass)
  WgradWorker worker;
};

DeviceWorker& device_worker(int dev) {
  static std::mutex mu;
  static DeviceWorker* per_device[gpn::kMaxDevices] = {};  // leaked on purpose: detached threads may outlive static destructors
  std::lock_guard<std::mutex> lk(mu);
  if (!per_device[dev]) per_device[dev] = new DeviceWorker();
  return *per_device[dev];
}

// hand a gradient buffer to a producer: the slot's own buffer if nothing was written there yet, else the staging buffer
struct GradTarget {
  float* ptr;
  bool staged;
};

inline GradTarget grad_target(gpn_net_slot_t& s, float* tmp) {
  return s.grad_state ? GradTarget{tmp, true} : GradTarget{s.grad, false};
}

int commit(gpn_net_slot_t& s, const GradTarget& t, hipStream_t stream) {
  if (t.staged) {
    const int64_t total4 = s.rows * s.channels / 4;
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid_for(plan_of(s) * s.channels / 4)), dim3(kThreads), 0, stream, (float4*)s.grad,
                       (const float4*)t.ptr, total4, s.channels / 4, s.rows_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      gpn::set_error("gpn_net_backward: accumulate launch failed: %s", hipGetErrorString(e));
      return GPN_ERR_HIP;
    }
  }
  s.grad_state = 1;
  return GPN_OK;
}

}  // namespace

namespace {

int net_backward_impl(const char* who, const gpn_net_op_t* ops, int n_ops, const NetSet* nets, int n_nets, int n_slots,
                      const gpn_net_rulebook_t* rbs, int n_rbs, int n_convs, int n_bns, int training, int need_input_grad,
                      void* ws, size_t ws_bytes, hipStream_t stream) {
  int rc = GPN_OK;
  for (int t = 0; t < n_nets; ++t) {
    rc = check_program(who, ops, n_ops, nets[t].slots, n_slots, rbs, n_rbs, nets[t].convs, n_convs, nets[t].bns, n_bns);
    if (rc) return rc;
  }
  const Need need = workspace_need(ops, n_ops, nets[0].slots, rbs, nets[0].convs);
  const size_t per_net = need.tmp + need.packed + need.stats;
  const size_t total_need = n_nets * per_net + need.op + n_nets * need.wgrad;
  if (!ws || ws_bytes < total_need) {
    gpn::set_error("%s: workspace too small (%zu needed, %zu given)", who, total_need, ws_bytes);
    return GPN_ERR_WS;
  }
  char* base = static_cast<char*>(ws);
  float* tmp[2] = {reinterpret_cast<float*>(base), reinterpret_cast<float*>(base + per_net)};
  void* op_ws = base + n_nets * per_net;
  const size_t op_ws_bytes = need.op;
  void* wgrad_ws = base + n_nets * per_net + need.op;
  const size_t wgrad_ws_bytes = ws_bytes - (n_nets * per_net + need.op);
  std::vector<const float*> packed_of[2];
  for (int t = 0; t < n_nets; ++t) {
    rc = pack_program(ops, n_ops, rbs, nets[t].convs, true, base + t * per_net + need.tmp, packed_of[t], stream);
    if (rc) return rc;
  }
  // A dgrad conv that writes the FINAL gradient of a BatchNorm's output (it is the slot's first reader in program order, so
  // the last one of this reverse walk) adds that BatchNorm's backward sums in its epilogue (bn_stats.h); the BatchNorm then
  // runs only its apply pass.  bn_slab[t][j]: the slab of BN op j of network t; fused_by[i]: the BN op CONV op i accumulates for.
  std::vector<unsigned long long*> bn_slab[2];
  for (int t = 0; t < n_nets; ++t) bn_slab[t].assign(n_ops, nullptr);
  std::vector<int> fused_by(n_ops, -1);
  std::vector<char> sums_done(n_ops, 0);
  if (g_bn_fusion.load(std::memory_order_relaxed)) {
    std::vector<int> first_reader(n_slots, -1), producer(n_slots, -1);
    for (int i = n_ops - 1; i >= 0; --i) {
      first_reader[ops[i].src0] = i;
      if (ops[i].src1 >= 0) first_reader[ops[i].src1] = i;
      producer[ops[i].dst] = i;
    }
    size_t off = 0;
    for (int i = 0; i < n_ops; ++i) {
      if (ops[i].kind != GPN_NET_CONV) continue;
      const int j = producer[ops[i].src0];
      if (j < 0 || ops[j].kind != GPN_NET_BN || first_reader[ops[i].src0] != i) continue;
      if (ops[i].src0 == 0 && !need_input_grad) continue;
      const gpn_net_rulebook_t& rb = rbs[ops[i].rulebook];
      const gpn_net_conv_t& cv = nets[0].convs[ops[i].param];
      const gpn_net_slot_t& in_slot = nets[0].slots[ops[i].src0];
      if (!gpn::bn_two_pass(plan_of(in_slot), cv.cin) ||
          !gpn::spconv_fwd_accumulates_stats(rb.K, rb.n_src, cv.cout, cv.cin, dev_rows(in_slot)))
        continue;
      for (int t = 0; t < n_nets; ++t)
        bn_slab[t][j] = reinterpret_cast<unsigned long long*>(base + t * per_net + need.tmp + need.packed + off);
      off += gpn::stat_slab_bytes(cv.cin);
      fused_by[i] = j;
    }
    if (off)
      for (int t = 0; t < n_nets; ++t) GPN_CHECK_HIP(hipMemsetAsync(base + t * per_net + need.tmp + need.packed, 0, off, stream));
  }
  SideStream* side = side_stream();
  if (!side) {
    gpn::set_error("%s: could not create the weight-gradient stream", who);
    return GPN_ERR_HIP;
  }
  // the side stream may still be reading the workspace of the previous call on this device
  GPN_CHECK_HIP(hipEventRecord(side->join, side->stream));
  GPN_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));
  bool forked = false;
  int device = 0;
  GPN_CHECK_HIP(hipGetDevice(&device));
  GPN_CHECK_ARG(device >= 0 && device < gpn::kMaxDevices);
  DeviceWorker& dw = device_worker(device);
  std::lock_guard<std::mutex> pass_lock(dw.pass_mu);
  WgradWorker& worker = dw.worker;
  worker.begin(device, side->stream, wgrad_ws, wgrad_ws_bytes, (size_t)n_ops * n_nets);
  // every exit below must close the pass, or the worker would spin forever
  struct PassGuard {
    WgradWorker& w;
    bool closed = false;
    ~PassGuard() {
      if (!closed) {
        std::string ignored;
        (void)w.finish(ignored);
      }
    }
  } guard{worker};
  const bool pair = n_nets == 2;
  for (int i = n_ops - 1; i >= 0; --i) {
    const gpn_net_op_t& op = ops[i];
    gpn_net_slot_t &s0 = nets[0].slots[op.src0], &d = nets[0].slots[op.dst];
    if (!d.grad_state) continue;  // nothing flows back through this op (its output does not reach the loss)
    for (int t = 0; t < n_nets; ++t) {
      const gpn_net_slot_t &ts0 = nets[t].slots[op.src0], &td = nets[t].slots[op.dst];
      if (!td.grad || !ts0.data || !td.data || td.grad_state != d.grad_state) {
        gpn::set_error("%s: op %d: null pointer (or the two networks' gradient states differ)", who, i);
        return GPN_ERR_ARG;
      }
    }
    if (op.kind == GPN_NET_CONV) {
      const gpn_net_rulebook_t& rb = rbs[op.rulebook];
      const gpn_net_conv_t& cv = nets[0].convs[op.param];
      bool any_dw = false;
      for (int t = 0; t < n_nets; ++t) any_dw = any_dw || nets[t].convs[op.param].dW;
      if (any_dw) {
        if (!rb.pair_src || !rb.pair_dst || !rb.tile_off) {
          gpn::set_error("%s: op %d: rulebook has no pair lists for wgrad", who, i);
          return GPN_ERR_ARG;
        }
        // d.grad is final here (every consumer of slot d ran earlier in this reverse walk): fork the contraction
        hipEvent_t ev = side->fork[side->next];
        side->next = (side->next + 1) % kForkEvents;
        GPN_CHECK_HIP(hipEventRecord(ev, stream));
        if (pair && nets[0].convs[op.param].dW && nets[1].convs[op.param].dW) {
          worker.push(WgradJob{nets[0].slots[op.src0].data, nets[0].slots[op.dst].grad, rb.pair_src, rb.pair_dst, rb.tile_off,
                               rb.K, rb.n_dst, dev_rows(d), cv.cin, cv.cout, nets[0].convs[op.param].dW, ev, nets[1].slots[op.src0].data,
                               nets[1].slots[op.dst].grad, nets[1].convs[op.param].dW});
        } else {
          for (int t = 0; t < n_nets; ++t)
            if (nets[t].convs[op.param].dW)
              worker.push(WgradJob{nets[t].slots[op.src0].data, nets[t].slots[op.dst].grad, rb.pair_src, rb.pair_dst, rb.tile_off,
                                   rb.K, rb.n_dst, dev_rows(d), cv.cin, cv.cout, nets[t].convs[op.param].dW, ev, nullptr, nullptr, nullptr});
        }
        forked = true;
      }
      if (op.src0 != 0 || need_input_grad) {
        for (int t = 0; t < n_nets; ++t)
          if (!nets[t].slots[op.src0].grad || !rb.nbr_t) {
            gpn::set_error("%s: op %d: null gradient buffer / transposed table", who, i);
            return GPN_ERR_ARG;
          }
        // a slot that already holds a gradient takes this one added in place (no staging buffer, no accumulate launch)
        gpn::ConvStats st;
        if (pair) {
          st.twin.in = nets[1].slots[op.dst].grad;
          st.twin.packed = packed_of[1][i];
          st.twin.out = nets[1].slots[op.src0].grad;
        }
        if (fused_by[i] >= 0) {
          const gpn_net_op_t& bo = ops[fused_by[i]];
          const gpn_net_bn_t& bn = nets[0].bns[bo.param];
          st.slab = bn_slab[0][fused_by[i]];
          st.slot_mask = gpn::stat_slot_count(plan_of(s0)) - 1;
          st.x = nets[0].slots[bo.src0].data;
          st.y = s0.data;
          st.mean = training ? bn.save_mean : bn.running_mean;
          st.invstd = bn.save_invstd;
          st.relu = (bo.flags & GPN_NET_RELU) ? 1 : 0;
          if (pair) {
            const gpn_net_bn_t& bn1 = nets[1].bns[bo.param];
            st.twin.slab = bn_slab[1][fused_by[i]];
            st.twin.x = nets[1].slots[bo.src0].data;
            st.twin.y = nets[1].slots[op.src0].data;
            st.twin.mean = training ? bn1.save_mean : bn1.running_mean;
            st.twin.invstd = bn1.save_invstd;
          }
          sums_done[fused_by[i]] = 1;
        }
        rc = gpn::spconv_fwd_into(d.grad, packed_of[0][i], rb.nbr_t, rb.nbr_t_p, rb.perm_t, rb.K, rb.n_src, cv.cout, cv.cin, s0.grad,
                                  s0.grad_state ? 1 : 0, st, op_ws, op_ws_bytes, stream, dev_rows(s0));
        if (rc) return rc;
        for (int t = 0; t < n_nets; ++t) nets[t].slots[op.src0].grad_state = 1;
      }
    } else if (op.kind == GPN_NET_BN) {
      const int relu = (op.flags & GPN_NET_RELU) ? 1 : 0;
      gpn::BnBwdPtrs pp[2];
      GradTarget tr[2] = {{nullptr, false}, {nullptr, false}};
      for (int t = 0; t < n_nets; ++t) {
        const gpn_net_bn_t& bn = nets[t].bns[op.param];
        gpn_net_slot_t& ts0 = nets[t].slots[op.src0];
        if (!ts0.grad || !bn.dweight || !bn.dbias) {
          gpn::set_error("%s: op %d: null gradient buffer", who, i);
          return GPN_ERR_ARG;
        }
        GradTarget tx = grad_target(ts0, tmp[t]);
        if (tx.staged) {
          gpn::set_error("%s: op %d: BatchNorm input consumed twice is not supported", who, i);
          return GPN_ERR_ARG;
        }
        if (op.src1 >= 0 && (op.src1 != 0 || need_input_grad)) {
          if (!nets[t].slots[op.src1].grad) {
            gpn::set_error("%s: op %d: null residual gradient buffer", who, i);
            return GPN_ERR_ARG;
          }
          tr[t] = grad_target(nets[t].slots[op.src1], tmp[t]);
        }
        pp[t].x = ts0.data, pp[t].y = nets[t].slots[op.dst].data, pp[t].dy = nets[t].slots[op.dst].grad;
        pp[t].partial = bn_slab[t][i];
        pp[t].mean = training ? bn.save_mean : bn.running_mean;
        pp[t].invstd = bn.save_invstd, pp[t].weight = bn.weight;
        pp[t].dx = tx.ptr, pp[t].dres = tr[t].ptr, pp[t].dweight = bn.dweight, pp[t].dbias = bn.dbias;
      }
      const gpn_net_bn_t& bn0 = nets[0].bns[op.param];
      if (sums_done[i] && pair) {
        rc = gpn::bn_bwd_fused(pp[0], &pp[1], s0.rows, bn0.C, relu, training ? 1 : 0, stream, dev_rows(s0));
      } else {
        for (int t = 0; t < n_nets && rc == GPN_OK; ++t) {
          if (sums_done[i])
            rc = gpn::bn_bwd_fused(pp[t], nullptr, s0.rows, bn0.C, relu, training ? 1 : 0, stream, dev_rows(s0));
          else
            rc = gpn::bn_bwd_rows(pp[t].x, pp[t].y, pp[t].dy, pp[t].weight, pp[t].mean, pp[t].invstd, s0.rows, dev_rows(s0), bn0.C, relu,
                                  training ? 1 : 0, pp[t].dx, pp[t].dres, pp[t].dweight, pp[t].dbias, op_ws, op_ws_bytes, stream);
        }
      }
      if (rc) return rc;
      for (int t = 0; t < n_nets; ++t) {
        nets[t].slots[op.src0].grad_state = 1;
        if (pp[t].dres) {
          rc = commit(nets[t].slots[op.src1], tr[t], stream);
          if (rc) return rc;
        }
      }
    } else {
      gpn_net_slot_t& s1 = nets[0].slots[op.src1];
      for (int t = 0; t < n_nets; ++t)
        if (!nets[t].slots[op.src0].grad || !nets[t].slots[op.src1].grad) {
          gpn::set_error("%s: op %d: null gradient buffer", who, i);
          return GPN_ERR_ARG;
        }
      SplitPtrs sa{(const float4*)d.grad, (float4*)s0.grad, (float4*)s1.grad}, sb = sa;
      if (pair)
        sb = SplitPtrs{(const float4*)nets[1].slots[op.dst].grad, (float4*)nets[1].slots[op.src0].grad,
                       (float4*)nets[1].slots[op.src1].grad};
      hipLaunchKernelGGL(split_kernel, dim3(grid_for(plan_of(d) * (d.channels / 4)), n_nets), dim3(kThreads), 0, stream, sa, sb, d.rows,
                         s0.channels / 4, s1.channels / 4, s0.grad_state, s1.grad_state, d.rows_dev);
      GPN_CHECK_LAUNCH();
      for (int t = 0; t < n_nets; ++t) nets[t].slots[op.src0].grad_state = nets[t].slots[op.src1].grad_state = 1;
    }
  }
  std::string worker_err;
  guard.closed = true;
  rc = worker.finish(worker_err);
  if (rc) {
    gpn::set_error("%s", worker_err.c_str());
    return rc;
  }
  if (forked) {  // join: whatever the caller enqueues next (optimizer, the next pass) sees every dW
    GPN_CHECK_HIP(hipEventRecord(side->join, side->stream));
    GPN_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));
  }
  return GPN_OK;
}

}  // namespace

extern "C" int gpn_net_backward(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots, int n_slots,
                                const gpn_net_rulebook_t* rbs, int n_rbs, const gpn_net_conv_t* convs, int n_convs,
                                const gpn_net_bn_t* bns, int n_bns, int training, int need_input_grad, void* ws,
                                size_t ws_bytes, gpn_stream_t stream_) {
  const NetSet one{slots, convs, bns};
  return net_backward_impl(__func__, ops, n_ops, &one, 1, n_slots, rbs, n_rbs, n_convs, n_bns, training, need_input_grad, ws,
                           ws_bytes, (hipStream_t)stream_);
}

extern "C" int gpn_net_backward_pair(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots_a, gpn_net_slot_t* slots_b,
                                     int n_slots, const gpn_net_rulebook_t* rbs, int n_rbs, const gpn_net_conv_t* convs_a,
                                     const gpn_net_conv_t* convs_b, int n_convs, const gpn_net_bn_t* bns_a,
                                     const gpn_net_bn_t* bns_b, int n_bns, int training, int need_input_grad, void* ws,
                                     size_t ws_bytes, gpn_stream_t stream_) {
  const NetSet two[2] = {{slots_a, convs_a, bns_a}, {slots_b, convs_b, bns_b}};
  return net_backward_impl(__func__, ops, n_ops, two, 2, n_slots, rbs, n_rbs, n_convs, n_bns, training, need_input_grad, ws,
                           ws_bytes, (hipStream_t)stream_);
}
