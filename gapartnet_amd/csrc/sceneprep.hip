// sceneprep.hip — the per-scene preparation of RAW scenes for a whole batch in three launches (include/gpn.h section SP).
//
// Reference: GAPartNetDataset.__getitem__ (dataset/gapartnet.py:55-82) prepares every scene in a CPU loader worker:
//   compact_instance_labels (:134-142)  the non-negative instance ids of a scene renumbered 0..K-1 in ascending order;
//   apply_augmentations     (:85-120)   xyz @ M (numpy: float32 @ float64 -> float64, stored back as float32), colour + shift;
//   generate_inst_info      (:145-176)  per point mean | min | max xyz of its instance, per instance point count and the
//                                       semantic label of its first point (a Python loop over the instances).
// The repo's per-batch device pipeline (dataset/device_pipeline.py) runs the same steps for a batch in ~80 torch launches with four
// host reads (unique, nonzero, two counts) on the training thread; this file is that pipeline as one call:
//   sp_ids_kernel    one workgroup per scene: the distinct non-negative ids through an LDS hash set, ranked (K <= 256 of them:
//                    rank = number of smaller ids), written as the scene's ascending id list; resets the scene's statistics slots;
//   sp_points_kernel 16 workgroups per scene: batch index, compact id (binary search in the list), augmented point; the instance
//                    statistics of the workgroup's points in LDS (integer atomics: count, first row, min / max as ordered
//                    integers, the coordinate sums in 2^-32 fixed point - exact for |x| >= 2^-9, 2^-33 otherwise - so the means do
//                    not depend on the order of the atomics), flushed to the scene's slots with one global atomic per used field;
//   sp_finish_kernel per point the 9 region floats of its instance; per (scene, instance) the padded count / label tables.
// The number of instances per scene goes back through a pinned word per scene (the caller waits for its own event): the one
// host read of the preparation.  A scene with more than kMaxInstances distinct ids raises the overflow word; the caller then
// takes the torch formulation.
#include "gpn_common.h"

namespace {

constexpr int kMaxInstances = 256;  // distinct instance ids per scene held in LDS (GAPartNet objects have tens of parts)
constexpr int kHash = 2048;         // LDS hash set slots of sp_ids_kernel
constexpr int kSplit = 16;          // workgroups per scene in sp_points_kernel
constexpr int kIdsThreads = 1024, kThreads = 256;
constexpr double kFix = 4294967296.0;  // 2^32

__device__ __forceinline__ uint32_t order_bits(float v) {  // monotone float -> unsigned
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float order_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// statistics of one (scene, instance) slot
struct SlotStats {
  unsigned long long sum[3];  // fixed point, two's complement
  uint32_t lo[3], hi[3];      // order_bits
  uint32_t count;
  int32_t first;              // smallest point row (batch-wide)
};

__global__ __launch_bounds__(kIdsThreads) void sp_ids_kernel(const int32_t* __restrict__ ins, const int64_t* __restrict__ seg,
                                                             int32_t* __restrict__ id_list, int64_t* __restrict__ k_dev,
                                                             int32_t* __restrict__ overflow, SlotStats* __restrict__ stats,
                                                             int64_t* __restrict__ k_host) {
  __shared__ int32_t table[kHash];
  __shared__ int32_t found[kMaxInstances];
  __shared__ int32_t n_found;
  const int s = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < kHash; i += kIdsThreads) table[i] = -1;
  if (tid == 0) n_found = 0;
  for (int i = tid; i < kMaxInstances; i += kIdsThreads) {
    SlotStats z;
    z.sum[0] = z.sum[1] = z.sum[2] = 0ull;
    z.lo[0] = z.lo[1] = z.lo[2] = 0xffffffffu;
    z.hi[0] = z.hi[1] = z.hi[2] = 0u;
    z.count = 0u, z.first = 0x7fffffff;
    stats[(int64_t)s * kMaxInstances + i] = z;
  }
  __syncthreads();
  const int64_t a = seg[s], b = seg[s + 1];
  int32_t last = -1;
  for (int64_t p = a + tid; p < b; p += kIdsThreads) {
    const int32_t id = ins[p];
    if (id < 0 || id == last) continue;
    last = id;
    uint32_t h = ((uint32_t)id * 2654435761u) >> 21;  // 11 bits
    for (int probe = 0; probe < kHash; ++probe) {
      const int32_t seen = atomicCAS(&table[h], -1, id);
      if (seen == -1) {  // this thread entered the id
        const int slot = atomicAdd(&n_found, 1);
        if (slot < kMaxInstances) found[slot] = id;
        break;
      }
      if (seen == id) break;
      h = (h + 1) & (kHash - 1);
    }
  }
  __syncthreads();
  const int n = n_found;
  if (n > kMaxInstances) {  // (also covers a full hash set: kHash > kMaxInstances entries entered before it fills)
    if (tid == 0) {
      atomicExch(overflow, 1);
      k_dev[s] = 0;
      if (k_host) k_host[s] = -1;
    }
    return;
  }
  if (tid < n) {
    const int32_t mine = found[tid];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += found[j] < mine ? 1 : 0;
    id_list[(int64_t)s * kMaxInstances + rank] = mine;
  }
  if (tid == 0) {
    k_dev[s] = n;
    if (k_host) k_host[s] = n;
  }
}

__global__ __launch_bounds__(kThreads) void sp_points_kernel(const float* __restrict__ points, const int32_t* __restrict__ ins,
                                                             const int64_t* __restrict__ seg, int C, const double* __restrict__ mats,
                                                             const double* __restrict__ shifts, const int32_t* __restrict__ id_list,
                                                             const int64_t* __restrict__ k_dev, float* __restrict__ points_out,
                                                             int32_t* __restrict__ batch_indices, int32_t* __restrict__ ins_out,
                                                             SlotStats* __restrict__ stats) {
  __shared__ int32_t ids[kMaxInstances];
  __shared__ unsigned long long s_sum[kMaxInstances * 3];
  __shared__ uint32_t s_lo[kMaxInstances * 3], s_hi[kMaxInstances * 3], s_count[kMaxInstances];
  __shared__ int32_t s_first[kMaxInstances];
  __shared__ double s_m[9];
  const int s = blockIdx.y, tid = threadIdx.x;
  const int n = (int)k_dev[s];
  for (int i = tid; i < n; i += kThreads) {
    ids[i] = id_list[(int64_t)s * kMaxInstances + i];
    s_sum[3 * i] = s_sum[3 * i + 1] = s_sum[3 * i + 2] = 0ull;
    s_lo[3 * i] = s_lo[3 * i + 1] = s_lo[3 * i + 2] = 0xffffffffu;
    s_hi[3 * i] = s_hi[3 * i + 1] = s_hi[3 * i + 2] = 0u;
    s_count[i] = 0u, s_first[i] = 0x7fffffff;
  }
  if (tid < 9) s_m[tid] = mats ? mats[s * 9 + tid] : (tid % 4 == 0 ? 1.0 : 0.0);
  __syncthreads();
  const int64_t a = seg[s], b = seg[s + 1];
  const int64_t per = (b - a + kSplit - 1) / kSplit;
  const int64_t p0 = a + per * blockIdx.x, p1 = p0 + per < b ? p0 + per : b;
  const int W = 3 + C;
  for (int64_t p = p0 + tid; p < p1; p += kThreads) {
    const float* __restrict__ src = points + p * W;
    float* __restrict__ dst = points_out + p * W;
    float q[3];
    if (mats) {  // xyz @ M in float64, rounded once (numpy: float32 @ float64)
      const double x = src[0], y = src[1], z = src[2];
#pragma unroll
      for (int j = 0; j < 3; ++j) q[j] = (float)((x * s_m[j] + y * s_m[3 + j]) + z * s_m[6 + j]);
    } else {
      q[0] = src[0], q[1] = src[1], q[2] = src[2];
    }
    dst[0] = q[0], dst[1] = q[1], dst[2] = q[2];
    for (int c = 0; c < C; ++c) dst[3 + c] = shifts ? (float)((double)src[3 + c] + shifts[s * C + c]) : src[3 + c];
    batch_indices[p] = s;
    const int32_t id = ins[p];
    int32_t slot = id;
    if (id >= 0) {
      int lo = 0, hi = n - 1;  // (present by construction)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ids[mid] < id) lo = mid + 1;
        else hi = mid;
      }
      slot = lo;
      atomicAdd(&s_count[slot], 1u);
      atomicMin(&s_first[slot], (int32_t)p);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const uint32_t k = order_bits(q[j]);
        atomicMin(&s_lo[3 * slot + j], k);
        atomicMax(&s_hi[3 * slot + j], k);
        atomicAdd(&s_sum[3 * slot + j], (unsigned long long)(long long)__double2ll_rn((double)q[j] * kFix));
      }
    }
    ins_out[p] = slot;
  }
  __syncthreads();
  for (int i = tid; i < n; i += kThreads) {
    if (!s_count[i]) continue;
    SlotStats* g = stats + (int64_t)s * kMaxInstances + i;
    atomicAdd(&g->count, s_count[i]);
    atomicMin(&g->first, s_first[i]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      atomicMin(&g->lo[j], s_lo[3 * i + j]);
      atomicMax(&g->hi[j], s_hi[3 * i + j]);
      atomicAdd(&g->sum[j], s_sum[3 * i + j]);
    }
  }
}

template <typename SemT>
__global__ __launch_bounds__(kThreads) void sp_finish_kernel(const int32_t* __restrict__ ins_out, const int32_t* __restrict__ batch_indices,
                                                             const SemT* __restrict__ sem, const SlotStats* __restrict__ stats,
                                                             int64_t N, int B, float* __restrict__ regions,
                                                             int32_t* __restrict__ num_points, int32_t* __restrict__ inst_sem) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (t < (int64_t)B * kMaxInstances) {  // the padded per-instance tables (columns past a scene's K: 0 / -1)
    const SlotStats& g = stats[t];
    num_points[t] = (int32_t)g.count;
    inst_sem[t] = g.count ? (int32_t)sem[g.first] : -1;
  }
  if (t >= N) return;
  const int32_t slot = ins_out[t];
  float r[9];
  if (slot >= 0) {
    const SlotStats& g = stats[(int64_t)batch_indices[t] * kMaxInstances + slot];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      r[j] = (float)(((double)(long long)g.sum[j] / kFix) / (double)g.count);
      r[3 + j] = order_float(g.lo[j]);
      r[6 + j] = order_float(g.hi[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 9; ++j) r[j] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) regions[t * 9 + j] = r[j];
}

}  // namespace

extern "C" int gpn_scene_prepare_max_instances() { return kMaxInstances; }

extern "C" size_t gpn_scene_prepare_ws_bytes(int B) {
  return gpn::align_up((size_t)B * kMaxInstances * sizeof(SlotStats)) + gpn::align_up((size_t)B * kMaxInstances * sizeof(int32_t)) +
         gpn::align_up((size_t)(B + 1) * sizeof(int64_t)) + 256;
}

extern "C" int gpn_scene_prepare(const float* points, const void* sem_labels, int sem_bytes, const int32_t* instance_labels,
                                 const int64_t* seg_offsets, int64_t N, int C, int B, const double* mats, const double* shifts,
                                 float* points_out, int32_t* batch_indices, int32_t* instance_out, float* regions,
                                 int32_t* num_points_per_instance, int32_t* instance_sem_labels, int64_t* num_instances_host,
                                 int32_t* overflow, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(N >= 0 && B >= 1 && C >= 0 && (sem_bytes == 2 || sem_bytes == 4 || sem_bytes == 8));
  GPN_CHECK_ARG(points && sem_labels && instance_labels && seg_offsets && points_out && batch_indices && instance_out && regions);
  GPN_CHECK_ARG(num_points_per_instance && instance_sem_labels && overflow);
  GPN_CHECK_ARG(N < (int64_t)0x7fffffff);
  gpn::WsCarver carve(ws, ws_bytes);
  SlotStats* stats = carve.take<SlotStats>((size_t)B * kMaxInstances);
  int32_t* id_list = carve.take<int32_t>((size_t)B * kMaxInstances);
  int64_t* k_dev = carve.take<int64_t>((size_t)B + 1);
  if (!carve.ok()) {
    gpn::set_error("gpn_scene_prepare: workspace too small");
    return GPN_ERR_WS;
  }
  GPN_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL(sp_ids_kernel, dim3(B), dim3(kIdsThreads), 0, stream, instance_labels, seg_offsets, id_list, k_dev, overflow,
                     stats, num_instances_host);
  GPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sp_points_kernel, dim3(kSplit, B), dim3(kThreads), 0, stream, points, instance_labels, seg_offsets, C, mats,
                     shifts, id_list, k_dev, points_out, batch_indices, instance_out, stats);
  GPN_CHECK_LAUNCH();
  const int64_t work = N > (int64_t)B * kMaxInstances ? N : (int64_t)B * kMaxInstances;
  const dim3 grid((unsigned)gpn::cdiv(work, kThreads));
  if (sem_bytes == 8)
    hipLaunchKernelGGL(sp_finish_kernel<int64_t>, grid, dim3(kThreads), 0, stream, instance_out, batch_indices,
                       static_cast<const int64_t*>(sem_labels), stats, N, B, regions, num_points_per_instance, instance_sem_labels);
  else if (sem_bytes == 4)
    hipLaunchKernelGGL(sp_finish_kernel<int32_t>, grid, dim3(kThreads), 0, stream, instance_out, batch_indices,
                       static_cast<const int32_t*>(sem_labels), stats, N, B, regions, num_points_per_instance, instance_sem_labels);
  else
    hipLaunchKernelGGL(sp_finish_kernel<int16_t>, grid, dim3(kThreads), 0, stream, instance_out, batch_indices,
                       static_cast<const int16_t*>(sem_labels), stats, N, B, regions, num_points_per_instance, instance_sem_labels);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}
