// gpn_core.hip — error reporting, entry-point registry and the in-library hipEvent profiler.
#include <cstdarg>
#include <algorithm>
#include <mutex>
#include <vector>

#include "gpn_common.h"

namespace {
thread_local char g_err[512] = "";

struct ProfRec {
  hipEvent_t a, b;
  int64_t tag;  // what the launch was (gpn::prof_shape_tag for the conv kernels), 0 = untagged
};
struct ProfLive {  // work accounted at a buffer bound whose live row count is a device counter: scaled when the totals are read
  const int64_t* rows_dev;
  int64_t bound;
  double flops, bytes;
};
struct ProfSlot {
  std::vector<ProfRec> recs;
  std::vector<ProfLive> live;
  double flops = 0, bytes = 0;
  int64_t launches = 0;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
ProfSlot g_prof[GPN_K_COUNT];

const char* kEntryPoints[] = {
    "gpn_voxelize", "gpn_voxelize_ws_bytes", "gpn_rulebook_subm3", "gpn_rulebook_subm3_ws_bytes",
    "gpn_rulebook_down", "gpn_rulebook_down_ws_bytes", "gpn_rulebook_down_lists",
    "gpn_rulebook_down_lists_ws_bytes", "gpn_rulebook_level_counts", "gpn_rulebook_identity", "gpn_rulebook_level_counts_ws_bytes", "gpn_voxelize_ex", "gpn_voxelize_scenes", "gpn_voxelize_scenes_sorted", "gpn_voxelize_scenes_ws_bytes", "gpn_spconv_pack_weights", "gpn_spconv_fwd", "gpn_spconv_fwd_ws_bytes", "gpn_spconv_fwd_ordered", "gpn_rulebook_tile_order_ws_bytes", "gpn_rulebook_tile_order", "gpn_spconv_tiles_min_tiles", "gpn_spconv_direct_split", "gpn_spconv_msplit", "gpn_spconv_fwd_w", "gpn_spconv_fwd_w_ws_bytes", "gpn_spconv_wgrad",
    "gpn_spconv_wgrad_ws_bytes", "gpn_gather_rows", "gpn_scatter_rows_csr", "gpn_bn_ws_bytes", "gpn_bn_fwd_train", "gpn_bn_fwd_eval", "gpn_bn_bwd", "gpn_net_ws_bytes", "gpn_net_bn_fusion", "gpn_net_wgrad_group", "gpn_net_forward", "gpn_linear_supported", "gpn_linear_fwd", "gpn_linear_bwd_ws_bytes", "gpn_linear_bwd", "gpn_net_backward", "gpn_net_forward_pair", "gpn_net_backward_pair", "gpn_point_losses_ws_bytes", "gpn_point_losses_fwd", "gpn_point_losses_fwd_metrics", "gpn_point_losses_bwd", "gpn_score_loss", "gpn_npcs_loss_fwd", "gpn_npcs_loss_bwd", "gpn_ball_query", "gpn_ball_query_grid_ws_bytes", "gpn_ball_query_grid", "gpn_ccl",
    "gpn_ccl_ws_bytes", "gpn_segmented_reduce", "gpn_segmented_maxpool_fwd", "gpn_segmented_maxpool_bwd",
    "gpn_instance_iou", "gpn_nms", "gpn_nms_ws_bytes", "gpn_pn2_ball_query", "gpn_pn2_group_points",
    "gpn_pn2_group_points_grad", "gpn_pn2_gather_points", "gpn_pn2_gather_points_grad",
    "gpn_pn2_furthest_point_sampling", "gpn_pn2_furthest_point_sampling_ws_bytes",
    "gpn_pn2_furthest_point_sampling_ws", "gpn_pn2_three_nn", "gpn_pn2_knn", "gpn_pn2_three_interpolate",
    "gpn_pn2_three_interpolate_grad", "gpn_proposals_max_proposals", "gpn_proposals_build_ws_bytes", "gpn_proposals_build", "gpn_proposals_voxel_mean",
    "gpn_proposals_voxel_mean_bwd", "gpn_proposals_revoxelize_ws_bytes", "gpn_proposals_revoxelize", "gpn_proposals_postprocess_ws_bytes", "gpn_proposals_postprocess", "gpn_proposals_postprocess_lds_proposals", "gpn_backbone_prepare_desc_words", "gpn_backbone_prepare_arena_bytes", "gpn_backbone_prepare", "gpn_scene_prepare_max_instances", "gpn_scene_prepare_ws_bytes", "gpn_scene_prepare", "gpn_pose_fit_ws_bytes", "gpn_pose_fit", "gpn_copy_many", "gpn_adam_blocks", "gpn_adam_step", "gpn_adam_step_gated", "gpn_prof_enable", "gpn_prof_bracket_overhead_us", "gpn_prof_reset", "gpn_prof_get", "gpn_prof_get_launches",
    "gpn_rulebook_subm3_dev", "gpn_rulebook_down_dev_ws_bytes", "gpn_rulebook_down_dev", "gpn_rulebook_down_lists_dev", "gpn_rulebook_identity_dev", "gpn_gather_rows_dev", "gpn_scatter_rows_csr_dev", "gpn_proposals_voxel_mean_dev", "gpn_proposals_targets_dev", "gpn_linear_fwd_dev", "gpn_linear_bwd_dev", "gpn_segmented_maxpool_fwd_dev", "gpn_segmented_maxpool_bwd_dev", "gpn_instance_iou_dev", "gpn_score_loss_dev", "gpn_npcs_loss_fwd_dev", "gpn_npcs_loss_bwd_dev",
    "gpn_last_error", "gpn_version"};
}  // namespace

namespace gpn {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

ProfScope::ProfScope(int kernel_id, hipStream_t s, double flops, double bytes, int64_t tag_) : id(kernel_id), stream(s), tag(tag_) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof[id].flops += flops;
  g_prof[id].bytes += bytes;
  g_prof[id].launches += 1;
  if (hipEventCreate(&start) != hipSuccess) { start = nullptr; return; }
  hipEventRecord(start, stream);
}
ProfScope::ProfScope(int kernel_id, hipStream_t s, double flops, double bytes, const int64_t* rows_dev, int64_t bound, int64_t tag_)
    : ProfScope(kernel_id, s, rows_dev ? 0.0 : flops, rows_dev ? 0.0 : bytes, tag_) {
  if (!g_prof_on || !rows_dev) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof[id].live.push_back({rows_dev, bound, flops, bytes});
}
ProfScope::~ProfScope() {
  if (!start) return;
  hipEvent_t stop;
  if (hipEventCreate(&stop) != hipSuccess) return;
  hipEventRecord(stop, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof[id].recs.push_back({start, stop, tag});
}
}  // namespace gpn

extern "C" {
const char* gpn_last_error(void) { return g_err; }
int gpn_version(void) { return 1; }
int gpn_num_entry_points(void) { return (int)(sizeof(kEntryPoints) / sizeof(kEntryPoints[0])); }
const char* gpn_entry_point_name(int i) {
  return (i >= 0 && i < gpn_num_entry_points()) ? kEntryPoints[i] : nullptr;
}

// fixed cost a (start event, launch, stop event) bracket adds to any kernel: measured around an empty kernel on the given
// stream (median of 33 brackets), so that callers can subtract it from hipEvent-measured launch durations
__global__ void gpn_prof_noop_kernel() {}

int gpn_prof_bracket_overhead_us(gpn_stream_t stream_, double* overhead_us) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(overhead_us != nullptr);
  constexpr int kReps = 33;
  hipEvent_t a[kReps], b[kReps];
  for (int i = 0; i < kReps; ++i) {
    GPN_CHECK_HIP(hipEventCreate(&a[i]));
    GPN_CHECK_HIP(hipEventCreate(&b[i]));
  }
  for (int i = 0; i < kReps; ++i) {
    GPN_CHECK_HIP(hipEventRecord(a[i], stream));
    hipLaunchKernelGGL(gpn_prof_noop_kernel, dim3(1), dim3(64), 0, stream);
    GPN_CHECK_HIP(hipEventRecord(b[i], stream));
  }
  GPN_CHECK_HIP(hipEventSynchronize(b[kReps - 1]));
  float t[kReps];
  for (int i = 0; i < kReps; ++i) {
    GPN_CHECK_HIP(hipEventElapsedTime(&t[i], a[i], b[i]));
    hipEventDestroy(a[i]);
    hipEventDestroy(b[i]);
  }
  std::sort(t, t + kReps);
  *overhead_us = (double)t[kReps / 2] * 1e3;
  return GPN_OK;
}

int gpn_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return GPN_OK;
}
int gpn_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof) {
    for (auto& r : s.recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    s.recs.clear();
    s.live.clear();
    s.flops = s.bytes = 0;
    s.launches = 0;
  }
  return GPN_OK;
}
int gpn_prof_get(int kernel_id, int64_t* launches_host, double* ms_host, double* flops_host,
                 double* bytes_host) {
  GPN_CHECK_ARG(kernel_id >= 0 && kernel_id < GPN_K_COUNT);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfSlot& s = g_prof[kernel_id];
  double ms = 0;
  for (auto& r : s.recs) {
    GPN_CHECK_HIP(hipEventSynchronize(r.b));
    float t = 0;
    GPN_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
  }
  // launches over device-counted rows: their work at the LIVE row count (the counters still hold the instrumented step's values:
  // the caller reads the totals before it runs another step)
  double live_flops = 0, live_bytes = 0;
  for (auto& l : s.live) {
    int64_t n = 0;
    GPN_CHECK_HIP(hipMemcpy(&n, l.rows_dev, sizeof(n), hipMemcpyDeviceToHost));
    n = n < 0 ? 0 : (n > l.bound ? l.bound : n);
    const double share = l.bound > 0 ? (double)n / (double)l.bound : 0.0;
    live_flops += l.flops * share;
    live_bytes += l.bytes * share;
  }
  if (launches_host) *launches_host = s.launches;
  if (ms_host) *ms_host = ms;
  if (flops_host) *flops_host = s.flops + live_flops;
  if (bytes_host) *bytes_host = s.bytes + live_bytes;
  return GPN_OK;
}
// the same measurements launch by launch, in launch order: ms[i] / tag[i] of launch i (tag: see gpn.h); returns the number of
// records through *count_host (only the first `cap` are written)
int gpn_prof_get_launches(int kernel_id, int64_t cap, double* ms_host, int64_t* tag_host, int64_t* count_host) {
  GPN_CHECK_ARG(kernel_id >= 0 && kernel_id < GPN_K_COUNT && cap >= 0 && count_host && (cap == 0 || (ms_host && tag_host)));
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfSlot& s = g_prof[kernel_id];
  int64_t i = 0;
  for (auto& r : s.recs) {
    if (i < cap) {
      GPN_CHECK_HIP(hipEventSynchronize(r.b));
      float t = 0;
      GPN_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
      ms_host[i] = t;
      tag_host[i] = r.tag;
    }
    ++i;
  }
  *count_host = i;
  return GPN_OK;
}
}
