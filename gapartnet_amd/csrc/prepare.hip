// prepare.hip — the preparation of one training / validation batch for a sparse U-Net in ONE library call (include/gpn.h section
// BP): scene-batch voxelisation, its one host read, and the rulebook pyramid of the backbone - SubM k3 tables (+ tile order) on
// every level, the stride-2 / inverse maps between levels (+ tile order), the k = 1 identity maps of the decoder's shortcut convs.
//
// Reference: the loader voxelises every scene on the CPU (gapartnet/dataset/gapartnet.py:179-205) and spconv builds the indice pairs
// inside the forward pass (gapartnet/network/backbone.py:19-36, 74-90, 149-152).  Here the pieces existed as separate entry points
// (sections V, K1, K2) and the host called them one by one from Python one step ahead of training: 17 library calls, ~230 small
// allocations and the wait for the voxeliser's sizes - ~1.3 ms of the training thread's time per step (profiles/r04_host_cprofile.txt),
// on a step where that thread is as busy as the GPU.  This call does the same launches from ONE native loop into ONE arena the
// caller allocated (one allocation instead of ~230), and waits for the sizes itself (a blocking-sync event).  Every sub-buffer's
// place is reported in a host-side descriptor; the arithmetic is that of the separate entry points (which this file calls), so
// results are bit-identical to the per-call path (tests/test_gpu_prepare.py).
// Measured (round 5, profiles/r05_findings.md): issued by the training thread it is time-neutral against the per-call path (7.57
// against 7.58 ms per step, three interleaved pairs: the step is GPU-bound); issued by a WORKER thread - the reason it blocks
// internally, and ctypes releases the interpreter lock around it - the step got SLOWER, 7.52 -> 7.78 ms and 12.7 -> 19.5 CPU-ms, with
// no Python in the worker at all: a second host thread launching beside the training thread (and the weight-gradient helper)
// costs more than the preparation it takes over.  The device prefetcher therefore calls it inline.
#include "gpn_common.h"

#include <algorithm>
#include <vector>

namespace {

// descriptor layout (int64 words, host memory).  Offsets are BYTES from the arena base; -1 = absent.
constexpr int kHead = 16;        // [0] V, [1..3] spatial shape, [4] dropped points, [5] fallback flag, [6] levels, [7] bytes used,
                                 // [8] vf, [9] idx4, [10] pid, [11] order, [12] vstart (offsets)
constexpr int kRbWords = 10;     // nbr, pair_src, pair_dst, tile_off, num_pairs, nbr_p, perm (offsets), n_src, n_dst, K
constexpr int kLevelWords = 8 + 4 * kRbWords;  // rows, shape[3], indices offset, 3 spare; rulebooks subm, down_fwd, down_bwd, ident

struct Arena {
  char* base;
  size_t used = 0, cap;
  template <typename T>
  T* take(size_t count, int64_t* off) {
    const size_t bytes = gpn::align_up(count * sizeof(T));
    T* p = reinterpret_cast<T*>(base + used);
    if (off) *off = (int64_t)used;
    used += bytes;
    return p;
  }
  bool ok() const { return used <= cap; }
};

inline int64_t tiles32(int64_t n) { return (n + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS; }

size_t level_ws_bytes(int64_t n) {
  size_t w = gpn_rulebook_subm3_ws_bytes(n);
  w = std::max(w, gpn_rulebook_tile_order_ws_bytes(n));
  w = std::max(w, gpn_rulebook_down_ws_bytes(n));
  w = std::max(w, gpn_rulebook_down_lists_ws_bytes(n, n));
  return w;
}

// bytes of one level's tables for n rows (and at most n coarse rows below it)
size_t level_table_bytes(int64_t n) {
  const size_t m = (size_t)(n > 0 ? n : 1);
  size_t b = 0;
  auto add = [&](size_t count, size_t elem) { b += gpn::align_up(count * elem); };
  add(27 * m + 1, 4), add(27 * m, 4), add(27 * m, 4), add(27 * (tiles32(m) + 1), 4), add(1, 8);  // subm
  add(m / 16 * 16 + 32, 4), add(27 * m + 1, 4);                                                   // its tile order
  add(m * 4, 4), add(m, 4), add(m, 4), add(1, 8);                                                 // coarse indices, fine_to_coarse, tap, num_out
  add(8 * m + 1, 4), add(m, 4), add(m, 4), add(8 * (tiles32(m) + 1), 4);                          // down fwd
  add(8 * m + 1, 4), add(m, 4), add(m, 4), add(8 * (tiles32(m) + 1), 4), add(1, 8);               // down bwd, num_pairs
  add(m / 16 * 16 + 32, 4), add(8 * m + 1, 4);                                                    // its tile order
  add(m, 4), add(tiles32(m) + 1, 4), add(m + 1, 4), add(1, 8);                                    // identity
  return b;
}

}  // namespace

extern "C" int gpn_backbone_prepare_desc_words(int n_levels) { return kHead + n_levels * kLevelWords; }

// an arena that always suffices: every level bounded by the point count M
extern "C" size_t gpn_backbone_prepare_arena_bytes(int64_t M, int C, int64_t S, int n_levels) {
  const size_t m = (size_t)(M > 0 ? M : 1);
  size_t b = gpn::align_up(m * C * 4) + gpn::align_up(m * 16) + 3 * gpn::align_up((m + 1) * 4) + gpn::align_up((size_t)(8 + n_levels) * 8);
  b += gpn::align_up(gpn_voxelize_scenes_ws_bytes(M, C, S, n_levels > 1 ? n_levels - 1 : 0));
  b += gpn::align_up(level_ws_bytes(M));
  b += (size_t)n_levels * level_table_bytes(M);
  return b + 4096;
}

extern "C" int gpn_backbone_prepare(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C, int64_t S,
                                    const float* voxel_size_host, int n_levels, uint32_t ident_levels, int64_t tile_order_min_rows,
                                    int tile_order_block, int device, void* arena_, size_t arena_bytes, int64_t* desc_host,
                                    int64_t* pinned_stats_host, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(points && feats && seg_offsets && voxel_size_host && arena_ && desc_host && pinned_stats_host);
  GPN_CHECK_ARG(M >= 1 && C >= 1 && S >= 1 && n_levels >= 1 && n_levels <= 16);
  GPN_CHECK_HIP(hipSetDevice(device));  // (a worker thread of the caller: the device is not inherited)
  const int words = kHead + n_levels * kLevelWords;
  for (int i = 0; i < words; ++i) desc_host[i] = -1;
  Arena a{static_cast<char*>(arena_), 0, arena_bytes};
  // ---- voxelisation (section V: gpn_voxelize_scenes), the batch's ONE host read ------------------------------------------------
  const int coarse = n_levels - 1;
  float* vf = a.take<float>((size_t)M * C, &desc_host[8]);
  int32_t* idx4 = a.take<int32_t>((size_t)M * 4, &desc_host[9]);
  int32_t* pid = a.take<int32_t>((size_t)M, &desc_host[10]);
  int32_t* order = a.take<int32_t>((size_t)M, &desc_host[11]);
  int32_t* vstart = a.take<int32_t>((size_t)M + 1, &desc_host[12]);
  int64_t* stats = a.take<int64_t>((size_t)(8 + coarse), nullptr);
  const size_t vws = gpn_voxelize_scenes_ws_bytes(M, C, S, coarse);
  const size_t lws = level_ws_bytes(M);
  const size_t ws_bytes = std::max(vws, lws);
  void* ws = a.take<char>(ws_bytes, nullptr);
  if (!a.ok()) {
    gpn::set_error("gpn_backbone_prepare: arena too small (%zu bytes before the rulebooks, %zu given)", a.used, arena_bytes);
    return GPN_ERR_WS;
  }
  int rc = gpn_voxelize_scenes(points, feats, seg_offsets, M, C, S, voxel_size_host, coarse, vf, idx4, pid, order, vstart, stats, ws,
                               ws_bytes, stream_);
  if (rc) return rc;
  GPN_CHECK_HIP(hipMemcpyAsync(pinned_stats_host, stats, sizeof(int64_t) * (size_t)(8 + coarse), hipMemcpyDeviceToHost, stream));
  {  // wait for the copy WITHOUT spinning (a blocking-sync event: the thread sleeps; hipStreamSynchronize busy-waits by default,
     // and this wait spans the kernels queued ahead on the stream - a core's worth of CPU per rank for nothing)
    hipEvent_t ev;
    GPN_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
    hipError_t e1 = hipEventRecord(ev, stream);
    hipError_t e2 = e1 == hipSuccess ? hipEventSynchronize(ev) : e1;
    (void)hipEventDestroy(ev);
    GPN_CHECK_HIP(e2);
  }
  const int64_t* st = pinned_stats_host;
  const int64_t V = st[0];
  desc_host[0] = V;
  for (int d = 0; d < 3; ++d) desc_host[1 + d] = V > 0 ? std::max<int64_t>(st[1 + d] + 1, 128) : 128;
  desc_host[4] = st[4];
  desc_host[5] = 0;
  desc_host[6] = n_levels;
  bool fallback = st[5] != 0 || V < 1;
  for (int l = 0; l < coarse; ++l) fallback = fallback || st[8 + l] < 1;
  if (fallback) {  // a cell index beyond the packed keys, an empty batch or an empty level: the caller's per-call path handles those
    desc_host[5] = 1;
    desc_host[7] = (int64_t)a.used;
    return GPN_OK;
  }
  // ---- the rulebook pyramid (sections K1 / K2) ---------------------------------------------------------------------------------
  // (Round 5 also measured these ~150 launches queued BEHIND an event recorded on the training stream at the prefetcher's mid-step
  // hook, so that they run beside the proposal stage instead of beside the backbone's decoder: 7.59 - 7.74 -> 7.73 - 7.95 ms per step,
  // four interleaved pairs - slower; the builders disturb the proposal stage's chain of small dependent kernels more than they
  // disturb big convs.  Removed.)
  const int32_t* indices = idx4;
  int64_t n = V;
  int32_t shape[3] = {(int32_t)desc_host[1], (int32_t)desc_host[2], (int32_t)desc_host[3]};
  for (int l = 0; l < n_levels; ++l) {
    int64_t* L = desc_host + kHead + l * kLevelWords;
    L[0] = n, L[1] = shape[0], L[2] = shape[1], L[3] = shape[2];
    L[4] = (int64_t)(reinterpret_cast<const char*>(indices) - a.base);
    auto put = [&](int which, int64_t o_nbr, int64_t o_src, int64_t o_dst, int64_t o_toff, int64_t o_np, int64_t o_nbrp, int64_t o_perm,
                   int64_t n_src, int64_t n_dst, int K) {
      int64_t* r = L + 8 + which * kRbWords;
      r[0] = o_nbr, r[1] = o_src, r[2] = o_dst, r[3] = o_toff, r[4] = o_np, r[5] = o_nbrp, r[6] = o_perm, r[7] = n_src, r[8] = n_dst, r[9] = K;
    };
    // SubM k = 3
    {
      int64_t o_nbr, o_src, o_dst, o_toff, o_np, o_perm = -1, o_nbrp = -1;
      int32_t* nbr = a.take<int32_t>((size_t)27 * n + 1, &o_nbr);
      int32_t* src = a.take<int32_t>((size_t)27 * n, &o_src);
      int32_t* dst = a.take<int32_t>((size_t)27 * n, &o_dst);
      int32_t* toff = a.take<int32_t>((size_t)27 * (tiles32(n) + 1), &o_toff);
      int64_t* np = a.take<int64_t>(1, &o_np);
      int32_t *perm = nullptr, *nbr_p = nullptr;
      const bool ordered = n >= tile_order_min_rows;
      if (ordered) {
        perm = a.take<int32_t>((size_t)((n + 15) / 16 * 16 + 16), &o_perm);
        nbr_p = a.take<int32_t>((size_t)27 * n + 1, &o_nbrp);
      }
      if (!a.ok()) break;
      rc = gpn_rulebook_subm3(indices, n, shape, nbr, src, dst, toff, np, ws, ws_bytes, stream_);
      if (rc) return rc;
      if (ordered) {
        rc = gpn_rulebook_tile_order(nbr, 27, n, tile_order_block, perm, nbr_p, ws, ws_bytes, stream_);
        if (rc) return rc;
      }
      put(0, o_nbr, o_src, o_dst, o_toff, o_np, o_nbrp, o_perm, n, n, 27);
    }
    // k = 1 identity map (the decoder's shortcut convs of this level)
    if (ident_levels & (1u << l)) {
      int64_t o_rows, o_toff, o_nbr, o_np;
      int32_t* rows = a.take<int32_t>((size_t)n, &o_rows);
      int32_t* toff = a.take<int32_t>((size_t)tiles32(n) + 1, &o_toff);
      int32_t* nbr = a.take<int32_t>((size_t)n + 1, &o_nbr);
      int64_t* np = a.take<int64_t>(1, &o_np);
      if (!a.ok()) break;
      rc = gpn_rulebook_identity(n, rows, toff, nbr, np, stream_);
      if (rc) return rc;
      put(3, o_nbr, o_rows, o_rows, o_toff, o_np, -1, -1, n, n, 1);
    }
    if (l + 1 == n_levels) break;
    // stride-2 map to the next level and its transpose
    {
      const int64_t n_out = st[8 + l];
      int64_t o_out, o_f2c, o_tap, o_nout;
      int32_t* out_idx = a.take<int32_t>((size_t)n * 4, &o_out);  // (capacity n rows: what gpn_rulebook_down may write)
      int32_t* f2c = a.take<int32_t>((size_t)n, &o_f2c);
      int32_t* tap = a.take<int32_t>((size_t)n, &o_tap);
      int64_t* nout = a.take<int64_t>(1, &o_nout);
      int64_t o_fn, o_fs, o_fd, o_ft, o_bn, o_bs, o_bd, o_bt, o_np, o_perm = -1, o_nbrp = -1;
      int32_t* fn = a.take<int32_t>((size_t)8 * n_out + 1, &o_fn);
      int32_t* fs = a.take<int32_t>((size_t)n, &o_fs);
      int32_t* fd = a.take<int32_t>((size_t)n, &o_fd);
      int32_t* ft = a.take<int32_t>((size_t)8 * (tiles32(n_out) + 1), &o_ft);
      int32_t* bn = a.take<int32_t>((size_t)8 * n + 1, &o_bn);
      int32_t* bs = a.take<int32_t>((size_t)n, &o_bs);
      int32_t* bd = a.take<int32_t>((size_t)n, &o_bd);
      int32_t* bt = a.take<int32_t>((size_t)8 * (tiles32(n) + 1), &o_bt);
      int64_t* np = a.take<int64_t>(1, &o_np);
      int32_t *perm = nullptr, *nbr_p = nullptr;
      const bool ordered = n >= tile_order_min_rows;  // (the transposed map: dst = fine rows)
      if (ordered) {
        perm = a.take<int32_t>((size_t)((n + 15) / 16 * 16 + 16), &o_perm);
        nbr_p = a.take<int32_t>((size_t)8 * n + 1, &o_nbrp);
      }
      if (!a.ok()) break;
      rc = gpn_rulebook_down(indices, n, S, shape, out_idx, f2c, tap, nout, ws, ws_bytes, stream_);
      if (rc) return rc;
      rc = gpn_rulebook_down_lists(f2c, tap, n, n_out, fn, fs, fd, ft, bn, bs, bd, bt, np, ws, ws_bytes, stream_);
      if (rc) return rc;
      if (ordered) {
        rc = gpn_rulebook_tile_order(bn, 8, n, tile_order_block, perm, nbr_p, ws, ws_bytes, stream_);
        if (rc) return rc;
      }
      put(1, o_fn, o_fs, o_fd, o_ft, o_np, -1, -1, n, n_out, 8);
      put(2, o_bn, o_bs, o_bd, o_bt, o_np, o_nbrp, o_perm, n_out, n, 8);
      indices = out_idx;
      n = n_out;
      for (int d = 0; d < 3; ++d) shape[d] /= 2;
    }
  }
  if (!a.ok()) {
    gpn::set_error("gpn_backbone_prepare: arena too small (%zu needed, %zu given)", a.used, arena_bytes);
    return GPN_ERR_WS;
  }
  desc_host[7] = (int64_t)a.used;
  return GPN_OK;
}
