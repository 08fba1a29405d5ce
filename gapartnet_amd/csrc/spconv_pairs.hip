// spconv_pairs.hip — pair-compacted sparse convolution (forward and SubM dgrad launches) for gfx950, round 4.
//
// The output-stationary kernels of spconv_tiles.hip / spconv_fwd.hip put destination row i of a 16-row tile on MFMA row i for
// every tap: a tap is contracted for the whole tile as soon as ANY of its rows has that neighbour, so 52-81 % of the matrix
// pipe's row-slots multiply zeros (DESIGN "useful-row fraction 0.19-0.48"), and every (tile, tap) re-fetches the tap's weight
// fragments (1 KiB per 16 x 16 block) for 16 rows of work - the B operand is as much vector-memory traffic as the gathered A rows.
// Here the rulebook's PAIR LISTS drive the contraction instead (gather - GEMM - scatter-add, the formulation spconv itself
// uses, network/backbone.py:25-36 via SubMConv3d), with the scatter kept on chip and ordered:
//   * a wave owns RW (32 ... 128) consecutive destination rows and NT output column tiles; its fp32 accumulators [RW][16 NT]
//     live in LDS (8 KB for 64 rows x 32 columns);
//   * taps in ascending order; the pairs of tap k whose destination is in the wave's rows are ONE contiguous range of the
//     (tap, dst)-ordered lists (tile_off gives its ends: the lists are cut at 32-row boundaries), walked in chunks of 16 pairs:
//     MFMA row i = pair i of the chunk - every row-slot is a real pair except in a range's last chunk;
//   * the tap's weight fragments are read once per (wave, tap) and serve every chunk of the range (RW / 16 x fewer fetches);
//   * a chunk's 16 x 16 products (one MFMA chain per tap from zero, input blocks ascending - the order of the other kernels)
//     are added to the accumulator rows of their destinations with ds_add_f32: within a tap a destination occurs once, taps
//     follow each other in program order of ONE wave, so the sums are those of the output-stationary kernels bit for bit
//     (acc = ((0 + p_k0) + p_k1) + ...), deterministic, no global atomics, every output row written once.
// No tile order and no neighbour table are needed: rows stay in voxel order.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "bn_stats.h"
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// CB = input blocks of 16 channels, NT = output column tiles of this wave
template <int CB, int NT>
__global__ __launch_bounds__(256) void spconv_pairs_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                           const int32_t* __restrict__ pair_src, const int32_t* __restrict__ pair_dst,
                                                           const int32_t* __restrict__ tile_off, int K, int64_t n_dst, int n_tiles32,
                                                           int rw_tiles, int n_units, int nt_total, int col_groups,
                                                           size_t packed_bytes, int accumulate, gpn::ConvStats stats,
                                                           float* __restrict__ out) {
  if (blockIdx.y) {  // the launch's second problem (gpn::ConvTwin)
    in = stats.twin.in, packed = stats.twin.packed, out = stats.twin.out;
    stats.slab = stats.twin.slab, stats.x = stats.twin.x, stats.y = stats.twin.y, stats.mean = stats.twin.mean,
    stats.invstd = stats.twin.invstd;
  }
  extern __shared__ __attribute__((aligned(16))) float pairs_lds[];
  constexpr int W = NT * 16;  // accumulator columns of a wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int RW = rw_tiles * 32;
  float* __restrict__ acc = pairs_lds + (size_t)wave * RW * W;
  // XCD-contiguous unit order (as the other conv kernels): the rows a workgroup's waves gather come through one L2
  const int per8 = (((n_units + 3) >> 2) + 7) >> 3;
  const int wg = (int)(blockIdx.x & 7) * per8 + (int)(blockIdx.x >> 3);
  const int unit = __builtin_amdgcn_readfirstlane(wg * 4 + wave);
  if (unit >= n_units) return;  // whole wave; no barrier in this kernel
  const int sb = unit / col_groups;
  const int nt0 = (unit - sb * col_groups) * NT;
  const int t0 = sb * rw_tiles;
  const int t1 = t0 + rw_tiles < n_tiles32 ? t0 + rw_tiles : n_tiles32;
  const int64_t row0 = (int64_t)t0 * 32;
  constexpr int cin = CB * 16;
  const int cout = nt_total * 16;

  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)packed_bytes, 0x00020000);

  for (int e = lane; e < RW * W; e += 64) acc[e] = 0.f;

  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const uint32_t bvoff = (uint32_t)lane * 16u;
  for (int k = 0; k < K; ++k) {
    const int32_t* __restrict__ off_k = tile_off + (int64_t)k * (n_tiles32 + 1);
    const int p0 = __builtin_amdgcn_readfirstlane(off_k[t0]), p1 = __builtin_amdgcn_readfirstlane(off_k[t1]);
    if (p0 >= p1) continue;
    f32x4 b[CB][NT];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        b[cb][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)bvoff, ((k * CB + cb) * nt_total + nt0 + nt) * 1024, 0));
    for (int c0 = p0; c0 < p1; c0 += 16) {
      const int pi = c0 + i16;
      const bool valid = pi < p1;
      const int32_t s = valid ? pair_src[pi] : -1;
      const int32_t d = valid ? pair_dst[pi] : -1;
      const uint32_t aoff = s < 0 ? 0x80000000u : ((uint32_t)s * (uint32_t)cin + 4u * (uint32_t)g) * 4u;
      f32x4 a[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
        a[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)aoff, cb * 64, 0));
      f32x4 part[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) part[nt] = zero;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][q], b[cb][nt][q], part[nt], 0, 0, 0);
      // D[row = 4g + r][col = i16]: add to the accumulator row of pair (4g + r)'s destination
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int32_t dr = __shfl(d, 4 * g + r, 64);
        if (dr >= 0) {
          float* row = acc + (size_t)(dr - (int32_t)row0) * W + i16;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) __hip_atomic_fetch_add(row + nt * 16, part[nt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS adds of this wave have landed

  // ---- accumulators -> out; BatchNorm column sums per 16-row tile (bn_stats.h), as the other conv kernels ----
  const bool st_fwd = stats.slab != nullptr && stats.x == nullptr;
  const bool st_bwd = stats.slab != nullptr && stats.x != nullptr;
  const int tiles16 = rw_tiles * 2;
  for (int t = 0; t < tiles16; ++t) {
    const int64_t trow0 = row0 + (int64_t)t * 16;
    if (trow0 >= n_dst) break;
    const int64_t tile = trow0 >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint32_t col = (uint32_t)((nt0 + nt) * 16 + i16);
      float mu = 0.f, is = 1.f;
      if (st_bwd) mu = stats.mean[col], is = stats.invstd[col];
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = trow0 + 4 * g + r;
        if (row < n_dst) {
          const uint32_t e = (uint32_t)row * (uint32_t)cout + col;
          float v = acc[(size_t)(t * 16 + 4 * g + r) * W + nt * 16 + i16];
          if (accumulate) v += out[e];
          out[e] = v;
          if (st_fwd) {
            s0 += (double)v;
            s1 += (double)v * (double)v;
          } else if (st_bwd) {
            const float by = stats.relu ? stats.y[e] : 1.f;
            const float gm = (stats.relu && !(by > 0.f)) ? 0.f : v;
            s0 += (double)gm;
            s1 += (double)gm * (double)((stats.x[e] - mu) * is);
          }
        }
      }
      if (st_fwd) gpn::stat_add<false>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
      else if (st_bwd) gpn::stat_add<true>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
    }
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
std::atomic<int> g_pairs_mode{env_int("GPN_CONV_PAIRS", 0)};           // 0 = off, 1 = on for the shapes below
std::atomic<int> g_pairs_rw{env_int("GPN_PAIRS_ROWS", 64)};             // rows per wave (32, 64, 128)
std::atomic<int> g_pairs_nt{env_int("GPN_PAIRS_COLS", 0)};              // column tiles per wave (0 = all, <= 4)
std::atomic<int64_t> g_pairs_min_rows{(int64_t)env_int("GPN_PAIRS_MIN_ROWS", 4096)};
std::atomic<int64_t> g_pairs_max_rows{(int64_t)env_int("GPN_PAIRS_MAX_ROWS", 1 << 30)};

template <int CB, int NT>
int launch_pairs(const float* in, const float* packed, const int32_t* pair_src, const int32_t* pair_dst, const int32_t* tile_off,
                 int K, int64_t n_dst, int nt_total, int rw_tiles, int accumulate, const gpn::ConvStats& stats, float* out,
                 hipStream_t stream) {
  const int n_tiles32 = (int)gpn::cdiv(n_dst, 32);
  const int col_groups = nt_total / NT;
  const int n_units = (int)gpn::cdiv(n_tiles32, rw_tiles) * col_groups;
  const size_t packed_bytes = (size_t)K * CB * nt_total * 1024;
  const size_t lds = (size_t)4 * rw_tiles * 32 * NT * 16 * sizeof(float);
  const dim3 grid((unsigned)(gpn::cdiv(gpn::cdiv(n_units, 4), 8) * 8), stats.twin.in ? 2 : 1);
  hipLaunchKernelGGL((spconv_pairs_kernel<CB, NT>), grid, dim3(256), lds, stream, in, packed, pair_src, pair_dst, tile_off, K, n_dst,
                     n_tiles32, rw_tiles, n_units, nt_total, col_groups, packed_bytes, accumulate, stats, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int CB>
int dispatch_pairs_nt(int NT, const float* in, const float* packed, const int32_t* pair_src, const int32_t* pair_dst,
                      const int32_t* tile_off, int K, int64_t n_dst, int nt_total, int rw_tiles, int accumulate,
                      const gpn::ConvStats& stats, float* out, hipStream_t stream) {
  switch (NT) {
    case 1: return launch_pairs<CB, 1>(in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 2: return launch_pairs<CB, 2>(in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 3: return launch_pairs<CB, 3>(in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    default: return launch_pairs<CB, 4>(in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
  }
}

}  // namespace

namespace gpn {

// shapes the pair kernel is instantiated for and switched on for
bool spconv_pairs_supported(int K, int64_t n_dst, int cin, int cout) {
  if (g_pairs_mode.load(std::memory_order_relaxed) == 0) return false;
  const int CB = cin / 16;
  if (cin % 16 || cout % 16 || CB < 1 || CB > 8 || K < 1) return false;
  if (n_dst < g_pairs_min_rows.load(std::memory_order_relaxed) || n_dst > g_pairs_max_rows.load(std::memory_order_relaxed)) return false;
  return n_dst * (int64_t)std::max(cin, cout) * 4 < ((int64_t)1 << 31);
}

int spconv_pairs_launch(const float* in, const float* packed, const int32_t* pair_src, const int32_t* pair_dst,
                        const int32_t* tile_off, int K, int64_t n_dst, int cin, int cout, int accumulate, const ConvStats& stats,
                        float* out, hipStream_t stream) {
  GPN_CHECK_ARG(in && packed && pair_src && pair_dst && tile_off && out);
  const int CB = cin / 16, nt_total = cout / 16;
  // column tiles per wave: all of them up to 4 (the gathered rows are shared by the columns), a divisor of the layer's count
  int NT = g_pairs_nt.load(std::memory_order_relaxed);
  if (NT <= 0 || NT > 4) NT = 4;
  while (NT > 1 && nt_total % NT) --NT;
  int rw = g_pairs_rw.load(std::memory_order_relaxed);
  rw = rw <= 32 ? 32 : (rw <= 64 ? 64 : 128);
  while (rw > 32 && (size_t)4 * rw * NT * 16 * sizeof(float) > 65536) rw >>= 1;
  const int rw_tiles = rw / 32;
  switch (CB) {
    case 1: return dispatch_pairs_nt<1>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 2: return dispatch_pairs_nt<2>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 3: return dispatch_pairs_nt<3>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 4: return dispatch_pairs_nt<4>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 5: return dispatch_pairs_nt<5>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 6: return dispatch_pairs_nt<6>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    case 7: return dispatch_pairs_nt<7>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
    default: return dispatch_pairs_nt<8>(NT, in, packed, pair_src, pair_dst, tile_off, K, n_dst, nt_total, rw_tiles, accumulate, stats, out, stream);
  }
}

}  // namespace gpn

// knobs of the pair-compacted kernel: mode (0 off / 1 on; < 0 keeps), rows per wave (32 / 64 / 128), column tiles per wave
// (0 = as many as fit, <= 4), smallest / largest layer (rows) it takes; a negative argument leaves that setting unchanged
extern "C" int gpn_spconv_pairs_config(int mode, int rows_per_wave, int cols_per_wave, int64_t min_rows, int64_t max_rows) {
  if (mode >= 0) g_pairs_mode.store(mode, std::memory_order_relaxed);
  if (rows_per_wave >= 0) g_pairs_rw.store(rows_per_wave, std::memory_order_relaxed);
  if (cols_per_wave >= 0) g_pairs_nt.store(cols_per_wave, std::memory_order_relaxed);
  if (min_rows >= 0) g_pairs_min_rows.store(min_rows, std::memory_order_relaxed);
  if (max_rows >= 0) g_pairs_max_rows.store(max_rows, std::memory_order_relaxed);
  return GPN_OK;
}

// the conv over pair lists as an entry point of its own (tests, tools/conv_pairs_bench.py): out [n_dst, cout] = conv(in) with
// packed weights (gpn_spconv_pack_weights); lists / tile_off as gpn_rulebook_subm3 writes them
extern "C" int gpn_spconv_fwd_pairs(const float* in, const float* packed_w, const int32_t* pair_src, const int32_t* pair_dst,
                                    const int32_t* tile_off, int K, int64_t n_dst, int cin, int cout, float* out,
                                    gpn_stream_t stream) {
  GPN_CHECK_ARG(K >= 1 && n_dst >= 0 && cin >= 16 && cin % 16 == 0 && cin <= 128 && cout >= 16 && cout % 16 == 0);
  if (n_dst == 0) return GPN_OK;
  return gpn::spconv_pairs_launch(in, packed_w, pair_src, pair_dst, tile_off, K, n_dst, cin, cout, 0, gpn::ConvStats(), out,
                                  (hipStream_t)stream);
}
