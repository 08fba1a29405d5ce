// spconv_msplit.hip — the masked tap-split sparse convolution kernel (forward and dgrad launches) for gfx950 (round 6):
// every k = 27 / 8 layer BELOW the masked-tile kernel's 4096 row tiles - the U-Net's levels of 25k rows and fewer, the stride-2 /
// inverse convs between them, the proposal networks (device-counted rows).
//
// Those layers were the direct kernel's (spconv_fwd.hip): one workgroup per (row tile, ONE column tile), its four waves each
// walking a fixed quarter of the 27 taps - dead taps included, as requests that read zeros - with a four-stage operand ring,
// the first wave adding the four accumulators and storing.  What such a launch is made of, wave by wave (tools/probes/
// msplit_trace.py, profiles/r06_conv_trace_*.txt): at 25k rows x 48 channels four ROUNDS of waves (18 900 waves,
// 4 400 slots), the same rows gathered by three workgroups on different CUs, every third ring slot a dead tap's; at 489 rows x 96
// channels one wave per SIMD, a serial chain of 42 stages: 2.2 us of dependent MFMAs and the round trips of a four-stage ring.
//
// Here the work of a row tile is cut differently:
//   * a workgroup owns ONE row tile and NT column tiles (the widest of 4 / 3 / 2 / 1 that leaves 384 workgroups: a gathered row
//     piece then feeds NT MFMA column tiles from registers, as in the masked-tile kernel); its SP = 4 waves own the taps
//     [p TP, (p + 1) TP), TP = ceil(K / SP) - the tap ranges of the direct kernel's 4-way form;
//   * prologue per wave: the table entries of ITS taps for the 16 rows in one or two coalesced loads (lane = (row, tap)),
//     ballots find the taps any row has, their gather offsets go compacted into the wave's LDS slab - a dead tap costs
//     nothing afterwards, and the table is read once per row tile and tap instead of once per column tile;
//   * the tap loop: a ring of operand slots at STAGE granularity (stage = one 16-channel input block of one live tap: 1 row
//     piece + NT weight fragments requested, 4 NT MFMAs), D - 1 stages requested ahead, written out so that every slot index is
//     a compile-time constant and a slot is refilled in the sub-step after the one that consumed it (counted waits, no copies);
//   * epilogue: every wave leaves its NT accumulators in LDS; after ONE barrier wave q sums column tiles q, q + SP, ... over the
//     waves in wave order, stores them and adds the BatchNorm column sums (bn_stats.h) - or applies an inference pass's
//     BatchNorm (gpn::ConvAffine) - so the stores and the atomics of a row tile are spread over its waves.
// Summation order per output element: ascending taps inside a wave (two-level: a tap's CB x 16 products in one MFMA chain, the
// taps' sums added in fp32), then the waves' sums in wave order.  A tap a row does not have adds exact zeros, so a row's result
// depends neither on the rows it shares a tile with (the tile order) nor on NT, and SP never follows the row count (a
// device-counted launch and its exactly-sized twin give the same bits): with SP = 4 the result is bit-equal to the direct
// kernel's 4-way form.  SP = 9 (three taps per wave) is instantiated for the sweep and slower everywhere it was measured.
// What a launch of THIS kernel waits for (profiles/r06_findings.md): not operand bytes - with the weight fragments served from
// the L1s and the gathers made local it takes the same time (r06_conv_msplit_ablation.txt) - but its phases in lock-step: all
// waves of a round read their table entries, then contend for the MFMA pipe, then store.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "bn_stats.h"
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTaps = 27;

// GPN_MSPLIT_TRACE (tools/probes/msplit_trace.py; off in the product): every wave records when it started, finished its prologue,
// got its first operands, finished its tap loop, passed the barrier and finished (s_memrealtime, 100 MHz), its live taps and where
// it ran (HW_ID, XCC_ID)
#ifndef GPN_MSPLIT_TRACE
#define GPN_MSPLIT_TRACE 0
#endif
#if GPN_MSPLIT_TRACE
__device__ unsigned long long* g_msplit_trace = nullptr;  // [waves][10]
#endif
// operand registers of the ring (it holds as many whole taps - CB stages of 1 + NT requests - as fit; 96 / 144 measured: no gain)
// GPN_MSPLIT_ABL (measurement builds, wrong results by design): see the two uses in the tap loop
#ifndef GPN_MSPLIT_ABL
#define GPN_MSPLIT_ABL 0
#endif
#ifndef GPN_MSPLIT_OPERAND_REGS
#define GPN_MSPLIT_OPERAND_REGS 48
#endif

template <class F, int... I>
__device__ __forceinline__ void ms_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void ms_static_for(F&& f) {
  ms_static_for_impl(f, std::make_integer_sequence<int, N>());
}

// taps in the ring: as many whole taps (CB stages of 1 + NT requests each) as the operand-register budget holds, at least one -
// and at least three stages, so that a narrow layer still has requests in flight while a stage is in the MFMAs
constexpr int ms_ring_taps(int CB, int NT, int max_tp) {
  const int regs_per_tap = CB * (1 + NT) * 4;
  int st = GPN_MSPLIT_OPERAND_REGS / regs_per_tap;
  st = st < 1 ? 1 : st;
  while (st * CB < 3) ++st;
  st = st > 4 ? 4 : st;
  return st > max_tp ? max_tp : st;
}

// waves per SIMD asked of the register allocator per instantiation: the shapes whose register count sits a few registers above an
// allocation step (100 against 96 for 48 -> 48 channels with three column tiles per wave: 5 instead of 4 waves per SIMD for two
// spilled dwords outside the tap loop).  GPN_MSPLIT_OCC_HINTS=0 builds without (measurement).
#ifndef GPN_MSPLIT_OCC_HINTS
#define GPN_MSPLIT_OCC_HINTS 1
#endif
constexpr int ms_occ(int CB, int NT, int SP, bool DEV) {
  if (!GPN_MSPLIT_OCC_HINTS || DEV) return 1;
  if (SP == 4 && CB == 3 && NT == 3) return 5;
  return 1;
}
#define GPN_MSPLIT_OCC __attribute__((amdgpu_waves_per_eu(ms_occ(CB, NT, SP, DEV))))
template <int CB, int NT, int SP, bool DEV, bool EP>
__global__ __launch_bounds__(SP * 64) GPN_MSPLIT_OCC void spconv_msplit_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                                const int32_t* __restrict__ nbr, const int32_t* __restrict__ perm,
                                                                int K, int64_t n_dst, int n_tiles, int n_units, int nt_total,
                                                                int col_groups, size_t packed_bytes, int accumulate,
                                                                gpn::ConvStats stats, float* __restrict__ out,
                                                                const int64_t* __restrict__ n_dev) {
  if (blockIdx.y) {  // the launch's second problem (gpn::ConvTwin)
    in = stats.twin.in, packed = stats.twin.packed, out = stats.twin.out;
    stats.slab = stats.twin.slab, stats.x = stats.twin.x, stats.y = stats.twin.y, stats.mean = stats.twin.mean,
    stats.invstd = stats.twin.invstd;
    if constexpr (EP) {
      const float eps = stats.ep.eps;
      const int relu = stats.ep.relu;
      stats.ep = stats.twin.ep, stats.ep.eps = eps, stats.ep.relu = relu;
    }
  }
  constexpr int MAXTP = (kMaxTaps + SP - 1) / SP;  // taps of a wave at K = 27
  constexpr int NI = (MAXTP + 3) / 4;              // table loads of the prologue (4 taps x 16 rows each)
  constexpr int NF = (NT + SP - 1) / SP;           // column tiles a wave finalises
  constexpr uint32_t kOob = 0x80000000u;
  __shared__ uint32_t slab[SP][MAXTP + 1][16];  // per wave: byte offset of the gathered row (kOob = none) by live-tap slot
  __shared__ f32x4 red[SP][NT][64];             // the waves' accumulators

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  if constexpr (DEV) {  // the row count is a device counter (gpn::DevRows): n_dst was the buffers' bound, the grid a guess
    n_dst = gpn::live_rows(n_dev, n_dst);
    n_tiles = (int)((n_dst + 15) >> 4);
    n_units = n_tiles * col_groups;
  }
  const int TP = (K + SP - 1) / SP;
  const int tap0 = wave * TP;
  const int ntaps = K - tap0 < 0 ? 0 : (K - tap0 < TP ? K - tap0 : TP);

  // one unit = one row tile x NT column tiles, by the SP waves of the workgroup (`more`: another round follows, uniform)
  auto run_unit = [&](const int unit, const bool more) {
#if GPN_MSPLIT_TRACE
  const unsigned long long tr0 = wall_clock64();
#endif
  const int tile = unit / col_groups;
  const int nt0 = (unit - tile * col_groups) * NT;
  constexpr int cin = CB * 16;
  const int cout = nt_total * 16;

  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbr), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)packed_bytes, 0x00020000);
  const uint32_t col_bytes = (uint32_t)n_dst * 4u;

  // ---- prologue: the table entries of this wave's taps; live taps; compacted offsets into the wave's slab -------------------
  const int lt = g;  // lane = (row i16, tap sub-slot lt)
  const int64_t pos = (int64_t)tile * 16 + i16;
  const bool row_ok = pos < n_dst;
  const uint32_t tvoff = (uint32_t)(row_ok ? pos : n_dst - 1) * 4u;
  int32_t raw[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = i * 4 + lt;
    const int tc = j < ntaps ? tap0 + j : 0;  // (lanes past the wave's last tap re-read tap 0 and are masked below)
    raw[i] = -1;
    if (i * 4 < ntaps)  // (uniform)
      raw[i] = __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, (int)(tvoff + (uint32_t)tc * col_bytes), 0, 0));
  }
  // what the finalising wave needs of the output rows - their positions and, for a dgrad launch that carries a BatchNorm's
  // backward sums (bn_stats.h), that BatchNorm's x, y and channel statistics at the elements it will write: requested now, they
  // arrive during the contraction
  const bool st_bwd = stats.slab != nullptr && stats.x != nullptr;
  int32_t orow[4];
  if (perm) {
    const int4 pv = *reinterpret_cast<const int4*>(perm + (int64_t)tile * 16 + 4 * g);
    orow[0] = pv.x, orow[1] = pv.y, orow[2] = pv.z, orow[3] = pv.w;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) orow[r] = tile * 16 + 4 * g + r;
  }
  float bx[NF][4], by[NF][4], bmu[NF], bis[NF];
  if (st_bwd) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int nt = wave + f * SP;
      bmu[f] = 0.f, bis[f] = 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) bx[f][r] = 0.f, by[f][r] = 1.f;
      if (nt < NT) {
        const uint32_t col = (uint32_t)((nt0 + nt) * 16 + i16);
        bmu[f] = stats.mean[col], bis[f] = stats.invstd[col];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((int64_t)tile * 16 + 4 * g + r < n_dst) {
            const uint32_t e = (uint32_t)orow[r] * (uint32_t)cout + col;
            bx[f][r] = stats.x[e];
            if (stats.relu) by[f][r] = stats.y[e];
          }
      }
    }
  }
  uint32_t um = 0;  // bit j = some row of the tile has the wave's tap j
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = i * 4 + lt;
    const bool valid = row_ok && j < ntaps && raw[i] >= 0;
    const uint64_t b = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (((b >> (16 * s)) & 0xffffull) != 0) um |= 1u << (i * 4 + s);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = i * 4 + lt;
    const bool valid = row_ok && j < ntaps && raw[i] >= 0;
    const bool live = ((um >> j) & 1u) != 0u;
    const int slot = __builtin_popcount(um & ((1u << j) - 1u));
    if (live) slab[wave][slot][i16] = valid ? (uint32_t)raw[i] * (uint32_t)(cin * 4) : kOob;
  }
  int remaining = __builtin_popcount(um);
#if GPN_MSPLIT_TRACE
  const unsigned long long tr_taps = (unsigned long long)remaining;
  const unsigned long long tr1 = wall_clock64();
  unsigned long long tr1b = 0;
#endif

  f32x4 acc[NT];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = zero;

  const uint32_t bvoff = (uint32_t)lane * 16u;
  const uint32_t g16 = (uint32_t)g * 16u;
  // ---- the tap loop: a ring of D = ST x CB operand slots at STAGE granularity (stage = one 16-channel input block of one live
  // tap: 1 row-piece request + NT weight-fragment requests, 4 NT MFMAs), D - 1 stages requested ahead of the one in the MFMAs.
  // The body is the D sub-steps of ST taps written out - slot indices are compile-time constants, a slot is refilled in the
  // sub-step AFTER the one that consumed it (no copies, counted s_waitcnt; see spconv_tiles.hip for what the compiler does to a
  // ring written as a loop over slots); a scheduling barrier keeps every request block above the MFMAs of the older slot.
  // Requests past the wave's last live tap read at out-of-range offsets (zeros, no memory access) and are never multiplied.
  constexpr int ST = ms_ring_taps(CB, NT, MAXTP);
  constexpr int D = ST * CB;
  f32x4 ra[D], rb[D][NT];
  int to_issue = remaining, issued = 0;
  uint32_t cur_ao = kOob + g16, cur_woff = 0;
  auto issue_stage = [&](auto slot_tag, auto cb_tag) {
    constexpr int sl = decltype(slot_tag)::value, cb = decltype(cb_tag)::value;
    if constexpr (cb == 0) {  // the next live tap of the wave
      const bool has = to_issue > 0;
      const int k = tap0 + (has ? __builtin_ctz(um) : 0);
      um &= um - 1u;
      const int ls = issued < MAXTP ? issued : MAXTP;  // (the slab has MAXTP + 1 slots)
      to_issue -= 1, issued += 1;
      cur_woff = has ? (uint32_t)(k * CB * nt_total + nt0) * 1024u : 0x7ffffc00u - (uint32_t)(CB * nt_total) * 1024u;
      cur_ao = (has ? slab[wave][ls][i16] : kOob) + g16;  // (kOob + g16 stays out of range)
#if GPN_MSPLIT_ABL & 1  // measurement build: every tap reads tap 0's weight fragments (the weight traffic stays in the L1s)
      if (has) cur_woff = (uint32_t)nt0 * 1024u;
#endif
#if GPN_MSPLIT_ABL & 2  // measurement build: every tap gathers the tile's own rows (no scattered row reads)
      if (has) cur_ao = (uint32_t)(tile * 16 + i16) * (uint32_t)(cin * 4) + g16;
#endif
    }
    ra[sl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)cur_ao, cb * 64, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      rb[sl][nt] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(bvoff + cur_woff), (cb * nt_total + nt) * 1024, 0));
  };
  // two-level summation, as the direct kernel: a tap's CB * 16 products accumulate in `part` (one MFMA chain per column tile from
  // zero), the taps' sums are added to `acc`
  f32x4 part[NT];
  auto consume_stage = [&](auto slot_tag, auto cb_tag) {
    constexpr int sl = decltype(slot_tag)::value, cb = decltype(cb_tag)::value;
    if constexpr (cb == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) part[nt] = zero;
    }
    // column tiles interleaved: a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles, an independent one after 32
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        part[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[sl][s4], rb[sl][nt][s4], part[nt], 0, 0, 0);
    if constexpr (cb == CB - 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] += part[nt];
    }
  };
  ms_static_for<D - 1>([&](auto s) {
    issue_stage(s, std::integral_constant<int, decltype(s)::value % CB>());
  });
  __builtin_amdgcn_sched_barrier(0);
#if GPN_MSPLIT_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  tr1b = wall_clock64();
#endif
  while (remaining >= ST) {
    remaining -= ST;
    ms_static_for<D>([&](auto s) {
      constexpr int sv = decltype(s)::value;
      issue_stage(std::integral_constant<int, (sv + D - 1) % D>(), std::integral_constant<int, (sv + D - 1) % CB>());
      __builtin_amdgcn_sched_barrier(0);
      consume_stage(s, std::integral_constant<int, sv % CB>());
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  ms_static_for<D>([&](auto s) {  // the last remaining (< ST) taps: requested already
    constexpr int sv = decltype(s)::value;
    if (sv / CB < remaining) consume_stage(s, std::integral_constant<int, sv % CB>());
  });
#if GPN_MSPLIT_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long tr2 = wall_clock64();
#endif

  // ---- the waves' sums meet in LDS; wave q finalises column tiles q, q + SP, ... ------------------------------------------------
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) red[wave][nt][lane] = acc[nt];
  __syncthreads();
#if GPN_MSPLIT_TRACE
  const unsigned long long tr3 = wall_clock64();
#endif
  const bool st_fwd = stats.slab != nullptr && stats.x == nullptr;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int nt = wave + f * SP;
    if (nt < NT) {
      f32x4 v4 = red[0][nt][lane];
#pragma unroll
      for (int q = 1; q < SP; ++q) v4 += red[q][nt][lane];
      const uint32_t col = (uint32_t)((nt0 + nt) * 16 + i16);
      double s0 = 0.0, s1 = 0.0;
      gpn::AffineCol ac;
      if constexpr (EP) ac = gpn::affine_col(stats.ep, col);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if ((int64_t)tile * 16 + 4 * g + r < n_dst) {
          const uint32_t e = (uint32_t)orow[r] * (uint32_t)cout + col;
          float v = v4[r];
          if (accumulate) v += out[e];  // (a second gradient of the same rows, added in place)
          if constexpr (EP) v = gpn::affine_apply(stats.ep, ac, v, e);  // an inference pass's BatchNorm [+ residual] [+ ReLU]
          out[e] = v;
          if (st_fwd) {
            s0 += (double)v;
            s1 += (double)v * (double)v;
          } else if (st_bwd) {
            const float gm = (stats.relu && !(by[f][r] > 0.f)) ? 0.f : v;
            s0 += (double)gm;
            s1 += (double)gm * (double)((bx[f][r] - bmu[f]) * bis[f]);
          }
        }
      }
      if (st_fwd) gpn::stat_add<false>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
      else if (st_bwd) gpn::stat_add<true>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
    }
  }
#if GPN_MSPLIT_TRACE
  if (g_msplit_trace && blockIdx.y == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tr4 = wall_clock64();
    if (lane == 0) {
      unsigned long long* t = g_msplit_trace + ((size_t)unit * SP + wave) * 10;
      t[0] = tr0, t[1] = tr1, t[2] = tr1b, t[3] = tr2, t[4] = tr3, t[5] = tr4, t[6] = tr_taps;
      t[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      t[8] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
      t[9] = (unsigned long long)blockIdx.x;
    }
  }
#endif
  if (more) __syncthreads();  // another round: `red` is rewritten
  };
  // workgroups are dealt round-robin to the 8 XCDs: every XCD takes one contiguous eighth of the units (the rows its waves gather
  // are fetched into ONE L2)
  if constexpr (!DEV) {
    const int unit = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (unit >= n_units) return;  // (the whole workgroup)
    run_unit(unit, false);
  } else {
    const int per8 = (n_units + 7) >> 3, step = (int)(gridDim.x >> 3);
    for (int wj = (int)(blockIdx.x >> 3); wj < per8; wj += step) {
      const int unit = (int)(blockIdx.x & 7) * per8 + wj;
      if (unit < n_units) run_unit(unit, wj + step < per8);  // (uniform per workgroup; a skipped unit is the eighth's last)
    }
  }
}

// ---- selection ---------------------------------------------------------------------------------------------------------------------
// mode: 0 = off (the direct kernel of spconv_fwd.hip keeps these layers), 1 = on.  force_nt / force_sp: 0 = the table below.
std::atomic<int> g_mode{1};
std::atomic<int> g_force_nt{0}, g_force_sp{0};
// layers of at least this many column tiles take nine waves per row tile.  Never, by measurement (profiles/r06_conv_msplit_sweep.txt:
// nine waves of three taps lose to four waves of seven at every level, 27.4 against 21.7 us at 7k rows x 64 channels, 7.3 against
// 6.7 at 489 rows x 96); the nine-wave form stays instantiated for the sweep
std::atomic<int> g_sp9_from_nt{1 << 30};

#define GPN_MSPLIT_CB(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(10) X(12) X(14)

bool ms_width(int CB) {
#define GPN_X(cb) if (CB == cb) return true;
  GPN_MSPLIT_CB(GPN_X)
#undef GPN_X
  return false;
}

struct Cut {
  int nt, sp;
};
// the nine-wave form is instantiated where its ring fits the 168 registers a wave of a 576-thread workgroup can have
constexpr bool ms_fits9(int CB, int NT) { return CB * (1 + NT) <= 28; }

// Waves per row tile and column tiles per workgroup (tools/conv_msplit_sweep.py, profiles/r06_conv_msplit_sweep.txt).
// SP fixes the summation grouping, so it must NOT depend on the row count: a device-counted launch (whose host only has a plan)
// and the exactly-sized launch of the same layer have to produce the same bits.  It follows the layer's WIDTH instead: in a
// U-Net the wide layers are the deep levels - few row tiles, long chains per tap - which is where nine waves of three taps pay.
// NT does not enter the result: the widest of {4, 3, 2, 1} dividing the layer's column tiles that still leaves `want` workgroups.
Cut pick_cut(int K, int64_t n_tiles, int CB, int nt_total) {
  Cut c{1, 4};
  const int fnt = g_force_nt.load(std::memory_order_relaxed), fsp = g_force_sp.load(std::memory_order_relaxed);
  c.sp = (K >= 27 && nt_total >= g_sp9_from_nt.load(std::memory_order_relaxed)) ? 9 : 4;
  if (fsp == 4 || fsp == 9) c.sp = fsp;
  if (K < c.sp) c.sp = 4;
  const int64_t want = 384;
  for (int d = 4; d >= 1; --d)
    if (nt_total % d == 0 && n_tiles * (nt_total / d) >= want) { c.nt = d; break; }
  if (fnt > 0 && nt_total % fnt == 0 && fnt <= 4) c.nt = fnt;
  while (c.sp == 9 && !ms_fits9(CB, c.nt)) {  // (the next narrower divisor)
    int d = c.nt - 1;
    while (d > 1 && nt_total % d) --d;
    c.nt = d;
  }
  return c;
}

template <int CB, int NT, int SP>
int launch_msplit(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                  int nt_total, int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream,
                  const gpn::DevRows& rows) {
  const int n_tiles = (int)gpn::cdiv(n_dst, 16);
  const int col_groups = nt_total / NT;
  const int n_units = n_tiles * col_groups;
  const int64_t plan_units = gpn::cdiv(gpn::plan_rows(n_dst, rows), 16) * col_groups;
  const size_t packed_bytes = (size_t)K * CB * nt_total * 1024;
  // (an exactly-sized launch as FEWER workgroups that walk the units - 512 ... 1536 workgroups for the 1575 row tiles of the
  // 25k-row level, the device-counted form's loop: 29.3 - 35 us against 29.5, profiles/r06_conv_msplit_walk.txt - does not pay)
  const dim3 grid(gpn::dev_grid(n_units, plan_units, rows.dev != nullptr, 8, 512), stats.twin.in ? 2 : 1);
#define GPN_MS_LAUNCH(DEVV, EPV)                                                                                                      \
  hipLaunchKernelGGL((spconv_msplit_kernel<CB, NT, SP, DEVV, EPV>), grid, dim3(SP * 64), 0, stream, in, packed, nbr, perm, K, n_dst, \
                     n_tiles, n_units, nt_total, col_groups, packed_bytes, accumulate, stats, out, rows.dev)
  if (stats.ep.mean) {  // (an inference pass: the BatchNorm behind the conv in the epilogue)
    if (rows.dev) GPN_MS_LAUNCH(true, true);
    else GPN_MS_LAUNCH(false, true);
  } else {
    if (rows.dev) GPN_MS_LAUNCH(true, false);
    else GPN_MS_LAUNCH(false, false);
  }
#undef GPN_MS_LAUNCH
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int CB, int NT>
int dispatch_sp(int SP, const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                int nt_total, int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream, const gpn::DevRows& rows) {
  if constexpr (ms_fits9(CB, NT)) {
    if (SP == 9) return launch_msplit<CB, NT, 9>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
  }
  return launch_msplit<CB, NT, 4>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
}

template <int CB>
int dispatch_nt(const Cut& c, const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                int nt_total, int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream, const gpn::DevRows& rows) {
  switch (c.nt) {
    case 1: return dispatch_sp<CB, 1>(c.sp, in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 2: return dispatch_sp<CB, 2>(c.sp, in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 3: return dispatch_sp<CB, 3>(c.sp, in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    default: return dispatch_sp<CB, 4>(c.sp, in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
  }
}

}  // namespace

namespace gpn {

bool spconv_msplit_supported(int K, int64_t n_dst, int cin, int cout) {
  if (g_mode.load(std::memory_order_relaxed) == 0) return false;
  if (!(K == 27 || K == 8) || cin % 16 || cout % 16 || n_dst < 1) return false;
  // 32-bit byte offsets: source rows (at most 8 n_dst of them, for a stride-2 conv), output rows, the neighbour table
  if (n_dst * (int64_t)8 * std::max(cin, cout) * 4 >= ((int64_t)1 << 31) || (int64_t)K * n_dst * 4 >= ((int64_t)1 << 31)) return false;
  return ms_width(cin / 16);
}

int spconv_msplit_launch(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                         int cin, int cout, int accumulate, const ConvStats& stats, float* out, hipStream_t stream,
                         const DevRows& rows) {
  const int CB = cin / 16, nt_total = cout / 16;
  const Cut c = pick_cut(K, gpn::cdiv(gpn::plan_rows(n_dst, rows), 16), CB, nt_total);
#define GPN_X(cb) \
  if (CB == cb) return dispatch_nt<cb>(c, in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
  GPN_MSPLIT_CB(GPN_X)
#undef GPN_X
  gpn::set_error("gpn_spconv_fwd: no masked tap-split kernel for %d -> %d channels", cin, cout);
  return GPN_ERR_ARG;
}

}  // namespace gpn

#if GPN_MSPLIT_TRACE
extern "C" int gpn_probe_msplit_trace(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_msplit_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

// the masked tap-split kernel on (1, default) / off (0: the direct kernel of spconv_fwd.hip takes its layers); mode < 0 leaves it.
// force_nt / force_sp: column tiles per workgroup (1 - 4, must divide the layer's) and waves per row tile (4 or 9) for every layer
// instead of the built-in table; 0 = the table; < 0 = unchanged.  Returns the previous mode.  (Measurement / test knob.)

extern "C" int gpn_spconv_msplit(int mode, int force_nt, int force_sp) {
  const int prev = g_mode.load(std::memory_order_relaxed);
  if (mode >= 0) g_mode.store(mode ? 1 : 0, std::memory_order_relaxed);
  if (force_nt >= 0) g_force_nt.store(force_nt, std::memory_order_relaxed);
  if (force_sp >= 0) g_force_sp.store(force_sp, std::memory_order_relaxed);
  return prev;
}
