// spconv_tiles.hip — the masked-tile sparse convolution kernel (forward and dgrad launches) for gfx950 (round 3; tap loop round 4).
//
// Same arithmetic as the direct kernel of spconv_fwd.hip (output-stationary: MFMA row i of a 16-row tile IS destination row
// i for every tap, fp32 accumulators in registers, every output row written once, no atomics, v_mfma_f32_16x16x4_f32),
// different work mapping.  The direct kernel spends ~26 VALU / SALU instructions of index, offset and skip handling per
// (tile, tap, column tile) for 4 MFMAs and walks all K taps of every tile (profiles/r02_conv_ablation.txt: without loads
// and without MFMAs it still takes 60 % of its time).  Here:
//   * prologue, once per wave: the wave's column of the neighbour table for ALL K taps is read with a handful of coalesced
//     loads that are in flight together (lane = (row of the wave, tap sub-slot): 64 entries per instruction), ballots find
//     the taps any row of the wave has, and the gather offsets of exactly those taps are written - compacted, in tap order -
//     to a per-wave LDS slab.  The long-latency table reads (HBM) are thereby paid once and kept out of the main loop:
//     vmcnt counts loads in issue order, so a table read requested ahead inside the loop would hold up every younger
//     operand load of the same wave (the first version of this kernel did that: 1 us per tap).
//   * main loop over the LIVE taps only (scalar bit scan, dynamic trip count; a dead tap costs nothing): per (tap, row tile)
//     one ds_read of the offset and one v_add, per 16-channel input block R row-gathers + NT weight fragments (1 KiB from
//     L2, shared by the R row tiles) feeding R x NT x 4 MFMAs.  Round 4: the loop is a RING of operand slots - the operands
//     of 1 - 5 taps are requested ahead of the tap in the MFMAs (see "the tap loop" below); unconditional buffer loads (an
//     absent neighbour reads at an out-of-range offset: zeros, no memory access), counted s_waitcnt; what latency remains
//     is covered by the other waves of the SIMD.
//   * the rulebook's TILE ORDER (gpn_rulebook_tile_order: rows sorted by neighbour mask inside 16384-row blocks) makes the
//     rows of a tile share their taps: 2.4x (level 0) / 1.4x (levels 1, 2) the MFMA row-slots of the useful pairs instead
//     of 4.3x / 2.2x / 2.0x in voxel order; a stride-2 / inverse conv drops from 4.4x to 1.0-1.3x.
//   * a wave owns ONE row tile and NT (1..7) column tiles (the widest divisor of the layer's column tiles that still leaves
//     ~1500 waves: cols_per_wave below).  R = 2 row tiles per wave - half the waves, a weight fragment shared by both - is
//     implemented (template parameter) and measured 25-38 % slower at the 80k-row level: the kernel lives on the number of
//     waves a SIMD interleaves.
// (Round 3's first attempt at a register ring - a loop over slot indices - was undone by hipcc 7.2: requests hoisted above the
// MFMAs that still read the slot, fresh registers, s_waitcnt vmcnt(0) and a block of v_mov copies per iteration.  The ring of
// round 4 writes the slots out (compile-time slot indices) and puts a scheduling barrier after every request block.)
// What a launch waits for, measured wave by wave (tools/probes/tiles_trace.py): the most loaded SIMD's MFMAs plus
// a prologue and an epilogue that all waves go through together - not the gathers.
// Summation order per output element = the direct kernel's: ascending tap, a tap's input blocks and channels in one MFMA
// chain, the taps' sums added in fp32 (two-level); taps a row does not have add exact zeros, so the result does not depend
// on the tile order, equals the direct kernel's bit for bit and is deterministic.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "bn_stats.h"
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kMaxTaps = 27;

// x = hi + mid + lo exactly (three round-to-nearest bf16 planes of eight fp32 values): the A-operand side of a split-bf16 product
__device__ __forceinline__ void split_bf16x3(const f32x4 a0, const f32x4 a1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r1 = x[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    hi[i] = h, mid[i] = m, lo[i] = (__bf16)(r1 - (float)m);
  }
}

// compile-time measurement switches (tools/probes/tiles_ablation.sh; 0 in the product): 1 = no row gathers, 2 = no weight loads,
// 4 = no MFMAs (an element-wise stand-in keeps the loads alive), 8 = no BatchNorm sums in the epilogue, 16 = the TIMING of a
// split-bf16 contraction (round 5, profiles/r05_bf16x6.txt): every gathered fp32 row piece split on the fly into three bf16 planes
// (hi / mid / lo, round-to-nearest), six v_mfma_f32_16x16x32_bf16 per (32-channel block, column tile) against weight planes that
// are stand-ins (the fp32 fragments' bits plus a third 1 KiB fragment per block pair, so that the weight traffic is the 1.5x a
// pre-split bf16 x 3 layout would move).  Results are wrong by design; CB must be even.
#ifndef GPN_TILES_ABL
#define GPN_TILES_ABL 0
#endif
#ifndef GPN_TILES_OPERAND_REGS
#define GPN_TILES_OPERAND_REGS 32
#endif

// GPN_TILES_TRACE (tools/probes/tiles_trace.py; off in the product): every wave records when it started, finished its prologue,
// its tap loop and its epilogue (s_memrealtime, 100 MHz), its live taps and where it ran (HW_ID, XCC_ID)
#ifndef GPN_TILES_TRACE
#define GPN_TILES_TRACE 0
#endif
#if GPN_TILES_TRACE
__device__ unsigned long long* g_tiles_trace = nullptr;  // [units][8]
#endif

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  static_for_impl(f, std::make_integer_sequence<int, N>());
}

constexpr int cfg_slots(int CB, int R, int NT) {  // operand slots of the tap loop's ring: what the register budget holds, 2 .. 6
  const int regs_per_tap = CB * (R + NT) * 4 + ((GPN_TILES_ABL & 16) ? (CB / 2) * NT * 4 : 0);
  const int S = GPN_TILES_OPERAND_REGS / regs_per_tap;
  return S < 2 ? 2 : (S > 6 ? 6 : S);
}

// DEV = false: one unit per wave, the grid is exactly the launch (straight-line).  DEV = true: the row count is a device
// counter (gpn::DevRows: n_dst is the buffers' bound, the grid a guess) and the waves walk the units of their XCD's eighth.
template <int CB, int NT, int R, bool DEV, bool EP>
__global__ __launch_bounds__(256) void spconv_tiles_kernel(const float* __restrict__ in, const float* __restrict__ packed,
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
  constexpr int RW = R * 16;        // rows of a wave
  constexpr int TPI = 64 / RW;      // taps covered by one table load of the prologue
  constexpr int NI = (kMaxTaps + TPI - 1) / TPI;
  constexpr uint32_t kOob = 0x80000000u;
  __shared__ uint32_t slab[4][kMaxTaps + 1][RW];  // per wave: byte offset of the gathered row (kOob = none) by live-tap slot

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  if constexpr (DEV) {  // the row count is a device counter (gpn::DevRows): n_dst was the buffers' bound
    n_dst = gpn::live_rows(n_dev, n_dst);
    n_tiles = (int)((n_dst + 15) >> 4);
    n_units = ((n_tiles + R - 1) / R) * col_groups;
  }
  // one unit = R row tiles x NT column tiles of one wave.  (A generic lambda called from ONE place per instantiation: wrapped
  // in the grid-stride loop of the DEV form the same code took 128 instead of 62 VGPRs - half the waves per SIMD, +19 % per
  // launch at the 80k-row level - so the exactly-sized form keeps its straight-line shape.)
  auto run_unit = [&](const int unit) {
#if GPN_TILES_TRACE
  const unsigned long long tr0 = wall_clock64();
#endif
  const int rg = unit / col_groups;
  const int nt0 = (unit - rg * col_groups) * NT;
  const int tile0 = rg * R;
  constexpr int cin = CB * 16;
  const int cout = nt_total * 16;

  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbr), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)packed_bytes, 0x00020000);
  const uint32_t col_bytes = (uint32_t)n_dst * 4u;

  // ---- a dgrad launch that carries a BatchNorm's backward sums (bn_stats.h) needs that BatchNorm's x and y at the elements
  // this wave will write, and the channel statistics: requested NOW, so that they arrive during the contraction - read in
  // the epilogue they were one (perm ->) x, y round trip of pure latency at the end of every wave (4 us per launch) ----------
  const bool st_bwd = stats.slab != nullptr && stats.x != nullptr;
  int32_t orow[R][4];
  float bx[R][NT][4], by[R][NT][4], bmu[NT], bis[NT];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    const int tile = tile0 + t;
    if (perm && tile < n_tiles) {
      const int4 pv = *reinterpret_cast<const int4*>(perm + (int64_t)tile * 16 + 4 * g);
      orow[t][0] = pv.x, orow[t][1] = pv.y, orow[t][2] = pv.z, orow[t][3] = pv.w;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) orow[t][r] = tile * 16 + 4 * g + r;
    }
  }
  if (st_bwd) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint32_t col = (uint32_t)((nt0 + nt) * 16 + i16);
      bmu[nt] = stats.mean[col], bis[nt] = stats.invstd[col];
#pragma unroll
      for (int t = 0; t < R; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bx[t][nt][r] = 0.f, by[t][nt][r] = 1.f;
          if ((int64_t)(tile0 + t) * 16 + 4 * g + r < n_dst) {
            const uint32_t e = (uint32_t)orow[t][r] * (uint32_t)cout + col;
            bx[t][nt][r] = stats.x[e];
            if (stats.relu) by[t][nt][r] = stats.y[e];
          }
        }
    }
  }

  // ---- prologue: table column of the wave's rows, all taps; live taps; compacted offsets into the slab --------------------
  const int lr = lane % RW, lt = lane / RW;
  const int64_t pos = (int64_t)tile0 * 16 + lr;
  const bool row_ok = pos < n_dst;
  const uint32_t tvoff = (uint32_t)(row_ok ? pos : n_dst - 1) * 4u;
  int32_t raw[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int tap = i * TPI + lt;
    const int tc = tap < K ? tap : 0;  // (lanes past the last tap re-read tap 0 and are masked below)
    raw[i] = -1;
    if (i * TPI < K)  // (uniform: a K = 8 table needs 2-4 of the loads)
      raw[i] = __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, (int)(tvoff + (uint32_t)tc * col_bytes), 0, 0));
  }
  uint32_t um = 0;  // bit k = some row of the wave has tap k
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int tap = i * TPI + lt;
    const bool valid = row_ok && tap < K && raw[i] >= 0;
    const uint64_t b = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int s = 0; s < TPI; ++s) {
      const uint64_t sub = RW == 32 ? (b >> (32 * s)) & 0xffffffffull : (b >> (16 * s)) & 0xffffull;
      if (sub != 0 && i * TPI + s < kMaxTaps) um |= 1u << (i * TPI + s);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int tap = i * TPI + lt;
    const bool valid = row_ok && tap < K && raw[i] >= 0;
    const bool live = tap < kMaxTaps && ((um >> tap) & 1u) != 0u;
    const int slot = __builtin_popcount(um & ((1u << tap) - 1u));
    if (live) slab[wave][slot][lr] = valid ? (uint32_t)raw[i] * (uint32_t)(cin * 4) : kOob;
  }
  int remaining = __builtin_popcount(um);
#if GPN_TILES_TRACE
  const unsigned long long tr_taps = (unsigned long long)remaining;
  const unsigned long long tr1 = wall_clock64();
#endif

  f32x4 acc[R][NT];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < R; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = zero;

  const uint32_t bvoff = (uint32_t)lane * 16u;
  const uint32_t g16 = (uint32_t)g * 16u;
  // ---- the tap loop: a ring of S operand slots, S - 1 taps requested ahead of the one in the MFMAs ------------------------
  // Left to itself the scheduler sinks a tap's loads to just above their use (one tap in flight per wave; a SIMD's few waves -
  // 5 at the 80k-row level - then keep the MFMA pipe ~half busy: tools/probes/tiles_trace.py), and it undoes a ring written as
  // a loop over slots (round 3: fresh registers per request, vmcnt(0) and a block of copies per iteration).  Here the loop
  // body is the S sub-steps written out - slot indices are compile-time constants, no copies - and a scheduling barrier
  // after every request block keeps it above the MFMAs of the older slot; the waits come out as counted s_waitcnt.  Requests
  // past the wave's last live tap read at out-of-range offsets (zeros, no memory access) and are never multiplied.
  constexpr int S = cfg_slots(CB, R, NT);
  f32x4 ra[S][CB][R], rb[S][CB][NT];
#if GPN_TILES_ABL & 16
  f32x4 rbx[S][(CB + 1) / 2][NT];
#endif
  int to_issue = remaining, issued = 0;
  auto issue = [&](auto slot_tag) {
    constexpr int sl = decltype(slot_tag)::value;
    const bool has = to_issue > 0;
    const int k = has ? __builtin_ctz(um) : 0;
    um &= um - 1u;
    const int ls = issued < kMaxTaps ? issued : kMaxTaps;  // (the slab has kMaxTaps + 1 slots)
    to_issue -= 1, issued += 1;
    const uint32_t woff = has ? (uint32_t)(k * CB * nt_total + nt0) * 1024u : 0x7ffffc00u - (uint32_t)(CB * nt_total) * 1024u;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const uint32_t ao = (has ? slab[wave][ls][t * 16 + i16] : kOob) + g16;  // (kOob + g16 stays out of range)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        if constexpr ((GPN_TILES_ABL & 1) != 0) {
          const float f = __builtin_bit_cast(float, ao + (uint32_t)cb);
          ra[sl][cb][t] = f32x4{f, f, f, f};
        } else {
          ra[sl][cb][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)ao, cb * 64, 0));
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr ((GPN_TILES_ABL & 2) != 0) {
          const float f = (float)(k + cb + nt);
          rb[sl][cb][nt] = f32x4{f, f, f, f};
        } else {
          rb[sl][cb][nt] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(bvoff + woff), (cb * nt_total + nt) * 1024, 0));
        }
      }
#if GPN_TILES_ABL & 16
    {  // the third weight plane: one more 1 KiB fragment per (block pair, column tile), from another tap's part of the weights
      const int k2 = has ? (k + 1 < K ? k + 1 : 0) : 0;
      const uint32_t woff2 = has ? (uint32_t)(k2 * CB * nt_total + nt0) * 1024u : 0x7ffffc00u - (uint32_t)(CB * nt_total) * 1024u;
#pragma unroll
      for (int c2 = 0; c2 < (CB + 1) / 2; ++c2)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          rbx[sl][c2][nt] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(bvoff + woff2), (c2 * nt_total + nt) * 1024, 0));
    }
#endif
  };
  auto consume = [&](auto slot_tag) {
    constexpr int sl = decltype(slot_tag)::value;
    // two-level summation, as the direct kernel: a tap's CB * 16 products accumulate in `part` (one MFMA chain from zero),
    // the taps' sums are added to `acc` - rounding error grows with sqrt(16 CB) + sqrt(K) terms instead of sqrt(16 CB K)
    // (2e-7 instead of 6e-7 relative; BatchNorm on the small deep levels amplifies it ~1000x in backward: with one chain
    // the golden-pipeline gradient bound of 1e-3 x max|g| is missed by 3 %)
    f32x4 part[R][NT];
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) part[t][nt] = zero;
#if GPN_TILES_ABL & 16
    if constexpr (CB >= 2) {  // (an odd block count: the last pair's second half is zeros, as a padded layout would have it)
#pragma unroll
      for (int c2 = 0; c2 < (CB + 1) / 2; ++c2) {
        constexpr int kLast = CB - 1;
#pragma unroll
        for (int t = 0; t < R; ++t) {
          bf16x8 ah, am, al;
          split_bf16x3(ra[sl][2 * c2][t], 2 * c2 + 1 < CB ? ra[sl][2 * c2 + 1 < CB ? 2 * c2 + 1 : kLast][t] : zero, ah, am, al);
          // smallest products first: hi x lo, mid x mid, hi x mid, lo x hi, mid x hi, hi x hi (column tiles interleaved)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, rbx[sl][c2][nt]), part[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, __builtin_bit_cast(bf16x8, rb[sl][2 * c2 + 1 < CB ? 2 * c2 + 1 : kLast][nt]), part[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, rb[sl][2 * c2 + 1 < CB ? 2 * c2 + 1 : kLast][nt]), part[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, rb[sl][2 * c2][nt]), part[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, __builtin_bit_cast(bf16x8, rb[sl][2 * c2][nt]), part[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, rb[sl][2 * c2][nt]), part[t][nt], 0, 0, 0);
        }
      }
    } else
#endif
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      // column tiles interleaved: a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles, an independent one after 32
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr ((GPN_TILES_ABL & 4) != 0) part[t][nt][s4] += ra[sl][cb][t][s4] * rb[sl][cb][nt][s4];
            else part[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[sl][cb][t][s4], rb[sl][cb][nt][s4], part[t][nt], 0, 0, 0);
          }
    }
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[t][nt] += part[t][nt];
  };
  // (compile-time loops over the slots)
  auto for_slots = [&](auto&& f) { static_for<S>(f); };
  for_slots([&](auto i) {
    if constexpr (decltype(i)::value < S - 1) issue(i);
  });
  __builtin_amdgcn_sched_barrier(0);
  while (remaining >= S) {
    remaining -= S;
    for_slots([&](auto i) {
      constexpr int iv = decltype(i)::value;
      issue(std::integral_constant<int, (iv + S - 1) % S>());
      __builtin_amdgcn_sched_barrier(0);
      consume(i);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  for_slots([&](auto i) {  // the last remaining (< S) taps: requested already
    if (decltype(i)::value < remaining) consume(i);
  });

#if GPN_TILES_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long tr2 = wall_clock64();
#endif
  // ---- D[row = 4g + r][col = i16] of every (row tile, column tile) -> out; BatchNorm column sums of the tile (bn_stats.h) ----
  const bool st_fwd = (GPN_TILES_ABL & 8) == 0 && stats.slab != nullptr && stats.x == nullptr;
#pragma unroll
  for (int t = 0; t < R; ++t) {
    const int tile = tile0 + t;
    if (tile < n_tiles) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint32_t col = (uint32_t)((nt0 + nt) * 16 + i16);
        double s0 = 0.0, s1 = 0.0;
        gpn::AffineCol ac;
        if constexpr (EP) ac = gpn::affine_col(stats.ep, col);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if ((int64_t)tile * 16 + 4 * g + r < n_dst) {
            const uint32_t e = (uint32_t)orow[t][r] * (uint32_t)cout + col;
            float v = acc[t][nt][r];
            if (accumulate) v += out[e];  // (a second gradient of the same rows, added in place)
            if constexpr (EP) v = gpn::affine_apply(stats.ep, ac, v, e);  // an inference pass's BatchNorm [+ residual] [+ ReLU]
            out[e] = v;
            if (st_fwd) {
              s0 += (double)v;
              s1 += (double)v * (double)v;
            } else if (st_bwd) {
              const float gm = (stats.relu && !(by[t][nt][r] > 0.f)) ? 0.f : v;
              s0 += (double)gm;
              s1 += (double)gm * (double)((bx[t][nt][r] - bmu[nt]) * bis[nt]);
            }
          }
        }
        if (st_fwd) gpn::stat_add<false>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
        else if (st_bwd) gpn::stat_add<true>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
      }
    }
  }
#if GPN_TILES_TRACE
  if (g_tiles_trace && blockIdx.y == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tr3 = wall_clock64();
    if (lane == 0) {
      unsigned long long* t = g_tiles_trace + (size_t)unit * 8;
      t[0] = tr0, t[1] = tr1, t[2] = tr2, t[3] = tr3, t[4] = tr_taps;
      t[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      t[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
      t[7] = (unsigned long long)blockIdx.x;
    }
  }
#endif
  };
  if constexpr (!DEV) {
    // workgroups are dealt round-robin to the 8 XCDs: every XCD takes one contiguous eighth of the units (the rows its waves
    // gather are fetched into ONE L2)
    const int wg = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int unit = __builtin_amdgcn_readfirstlane(wg * 4 + wave);
    if (unit >= n_units) return;  // whole wave; no barrier in this kernel
    run_unit(unit);
  } else {
    // the workgroups of an XCD share one eighth of the row-tile range (its rows stay in one L2) and deal its units out
    // boustrophedon: wave j of the XCD's J waves takes the units at positions j, 2J-1-j, 2J+j, ... of the eighth - one round
    // unless the launch outgrew its plan.
    // (Round 4 tried this walk for the exactly-sized launches too - a grid of 2 - 6 waves per SIMD, with the rulebook's tiles
    // of an eighth sorted by descending live taps so that a SIMD's units are a stratified sample, and, before that, unit
    // queues on atomic counters.  tools/probes/tiles_trace.py: with one unit per wave a launch lasts as long as its most
    // loaded SIMD's MFMAs - 1.47x the mean at the 80k-row level - plus a prologue and an epilogue all waves go through
    // together; the walk cut the imbalance to 1.3x but leaves 3 - 4 waves per SIMD in the tap loop instead of 5, and a
    // level has only ~5 row tiles per SIMD to balance with: 35.5 - 38 us per launch instead of 38, 7.92 - 8.10 ms per
    // step instead of 7.94.  A same-address atomic across XCDs takes ~13 ns: the queues ran 300 - 900 us.  Removed.)
    static_assert(R == 1, "units are single row tiles");
    const int tp8 = (n_tiles + 7) >> 3;
    const int x = (int)(blockIdx.x & 7);
    const int a = min(x * tp8, n_tiles), b = min(a + tp8, n_tiles);
    const int units = (b - a) * col_groups;
    const int J = (int)(gridDim.x >> 3) * 4, j = (int)(blockIdx.x >> 3) * 4 + wave;
    for (int r = 0;; ++r) {
      const int pos = __builtin_amdgcn_readfirstlane(r * J + ((r & 1) ? J - 1 - j : j));
      if (pos >= units) break;  // (positions grow with r for every j)
      run_unit(a * col_groups + pos);
    }
  }
}

std::atomic<int64_t> g_min_tiles{4096};  // (gpn_spconv_tiles_min_tiles changes it: tests, measurements)

constexpr int kMinWaves = 1536;  // a launch takes as many column tiles per wave as still leave this many waves

template <int CB, int NT>
int launch_tiles(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                 int nt_total, int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream,
                 const gpn::DevRows& rows) {
  constexpr int R = 1;
  const int n_tiles = (int)gpn::cdiv(n_dst, 16);
  const int col_groups = nt_total / NT;
  const int n_units = (int)gpn::cdiv(n_tiles, R) * col_groups;
  const int64_t plan_units = gpn::cdiv(gpn::cdiv(gpn::plan_rows(n_dst, rows), 16), R) * col_groups;
  const size_t packed_bytes = (size_t)K * CB * nt_total * 1024;
  const dim3 grid(gpn::dev_grid(gpn::cdiv(n_units, 4), gpn::cdiv(plan_units, 4), rows.dev != nullptr, 8), stats.twin.in ? 2 : 1);
#define GPN_TILES_LAUNCH(DEVV, EPV)                                                                                                   \
  hipLaunchKernelGGL((spconv_tiles_kernel<CB, NT, R, DEVV, EPV>), grid, dim3(256), 0, stream, in, packed, nbr, perm, K, n_dst, n_tiles, \
                     n_units, nt_total, col_groups, packed_bytes, accumulate, stats, out, rows.dev)
  if (stats.ep.mean) {  // (an inference pass: the BatchNorm behind the conv in the epilogue)
    if (rows.dev) GPN_TILES_LAUNCH(true, true);
    else GPN_TILES_LAUNCH(false, true);
  } else {
    if (rows.dev) GPN_TILES_LAUNCH(true, false);
    else GPN_TILES_LAUNCH(false, false);
  }
#undef GPN_TILES_LAUNCH
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// Column tiles per wave (a divisor of the layer's nt_total, at most 7).  Measured on the bench's level shapes
// (tools/conv_tiles_bench.py, profiles/r03_conv_wave_shapes.txt): what pays is the number of WAVES first - a wave is a
// serial chain of (offset read, operand loads, MFMAs) per tap, and the SIMDs hide it only with several waves each - and
// the reuse of a gathered row across column tiles second.  Two row tiles per wave (half the waves, weight fragments shared)
// lost 25-38 % at the 80k-row level; one column tile per wave (rows re-gathered per column tile) loses as much where
// >= 1536 waves are available with more.  So: the widest divisor that leaves >= 1536 waves, else two (if that still
// gives 512 waves), else one.
int cols_per_wave(int64_t n_tiles, int nt_total) {
  for (int d = nt_total < 7 ? nt_total : 7; d >= 1; --d)
    if (nt_total % d == 0 && n_tiles * (nt_total / d) >= kMinWaves) return d;
  return (nt_total % 2 == 0 && n_tiles * (nt_total / 2) >= 512) ? 2 : 1;
}

// input widths (16-channel blocks) the kernel is instantiated for: those of a residual U-Net with channels 16 (l + 1),
// l < 7, and of its decoder convs behind the skip concats (2c -> c)
#define GPN_TILES_CB(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(10) X(12) X(14)

bool supported_width(int CB) {
#define GPN_X(cb) if (CB == cb) return true;
  GPN_TILES_CB(GPN_X)
#undef GPN_X
  return false;
}

template <int CB>
int dispatch_cols(int NT, const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                  int nt_total, int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream,
                  const gpn::DevRows& rows) {
  switch (NT) {
    case 1: return launch_tiles<CB, 1>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 2: return launch_tiles<CB, 2>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 3: return launch_tiles<CB, 3>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 4: return launch_tiles<CB, 4>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 5: return launch_tiles<CB, 5>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 6: return launch_tiles<CB, 6>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
    default: return launch_tiles<CB, 7>(in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
  }
}

}  // namespace

namespace gpn {

bool spconv_tiles_supported(int K, int64_t n_dst, int cin, int cout) {
  if (!(K >= 1 && K <= kMaxTaps) || cin % 16 || cout % 16) return false;
  // 32-bit byte offsets: source rows (at most 8 n_dst of them, for a stride-2 conv), output rows, the neighbour table
  if (n_dst * (int64_t)8 * std::max(cin, cout) * 4 >= ((int64_t)1 << 31) || (int64_t)K * n_dst * 4 >= ((int64_t)1 << 31)) return false;
  // Below ~4096 tiles a launch has too few waves for the SIMDs to hide a wave's serial chain of (offset read, operand loads,
  // MFMAs) per tap behind other waves: in the training step (cold table, operands written by the previous kernel) the
  // direct kernel with its 4-stage operand ring and 10-tap index ring per wave is faster there (25k rows: 41 vs 48 us,
  // 1.8k rows: 22 vs 28 us; profiles/r03_conv_in_situ.txt), this kernel is at 80k / 144k rows (44 vs 60 us, 26 vs 28 us).
  if (gpn::cdiv(n_dst, 16) < std::max<int64_t>(g_min_tiles.load(std::memory_order_relaxed), 16)) return false;
  return supported_width(cin / 16);
}

int spconv_tiles_launch(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int K, int64_t n_dst,
                        int cin, int cout, int accumulate, const ConvStats& stats, float* out, hipStream_t stream,
                        const DevRows& rows) {
  const int CB = cin / 16, nt_total = cout / 16;
  const int NT = cols_per_wave(gpn::cdiv(gpn::plan_rows(n_dst, rows), 16), nt_total);
#define GPN_X(cb) \
  if (CB == cb) return dispatch_cols<cb>(NT, in, packed, nbr, perm, K, n_dst, nt_total, accumulate, stats, out, stream, rows);
  GPN_TILES_CB(GPN_X)
#undef GPN_X
  gpn::set_error("gpn_spconv_fwd: no masked-tile kernel for %d -> %d channels", cin, cout);
  return GPN_ERR_ARG;
}

}  // namespace gpn

#if GPN_TILES_TRACE
extern "C" int gpn_probe_tiles_trace(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tiles_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

// smallest layer (in 16-row tiles) the masked-tile kernel takes; smaller ones run on the direct / lock-step kernels of
// spconv_fwd.hip.  min_tiles < 0 only queries.  Returns the previous value.  (Default 4096; tests and tools lower it to run the
// kernel on small inputs, or raise it past every layer to keep the kernel away.)
extern "C" int64_t gpn_spconv_tiles_min_tiles(int64_t min_tiles) {
  return min_tiles < 0 ? g_min_tiles.load(std::memory_order_relaxed) : g_min_tiles.exchange(min_tiles, std::memory_order_relaxed);
}
