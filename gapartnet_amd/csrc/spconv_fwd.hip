// spconv_fwd.hip — the fused gather-MFMA sparse convolution (forward and dgrad launches) for gfx950.
//
// Output-stationary, accumulators in registers, no scatter.  A wave owns 16 destination rows (the M dimension of
// v_mfma_f32_16x16x4_f32) and NTW 16-wide output-column tiles; MFMA row i of every tap IS destination row i of the tile,
// so the fp32 accumulators stay in registers for the whole kernel and each output row is written once.  For tap k the A
// operand row i is the gathered source row nbr[k][row0 + i] (zero where the neighbour does not exist), read straight
// from the tap-major neighbour table the rulebook builder produces — one coalesced 64-byte index read per wave and tap,
// no pair lists, no atomics, no LDS traffic for the accumulation.  (Block-compacted pair lists with an LDS scatter
// measured 3-5x slower here: with 16..112 channels the layers are bound by instruction issue and dependent-load
// latency, not by the MFMA work that zero rows waste.)
//
// Workgroup = WPB waves (4..16) walking stages (tap k, chunk of CW 16-channel blocks) in lock-step, so a stage's weight
// slab (CW x NTW pre-packed 1 KiB MFMA-B fragments) comes from L2 once per workgroup and is shared through a
// double-buffered LDS slab (conflict-free ds_read_b128), one barrier per stage.  Global loads are software-pipelined
// with compile-time ring slots (stage loop unrolled by 2D, stage count padded with empty stages):
//     neighbour indices 2D stages ahead, gathered rows and weight slabs D stages ahead.
// Every load is unconditional and branch-free (clamped indices + selects) so that the compiler can count them and emit
// partial s_waitcnt vmcnt(N); with lane-divergent guards it falls back to vmcnt(0) and every stage pays a full memory
// latency.  Lane (i = l&15, g = l>>4) loads channels [16cb+4g, 16cb+4g+4) of row i's neighbour with one 16-byte load
// (whole 64-byte pieces of each gathered row); the 4 channels feed 4 consecutive MFMA steps and the matching
// K-permutation is baked into the packed weights.  The fp32 MFMA is exact (== an fmaf chain) and the summation order
// is fixed (tap-major), so results are deterministic.  Small layers split their taps over grid.z into partial outputs
// that a fixed-order kernel sums.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NTW, int CW>
struct FwdCfg {
  static constexpr int SLAB_V4 = CW * NTW * 64;  // float4 per slab buffer: [CW][NTW][64 lanes]
  static constexpr int D = (CW * NTW >= 8) ? 2 : 4;  // prefetch distance in stages
  static constexpr size_t lds_bytes = (size_t)2 * SLAB_V4 * 16;
};

struct StageCursor {
  int k, ch;
  __device__ __forceinline__ void advance(int nch) {
    ++ch;
    if (ch == nch) { ch = 0; ++k; }
  }
};

template <int NTW, int CW, int NS>  // NS = slab float4 per thread = ceil(SLAB_V4 / blockDim.x)
__global__ void spconv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                  const int32_t* __restrict__ nbr, int K, int64_t n_dst, int cin, int nt_total,
                                  int taps_per_split, float* __restrict__ out) {
  using C = FwdCfg<NTW, CW>;
  constexpr int SLAB_V4 = C::SLAB_V4, D = C::D, E = 2 * C::D;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* slab = reinterpret_cast<f32x4*>(smem);  // [2][SLAB_V4]

  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, wpb = nthreads >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * wpb + wave;
  const int64_t row0 = tile * 16;
  const int64_t my_row = row0 + i16;
  const bool row_ok = my_row < n_dst;
  const int64_t row_c = row_ok ? my_row : (n_dst - 1);

  const int nt0 = blockIdx.y * NTW;
  const int ntw = (nt_total - nt0 < NTW) ? (nt_total - nt0) : NTW;
  const int k_lo = blockIdx.z * taps_per_split;
  const int k_hi = (k_lo + taps_per_split < K) ? (k_lo + taps_per_split) : K;
  const int cout = nt_total * 16;
  const int CB = cin >> 4;
  const int NCH = CB / CW;  // CW divides CB (host guarantees)
  const int n_stages = (k_hi - k_lo) * NCH;
  const int n_pad = (n_stages + E - 1) / E * E;
  const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(packed);

  // ---- branch-free loaders (stage cursors past the end are clamped: harmless duplicates) --------------------------
  auto load_slab = [&](StageCursor sc, f32x4 (&r)[NS]) {
    const int k = sc.k < k_hi ? sc.k : k_hi - 1;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      int q = j * nthreads + tid;
      q = q < SLAB_V4 ? q : SLAB_V4 - 1;
      const int p = q >> 6;  // piece = c * NTW + nt
      const int c = p / NTW;
      int nt = p - c * NTW;
      nt = nt < ntw ? nt : 0;
      r[j] = pw[((int64_t)(k * CB + sc.ch * CW + c) * nt_total + nt0 + nt) * 64 + (q & 63)];
    }
  };
  auto store_slab = [&](int buf, const f32x4 (&r)[NS]) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int q = j * nthreads + tid;
      if (q < SLAB_V4) slab[buf * SLAB_V4 + q] = r[j];
    }
  };
  // raw table entry; validity (stage inside the range, row inside the tensor) is applied when the value is USED, so
  // that nothing consumes the load result - and forces a wait - at issue time
  auto load_idx = [&](StageCursor sc) -> int32_t {
    const int k = sc.k < k_hi ? sc.k : k_hi - 1;
    return nbr[(int64_t)k * n_dst + row_c];
  };
  auto load_a = [&](int32_t idx, int ch, f32x4 (&a)[CW]) {
    const int32_t s = idx < 0 ? 0 : idx;  // absent neighbours gather row 0 and are zeroed before the MFMA
    const f32x4* arow = reinterpret_cast<const f32x4*>(in + (int64_t)s * cin + ch * (CW * 16) + 4 * g);
#pragma unroll
    for (int c = 0; c < CW; ++c) a[c] = arow[c * 4];
  };

  // ---- prologue: fill the rings ----------------------------------------------------------------------------------------
  StageCursor cur{k_lo, 0}, mid{k_lo, 0}, far{k_lo, 0};  // stages s, s + D, s + E
  int32_t ireg[E];
  f32x4 areg[D][CW];
  f32x4 sreg[D][NS];
#pragma unroll
  for (int u = 0; u < E; ++u) { ireg[u] = load_idx(far); far.advance(NCH); }
#pragma unroll
  for (int u = 0; u < D; ++u) { load_slab(mid, sreg[u]); load_a(ireg[u], mid.ch, areg[u]); mid.advance(NCH); }
  store_slab(0, sreg[0]);
  StageCursor slab_next = mid;  // stage D
  load_slab(slab_next, sreg[0]);
  slab_next.advance(NCH);
  __syncthreads();

  f32x4 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int s0 = 0; s0 < n_pad; s0 += E) {
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int stage = s0 + u;
      // ---- contraction of the current stage (skipped when no row of the tile has this neighbour) ----
      const int32_t idx = (stage < n_stages && row_ok) ? ireg[u] : -1;
      if (__builtin_amdgcn_ballot_w64(idx >= 0) != 0) {
        const f32x4* sb = slab + (stage & 1) * SLAB_V4 + lane;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          f32x4 a = areg[u % D][c];
          if (idx < 0) a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            if (nt < ntw) {
              const f32x4 bf = sb[(c * NTW + nt) * 64];
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bf.x, acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bf.y, acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bf.z, acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bf.w, acc[nt], 0, 0, 0);
            }
          }
        }
      }
      // ---- hand the next stage its slab, refill the ring slots this stage freed (all loads unconditional) ----
      store_slab((stage + 1) & 1, sreg[(u + 1) % D]);
      load_slab(slab_next, sreg[(u + 1) % D]);
      slab_next.advance(NCH);
      load_a(ireg[(u + D) % E], mid.ch, areg[u % D]);
      mid.advance(NCH);
      ireg[u] = load_idx(far);
      far.advance(NCH);
      cur.advance(NCH);
      __syncthreads();
    }
  }

  // ---- D[row = 4g + r][col = i16] -> out (split z writes partial z) ---------------------------------------------------
  float* outz = out + (int64_t)blockIdx.z * n_dst * cout;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + 4 * g + r;
    if (row < n_dst) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        if (nt < ntw) outz[row * cout + (nt0 + nt) * 16 + i16] = acc[nt][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent tile-streaming variant for the large levels (K = 27 or 8, >= 1024 tiles, no tap split).
// The lock-step kernel above pays a workgroup start-up (index column + slab round trips before the first MFMA) once
// per 16-row tile and one barrier per stage, and its 1024-thread workgroups leave only two slots per CU to hide that
// in: on the two largest levels it ran at 3.5-4.6x its MFMA time.  Here a workgroup
// (8 waves, one or two per CU) is persistent: it stages the slab of an input block once and every wave streams
// through many tiles with the rings carried ACROSS tile boundaries - while tile i is contracted, the index column
// of tile i+1 refills the slots tile i has consumed and its first gathers are already in flight, so a wave never
// restarts cold.  For cin > 16 the input blocks are the outer loop (slab staged CB times per workgroup, no barrier
// inside a pass); block 0 writes the partial output rows and later blocks add to them (fixed order: deterministic).
// Tiles are dealt so that each XCD owns a contiguous range of rows and neighbouring tiles run at the same time on
// the same XCD (they gather largely the same source rows: L2 hits).
template <int NTW, int KT, int D, int CW>  // CW = 16-channel input blocks contracted per pass (slab = KT x CW x NTW KiB)
__global__ __launch_bounds__(512) void spconv_fwd_stream_kernel(const float* __restrict__ in,
                                                                const float* __restrict__ packed,
                                                                const int32_t* __restrict__ nbr, int64_t n_dst, int cin,
                                                                int nt_total, int64_t tiles, int cb_per_split, const int32_t* __restrict__ perm,
                                                                float* __restrict__ out_base) {
  static_assert(KT % D == 0, "ring slots are compile-time: the gather distance must divide the tap count");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* slab = reinterpret_cast<f32x4*>(smem);  // [KT][CW][NTW][64 lanes]
  constexpr int WPB = 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;

  // tile ownership: XCD x (workgroup ids go round-robin over the 8 XCDs) owns tiles [tile_lo, tile_hi); inside it
  // wave slot j takes tiles tile_lo + j, + waves_per_xcd, ...
  const int xcd = blockIdx.x & 7;
  const int waves_per_xcd = (gridDim.x >> 3) * WPB;
  const int slot = (blockIdx.x >> 3) * WPB + wave;
  const int64_t tiles_per_xcd = (tiles + 7) >> 3;
  const int64_t tile_lo = xcd * tiles_per_xcd;
  const int64_t tile_hi = tile_lo + tiles_per_xcd < tiles ? tile_lo + tiles_per_xcd : tiles;
  const int64_t first = tile_lo + slot;
  const int m = first < tile_hi ? (int)((tile_hi - first + waves_per_xcd - 1) / waves_per_xcd) : 0;

  const int nt0 = blockIdx.y * NTW;
  const int ntw = (nt_total - nt0 < NTW) ? (nt_total - nt0) : NTW;
  const int cout = nt_total * 16;
  const int CB = cin >> 4;
  const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(packed);
  // mid-size levels (about one tile per wave) split the input blocks over grid.z: each slice is a single pass writing
  // its own partial output (summed afterwards in slice order) instead of CB sequential passes with their start-ups
  const int cb_lo = blockIdx.z * cb_per_split;
  const int cb_hi = (cb_lo + cb_per_split < CB) ? (cb_lo + cb_per_split) : CB;
  float* __restrict__ out = out_base + (int64_t)blockIdx.z * n_dst * cout;

  // Gathers and index reads are buffer loads: wave-uniform descriptor + 32-bit per-lane byte offset (one VGPR per
  // pending load, no per-lane 64-bit arithmetic), and an absent neighbour becomes an out-of-range offset, which the
  // hardware answers with zeros WITHOUT touching memory.  (The host guarantees every byte offset < 2^31.)
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbr), 0, 0x7fffffff, 0x00020000);
  const uint32_t col_bytes = (uint32_t)n_dst * 4u;  // one tap's column of the neighbour table
  auto clamp_row = [&](int64_t tile) -> uint32_t {
    const int64_t r = tile * 16 + i16;
    return (uint32_t)(r < n_dst ? r : n_dst - 1);
  };
  auto load_idx = [&](int u, uint32_t row) -> int32_t {
    return __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, (int)(row * 4u), (int)(u * col_bytes), 0));
  };
  // the CW 64-byte pieces of one gathered row; absent neighbours and rows past the end of the tensor read as zeros
  // (out-of-range offset), so the MFMA operands need no select - a VALU op between two MFMAs of a dependent chain
  // costs ~40 cycles on this core
  auto load_a = [&](int32_t idx, bool row_valid, int cb, f32x4 (&a)[CW]) {
    const uint32_t off = (idx < 0 || !row_valid) ? 0x80000000u : ((uint32_t)idx * (uint32_t)cin + 4u * (uint32_t)g) * 4u;
#pragma unroll
    for (int c = 0; c < CW; ++c)
      a[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)off, (cb + c) * 64, 0));
  };

  for (int cb = cb_lo; cb < cb_hi; cb += CW) {  // CW divides CB and cb_per_split (host)
    if (cb > cb_lo) __syncthreads();  // every wave has finished its pass over the previous blocks' slab
    for (int q = tid; q < KT * CW * NTW * 64; q += WPB * 64) {
      const int t = q / (CW * NTW * 64);
      const int rem = q - t * (CW * NTW * 64);
      const int c = rem / (NTW * 64);
      int nt = (rem - c * (NTW * 64)) >> 6;
      nt = nt < ntw ? nt : 0;
      slab[q] = pw[((int64_t)(t * CB + cb + c) * nt_total + nt0 + nt) * 64 + (rem & 63)];
    }
    // rings of the wave's first tile (their round trip overlaps the slab's)
    int32_t ireg[KT];
    f32x4 areg[D][CW];
    if (m > 0) {
      const uint32_t rc = clamp_row(first);
#pragma unroll
      for (int u = 0; u < KT; ++u) ireg[u] = load_idx(u, rc);
      const bool ok_first = first * 16 + i16 < n_dst;
#pragma unroll
      for (int u = 0; u < D; ++u) load_a(ireg[u], ok_first, cb, areg[u]);
    }
    __syncthreads();

    for (int i = 0; i < m; ++i) {
      const int64_t cur = first + (int64_t)i * waves_per_xcd;
      const int64_t nxt = i + 1 < m ? cur + waves_per_xcd : cur;  // last tile: refills become harmless duplicates
      const int64_t row0 = cur * 16;
      const bool row_ok = row0 + i16 < n_dst;
      const bool row_ok_next = nxt * 16 + i16 < n_dst;
      const uint32_t rc_next = clamp_row(nxt);

      // destination rows of this lane's accumulator fragment: tile positions row0 + 4g + r, or the rows the rulebook's
      // tile order maps them to (`nbr` is then the table in that order; perm is padded past n_dst)
      int32_t orow[4];
      if (perm) {
        const int4 pv = *reinterpret_cast<const int4*>(perm + row0 + 4 * g);
        orow[0] = pv.x, orow[1] = pv.y, orow[2] = pv.z, orow[3] = pv.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + 4 * g + r;
          orow[r] = (int32_t)(row < n_dst ? row : n_dst - 1);
        }
      }
      f32x4 acc[NTW], prev[NTW];
      f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};  // second accumulator of the NTW == 1 case
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}, prev[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (cb > cb_lo) {  // partial sums of the earlier input blocks (issued now, consumed at the end of the tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t rr = (uint32_t)orow[r];
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            prev[nt][r] = out[rr * (uint32_t)cout + (uint32_t)((nt0 + (nt < ntw ? nt : 0)) * 16 + i16)];
        }
      }

#pragma unroll
      for (int u = 0; u < KT; ++u) {
        // wave-uniform skip of taps no row of the tile has (64-bit compare mask straight into a scalar branch)
        if (__builtin_amdgcn_sicmp(ireg[u], -1, 38 /* ICMP_SGT */) != 0) {
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            const f32x4 a = areg[u % D][c];
            const f32x4* sb = slab + (u * CW + c) * (NTW * 64) + lane;
            f32x4 bf[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bf[nt] = sb[nt * 64];
            // the four k-steps of a tap stay a dependent chain per accumulator (interleaving column tiles measured 1.6x
            // slower); with a single column tile and a single block two accumulators alternate (40-cycle dependent latency
            // vs 32-cycle issue)
            if (NTW == 1 && CW == 1) {
              acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bf[0].x, acc[0], 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bf[0].y, acc2, 0, 0, 0);
              acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bf[0].z, acc[0], 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bf[0].w, acc2, 0, 0, 0);
            } else {
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt) {
                if (nt < ntw) {
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bf[nt].x, acc[nt], 0, 0, 0);
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bf[nt].y, acc[nt], 0, 0, 0);
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bf[nt].z, acc[nt], 0, 0, 0);
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bf[nt].w, acc[nt], 0, 0, 0);
                }
              }
            }
          }
        }
        asm volatile("" ::: "memory");  // keep the unrolled taps in program order
        // slot u now belongs to the next tile; the gather D taps ahead uses this tile's column while u + D < KT and
        // the next tile's (refilled KT - D taps ago) after that - the same expression either way
        ireg[u] = load_idx(u, rc_next);
        load_a(ireg[(u + D) % KT], (u + D < KT) ? row_ok : row_ok_next, cb, areg[u % D]);
      }

#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * g + r;
        if (row < n_dst) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            if (nt < ntw)
              out[(uint32_t)orow[r] * (uint32_t)cout + (uint32_t)((nt0 + nt) * 16 + i16)] =
                  (NTW == 1 && CW == 1 ? acc[nt][r] + acc2[r] : acc[nt][r]) + prev[nt][r];
        }
      }
    }
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, int splits, int64_t elems4,
                                       float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= elems4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(partial);
  f32x4 acc = p[t];
  int s = 1;
  for (; s + 3 < splits; s += 4) {  // four slices in flight, added in slice order
    const f32x4 v0 = p[(int64_t)s * elems4 + t], v1 = p[(int64_t)(s + 1) * elems4 + t];
    const f32x4 v2 = p[(int64_t)(s + 2) * elems4 + t], v3 = p[(int64_t)(s + 3) * elems4 + t];
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; s < splits; ++s) acc += p[(int64_t)s * elems4 + t];
  reinterpret_cast<f32x4*>(out)[t] = acc;
}

struct FwdPlan {
  int ntw, cw, wpb, splits, taps_per_split;
};

FwdPlan plan_fwd(int K, int64_t n_dst, int cin, int cout) {
  const int nt = cout / 16, CB = cin / 16;
  const int64_t tiles = gpn::cdiv(n_dst, 16);
  FwdPlan p;
  p.cw = (CB % 4 == 0) ? 4 : (CB % 2 == 0) ? 2 : 1;
  // waves per workgroup: more waves share one weight slab (less L2 traffic) as long as >= 512 workgroups remain
  p.wpb = 4;
  if (tiles / 16 >= 512) p.wpb = 16;
  else if (tiles / 8 >= 512) p.wpb = 8;
  const int64_t row_wgs = gpn::cdiv(tiles, p.wpb);
  // column tiles per wave: as many as possible (rows are re-gathered once per column group) with >= 512 workgroups
  p.ntw = 1;
  for (int ntw = 4; ntw > 1; --ntw) {
    if (ntw > nt) continue;
    if (row_wgs * gpn::cdiv(nt, ntw) >= 512) { p.ntw = ntw; break; }
  }
  const int64_t wgs = row_wgs * gpn::cdiv(nt, p.ntw);
  // small layers: split the taps across workgroups (partials + fixed-order reduce) until ~512 workgroups exist
  p.splits = 1;
  if (wgs < 256 && K > 1) {
    int64_t s = gpn::cdiv(512, wgs);
    if (s > K) s = K;
    p.splits = (int)s;
  }
  p.taps_per_split = (int)gpn::cdiv(K, p.splits);
  p.splits = (int)gpn::cdiv(K, p.taps_per_split);
  return p;
}

template <int NTW, int CW, int NS>
int launch_fwd(const FwdPlan& p, const float* in, const float* packed, const int32_t* nbr, int K, int64_t n_dst,
               int cin, int nt_total, float* out, hipStream_t stream) {
  const int64_t tiles = gpn::cdiv(n_dst, 16);
  const dim3 grid((unsigned)gpn::cdiv(tiles, p.wpb), (unsigned)gpn::cdiv(nt_total, NTW), (unsigned)p.splits);
  const size_t lds = FwdCfg<NTW, CW>::lds_bytes;
  hipLaunchKernelGGL((spconv_fwd_kernel<NTW, CW, NS>), grid, dim3(p.wpb * 64), lds, stream, in, packed, nbr, K, n_dst,
                     cin, nt_total, p.taps_per_split, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int NTW, int CW>
int dispatch_ns(const FwdPlan& p, const float* in, const float* packed, const int32_t* nbr, int K, int64_t n_dst,
                int cin, int nt_total, float* out, hipStream_t stream) {
  constexpr int SLAB_V4 = FwdCfg<NTW, CW>::SLAB_V4;
  const int ns = (int)gpn::cdiv(SLAB_V4, p.wpb * 64);
  switch (ns) {
    case 1: return launch_fwd<NTW, CW, 1>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
    case 2: return launch_fwd<NTW, CW, 2>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
    case 3: return launch_fwd<NTW, CW, 3>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
    default: return launch_fwd<NTW, CW, 4>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
  }
}

template <int NTW>
int dispatch_cw(const FwdPlan& p, const float* in, const float* packed, const int32_t* nbr, int K, int64_t n_dst,
                int cin, int nt_total, float* out, hipStream_t stream) {
  switch (p.cw) {
    case 1: return dispatch_ns<NTW, 1>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
    case 2: return dispatch_ns<NTW, 2>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
    default: return dispatch_ns<NTW, 4>(p, in, packed, nbr, K, n_dst, cin, nt_total, out, stream);
  }
}

// the persistent tile-streaming kernel serves the large levels: K = 27 / 8 and enough tiles that no tap split is needed
struct StreamPlan {
  bool use;
  int ntw, groups;
  int cw, cb_splits, cb_per_split;
};

StreamPlan plan_stream(int K, int64_t n_dst, int cin, int cout) {
  const int nt = cout / 16;
  StreamPlan p;
  p.groups = (int)gpn::cdiv(nt, 4);  // up to 4 output-column tiles per workgroup (slab = K x ntw KiB of LDS)
  p.ntw = (int)gpn::cdiv(nt, p.groups);
  // the kernel addresses with 32-bit byte offsets (< 2^31): source rows (at most 8 n_dst of them, for a stride-2 conv),
  // output rows and the neighbour table must fit
  p.use = (K == 27 || K == 8) && gpn::cdiv(n_dst, 16) >= 1024 &&
          n_dst * (int64_t)8 * std::max(cin, cout) * 4 < ((int64_t)1 << 31) && (int64_t)K * n_dst * 4 < ((int64_t)1 << 31);
  // two input blocks per pass when the slab allows it (K x 2 x ntw KiB <= 108): a 32-channel level is then one pass
  const int CB = cin / 16;
  p.cw = (CB % 2 == 0 && p.ntw <= 2) ? 2 : 1;
  // fewer than ~2 tiles per wave slot: the passes over the input blocks run side by side (grid.z) into partial outputs
  const int passes = CB / p.cw;
  p.cb_splits = (gpn::cdiv(n_dst, 16) < 8192 && passes > 1) ? passes : 1;
  p.cb_per_split = (passes / p.cb_splits) * p.cw;
  return p;
}

template <int NTW, int KT, int D, int CW>
int launch_stream(const StreamPlan& sp, const float* in, const float* packed, const int32_t* nbr, const int32_t* perm,
                  int64_t n_dst, int cin, int nt_total, float* out, hipStream_t stream) {
  constexpr size_t lds = (size_t)KT * CW * NTW * 64 * 16;
  // persistent grid = (workgroups that are resident at once) x CUs: asked from the runtime once per instantiation AND
  // device (a process may drive several devices; occupancy and CU count are properties of the device)
  static std::atomic<int> wgs_of_device[gpn::kMaxDevices] = {};
  int dev = 0;
  GPN_CHECK_HIP(hipGetDevice(&dev));
  GPN_CHECK_ARG(dev >= 0 && dev < gpn::kMaxDevices);
  int wgs = wgs_of_device[dev].load(std::memory_order_acquire);
  if (wgs == 0) {
    GPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_fwd_stream_kernel<NTW, KT, D, CW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0, cus = 0;
    GPN_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spconv_fwd_stream_kernel<NTW, KT, D, CW>, 512, lds));
    GPN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    wgs = (cus * per_cu + 7) / 8 * 8;
    wgs_of_device[dev].store(wgs, std::memory_order_release);  // idempotent: racing threads compute the same value
  }
  const dim3 grid((unsigned)wgs, (unsigned)sp.groups, (unsigned)sp.cb_splits);
  hipLaunchKernelGGL((spconv_fwd_stream_kernel<NTW, KT, D, CW>), grid, dim3(512), lds, stream, in, packed, nbr, n_dst, cin,
                     nt_total, gpn::cdiv(n_dst, 16), sp.cb_per_split, perm, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int NTW>
int dispatch_stream(const StreamPlan& sp, const float* in, const float* packed, const int32_t* nbr, const int32_t* perm,
                    int K, int64_t n_dst, int cin, int nt_total, float* out, hipStream_t stream) {
  constexpr int CW2 = NTW <= 2 ? 2 : 1;  // (instantiated only where the slab fits)
  if (K == 27) {
    if (sp.cw == 2) return launch_stream<NTW, 27, 3, CW2>(sp, in, packed, nbr, perm, n_dst, cin, nt_total, out, stream);
    return launch_stream<NTW, 27, 9, 1>(sp, in, packed, nbr, perm, n_dst, cin, nt_total, out, stream);
  }
  if (sp.cw == 2) return launch_stream<NTW, 8, 4, CW2>(sp, in, packed, nbr, perm, n_dst, cin, nt_total, out, stream);
  return launch_stream<NTW, 8, 8, 1>(sp, in, packed, nbr, perm, n_dst, cin, nt_total, out, stream);
}


// ------------------------------------------------------------------------------------------------------------------
// "Direct" variant for the mid-size levels (a few thousand tiles, 32..64 channels): no LDS, no barrier, no persistence.
// A wave owns ONE 16-row tile and ONE 16-wide output-column tile and walks all (tap, input block) stages with its
// accumulator in registers; the weight fragment of a stage (1 KiB, the MFMA B operand) comes straight from L2 with one
// coalesced 16-byte load per lane, D stages ahead, next to the gathered rows.  The streaming kernel above stages all
// taps' weights of an input block in LDS (81 KiB at 48 channels => one workgroup per CU, input blocks as separate passes
// with partial outputs); at these sizes there are only ~1-3 tiles per resident wave slot to amortise that over.  Here
// every (tile, column tile) pair is its own wave - 4 425 waves for a 23 k-row, 48-channel level - and a SIMD interleaves
// 4-5 of them, which is what hides the gather latency.  The waves of one tile are adjacent (they gather the same rows:
// L1 / L2 hits).  Same arithmetic and summation order per output element as the other variants (tap-major, then input
// block, then channel).
#ifndef GPN_DIRECT_D
#define GPN_DIRECT_D 4
#endif
template <int KT, int CB>
__global__ __launch_bounds__(256) void spconv_fwd_direct_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                                const int32_t* __restrict__ nbr, int64_t n_dst, int nt_total,
                                                                int64_t units, size_t packed_bytes,
                                                                const int32_t* __restrict__ perm, float* __restrict__ out) {
  constexpr int S = KT * CB;
  constexpr int D = S >= GPN_DIRECT_D ? GPN_DIRECT_D : S;  // prefetch depth in stages
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  if (unit >= units) return;  // whole wave; no barrier in this kernel
  const int64_t tile = unit / nt_total;
  const int nt = (int)(unit - tile * nt_total);
  const int cin = CB * 16, cout = nt_total * 16;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbr), 0, 0x7fffffff, 0x00020000);
  // weight fragments through a buffer descriptor as well: a stage whose tap no row of the tile has (wave-uniform) reads at
  // an out-of-range offset - zeros, and no memory access
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)packed_bytes, 0x00020000);
  const uint32_t col_bytes = (uint32_t)n_dst * 4u;
  const int64_t row0 = tile * 16;
  const bool row_ok = row0 + i16 < n_dst;
  const uint32_t rc = (uint32_t)(row_ok ? row0 + i16 : n_dst - 1);

  // the tile's column of the neighbour table, DI taps ahead of the contraction (a ring instead of all KT entries: 92 -> ~75
  // VGPRs, one more wave per SIMD)
  constexpr int DI = KT < 10 ? KT : 10;
  static_assert(DI == KT || DI >= D + 2, "the index of a stage issued D stages ahead must already be in the ring");
  int32_t ireg[DI];
  auto load_idx = [&](int tap) -> int32_t {
    return __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, (int)(rc * 4u), (int)(tap * col_bytes), 0));
  };
#pragma unroll
  for (int u = 0; u < DI; ++u) ireg[u] = load_idx(u);
  f32x4 areg[D], breg[D];
  auto issue = [&](int s, int slot) {  // s = tap * CB + cb: compile-time after unrolling
    const int tap = s / CB, cb = s - tap * CB;
    const int32_t idx = row_ok ? ireg[tap % DI] : -1;
    const bool live = __builtin_amdgcn_sicmp(idx, -1, 38 /* ICMP_SGT */) != 0;  // wave-uniform: some row of the tile has the tap
    const uint32_t off = idx < 0 ? 0x80000000u : ((uint32_t)idx * (uint32_t)cin + 4u * (uint32_t)g) * 4u;
    areg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)off, cb * 64, 0));
    const uint32_t boff = live ? (uint32_t)((s * nt_total + nt) * 1024 + lane * 16) : 0x80000000u;
    breg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)boff, 0, 0));
  };
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s, s);

  int32_t orow[4];
  if (perm) {
    const int4 pv = *reinterpret_cast<const int4*>(perm + row0 + 4 * g);
    orow[0] = pv.x, orow[1] = pv.y, orow[2] = pv.z, orow[3] = pv.w;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * g + r;
      orow[r] = (int32_t)(row < n_dst ? row : n_dst - 1);
    }
  }
  // two-level summation: a tap's CB * 16 products accumulate in `part` (one MFMA chain), the taps' partial sums are
  // added to `acc` - rounding error grows with sqrt(16 CB) + sqrt(KT) terms instead of sqrt(16 CB KT) (measured against a
  // float64 evaluation: 2e-7 relative instead of 6e-7; BatchNorm on the tiny deep levels amplifies it ~1000x in backward)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < KT; ++tap) {
    if (__builtin_amdgcn_sicmp(row_ok ? ireg[tap % DI] : -1, -1, 38 /* ICMP_SGT */) != 0) {  // some row of the tile has this tap
      f32x4 part = zero;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int s = tap * CB + cb;
        const f32x4 a = areg[s % D], b = breg[s % D];
        if (s + D < S) issue(s + D, s % D);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, part, 0, 0, 0);
      }
      acc += part;
    } else {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {  // nothing to contract (the operands of these stages were read as zeros): keep the ring moving
        const int s = tap * CB + cb;
        if (s + D < S) issue(s + D, s % D);
      }
    }
    if (tap + DI < KT) ireg[tap % DI] = load_idx(tap + DI);
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + 4 * g + r;
    if (row < n_dst) out[(uint32_t)orow[r] * (uint32_t)cout + (uint32_t)(nt * 16 + i16)] = acc[r];
  }
}

template <int KT, int CB>
int launch_direct(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int64_t n_dst, int nt_total,
                  float* out, hipStream_t stream) {
  const int64_t units = gpn::cdiv(n_dst, 16) * nt_total;
  const size_t packed_bytes = (size_t)KT * CB * nt_total * 1024;
  hipLaunchKernelGGL((spconv_fwd_direct_kernel<KT, CB>), dim3((unsigned)gpn::cdiv(units, 4)), dim3(256), 0, stream, in, packed,
                     nbr, n_dst, nt_total, units, packed_bytes, perm, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// which shapes take the direct variant.  Measured on the bench shapes (8 x 20k-point scenes; tools/kernel_rooflines.py, us per
// launch, streaming / lock-step kernel -> direct): L0 16->16 144k rows 32.5 -> 27, L1 32->32 80k rows 71.6 -> 56,
// L2 48->48 25k rows 74.0 -> 42; conv family per training step 5.23 -> 4.1 ms.  Below 16 tiles (the two deepest levels)
// a layer has too few (tile, column) units to fill the chip and the tap-split lock-step kernel stays in use; so do the
// k = 1 layers and input widths the unrolled stage loop is not instantiated for.
bool use_direct(int K, int64_t n_dst, int cin, int cout) {
  static const bool disabled = getenv("GPN_CONV_NO_DIRECT") != nullptr;  // A/B switch for measurements
  if (disabled) return false;
  const int CB = cin / 16;
  if (!(K == 27 || K == 8)) return false;
  if (!(CB >= 1 && (CB <= 8 || CB == 10 || CB == 12))) return false;
  // 32-bit byte offsets: source rows (at most 8 n_dst of them, for a stride-2 conv), output rows, the neighbour table
  if (n_dst * (int64_t)8 * std::max(cin, cout) * 4 >= ((int64_t)1 << 31) || (int64_t)K * n_dst * 4 >= ((int64_t)1 << 31)) return false;
  return gpn::cdiv(n_dst, 16) >= 16;
}

template <int KT>
int dispatch_direct(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int64_t n_dst, int cin,
                    int nt_total, float* out, hipStream_t stream) {
  switch (cin / 16) {
    case 1: return launch_direct<KT, 1>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 2: return launch_direct<KT, 2>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 3: return launch_direct<KT, 3>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 4: return launch_direct<KT, 4>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 5: return launch_direct<KT, 5>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 6: return launch_direct<KT, 6>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 7: return launch_direct<KT, 7>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 8: return launch_direct<KT, 8>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    case 10: return launch_direct<KT, 10>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
    default: return launch_direct<KT, 12>(in, packed, nbr, perm, n_dst, nt_total, out, stream);
  }
}

}  // namespace

extern "C" size_t gpn_spconv_fwd_ws_bytes(int K, int64_t n_dst, int cin, int cout) {
  if (n_dst <= 0 || cin < 16 || cout < 16) return 0;
  {
    const StreamPlan sp = plan_stream(K, n_dst, cin, cout);
    if (sp.use) return sp.cb_splits > 1 ? gpn::align_up((size_t)sp.cb_splits * n_dst * cout * sizeof(float)) : 0;
  }
  const FwdPlan p = plan_fwd(K, n_dst, cin, cout);
  return p.splits > 1 ? gpn::align_up((size_t)p.splits * n_dst * cout * sizeof(float)) : 0;
}

extern "C" int gpn_spconv_fwd_ordered(const float* in, const float* packed_w, const int32_t* nbr, const int32_t* nbr_p,
                                      const int32_t* perm, int K, int64_t n_dst, int cin, int cout, float* out,
                                      void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG((nbr_p == nullptr) == (perm == nullptr));
  GPN_CHECK_ARG(K >= 1 && n_dst >= 0);
  GPN_CHECK_ARG(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0);
  if (n_dst == 0) return GPN_OK;
  GPN_CHECK_ARG(in && packed_w && nbr && out);
  const int nt = cout / 16;
  if (use_direct(K, n_dst, cin, cout)) {
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout);
    const int32_t* table = nbr_p ? nbr_p : nbr;
    return K == 27 ? dispatch_direct<27>(in, packed_w, table, perm, n_dst, cin, nt, out, stream)
                   : dispatch_direct<8>(in, packed_w, table, perm, n_dst, cin, nt, out, stream);
  }
  const StreamPlan sp = plan_stream(K, n_dst, cin, cout);
  if (sp.use) {
    float* target = out;
    if (sp.cb_splits > 1) {
      if (!ws || ws_bytes < (size_t)sp.cb_splits * n_dst * cout * sizeof(float)) {
        gpn::set_error("gpn_spconv_fwd: workspace too small for %d input-block slices", sp.cb_splits);
        return GPN_ERR_WS;
      }
      target = static_cast<float*>(ws);
    }
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout);
    int rc;
    switch (sp.ntw) {
      case 1: rc = dispatch_stream<1>(sp, in, packed_w, nbr_p ? nbr_p : nbr, perm, K, n_dst, cin, nt, target, stream); break;
      case 2: rc = dispatch_stream<2>(sp, in, packed_w, nbr_p ? nbr_p : nbr, perm, K, n_dst, cin, nt, target, stream); break;
      case 3: rc = dispatch_stream<3>(sp, in, packed_w, nbr_p ? nbr_p : nbr, perm, K, n_dst, cin, nt, target, stream); break;
      default: rc = dispatch_stream<4>(sp, in, packed_w, nbr_p ? nbr_p : nbr, perm, K, n_dst, cin, nt, target, stream); break;
    }
    if (rc == GPN_OK && sp.cb_splits > 1) {
      const int64_t elems4 = n_dst * cout / 4;
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)gpn::cdiv(elems4, 256)), dim3(256), 0, stream, target,
                         sp.cb_splits, elems4, out);
      hipError_t e_ = hipGetLastError();
      if (e_ != hipSuccess) { gpn::set_error("gpn_spconv_fwd: reduce launch failed: %s", hipGetErrorString(e_)); rc = GPN_ERR_HIP; }
    }
    return rc;
  }
  const FwdPlan p = plan_fwd(K, n_dst, cin, cout);
  float* target = out;
  if (p.splits > 1) {
    if (!ws || ws_bytes < (size_t)p.splits * n_dst * cout * sizeof(float)) {
      gpn::set_error("gpn_spconv_fwd: workspace too small for %d tap splits", p.splits);
      return GPN_ERR_WS;
    }
    target = static_cast<float*>(ws);
  }
  int rc;
  {
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout);
    switch (p.ntw) {
      case 1: rc = dispatch_cw<1>(p, in, packed_w, nbr, K, n_dst, cin, nt, target, stream); break;
      case 2: rc = dispatch_cw<2>(p, in, packed_w, nbr, K, n_dst, cin, nt, target, stream); break;
      case 3: rc = dispatch_cw<3>(p, in, packed_w, nbr, K, n_dst, cin, nt, target, stream); break;
      default: rc = dispatch_cw<4>(p, in, packed_w, nbr, K, n_dst, cin, nt, target, stream); break;
    }
    if (rc == GPN_OK && p.splits > 1) {
      const int64_t elems4 = n_dst * cout / 4;
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)gpn::cdiv(elems4, 256)), dim3(256), 0, stream, target,
                         p.splits, elems4, out);
      hipError_t e_ = hipGetLastError();
      if (e_ != hipSuccess) { gpn::set_error("gpn_spconv_fwd: reduce launch failed: %s", hipGetErrorString(e_)); rc = GPN_ERR_HIP; }
    }
  }
  return rc;
}

extern "C" int gpn_spconv_fwd(const float* in, const float* packed_w, const int32_t* nbr, int K, int64_t n_dst,
                              int cin, int cout, float* out, void* ws, size_t ws_bytes, gpn_stream_t stream) {
  return gpn_spconv_fwd_ordered(in, packed_w, nbr, nullptr, nullptr, K, n_dst, cin, cout, out, ws, ws_bytes, stream);
}

// One-call form used by the host wrapper: packs the weight (canonical or parameter layout, optional transpose /
// tap reversal: see gpn_spconv_pack_weights) into the head of the workspace and runs the conv.  Halves the number of
// host->library calls per layer; the packed copy never needs its own allocation.
extern "C" size_t gpn_spconv_fwd_w_ws_bytes(int K, int64_t n_dst, int cin, int cout) {
  return gpn::align_up((size_t)K * cin * cout * sizeof(float)) + gpn_spconv_fwd_ws_bytes(K, n_dst, cin, cout);
}

extern "C" int gpn_spconv_fwd_w(const float* in, const float* W, int K, int cin_w, int cout_w, int pack_flags,
                                const int32_t* nbr, int64_t n_dst, float* out, void* ws, size_t ws_bytes,
                                gpn_stream_t stream) {
  GPN_CHECK_ARG(W && K >= 1 && cin_w >= 16 && cout_w >= 16);
  const int cin = (pack_flags & GPN_PACK_TRANSPOSE) ? cout_w : cin_w;
  const int cout = (pack_flags & GPN_PACK_TRANSPOSE) ? cin_w : cout_w;
  const size_t packed_bytes = gpn::align_up((size_t)K * cin * cout * sizeof(float));
  if (!ws || ws_bytes < packed_bytes + gpn_spconv_fwd_ws_bytes(K, n_dst, cin, cout)) {
    gpn::set_error("gpn_spconv_fwd_w: workspace too small");
    return GPN_ERR_WS;
  }
  float* packed = static_cast<float*>(ws);
  int rc = gpn_spconv_pack_weights(W, K, cin_w, cout_w, pack_flags, packed, stream);
  if (rc != GPN_OK) return rc;
  return gpn_spconv_fwd(in, packed, nbr, K, n_dst, cin, cout, out, static_cast<char*>(ws) + packed_bytes,
                        ws_bytes - packed_bytes, stream);
}
