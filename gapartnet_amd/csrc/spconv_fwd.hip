// spconv_fwd.hip — the fused gather-MFMA sparse convolution (forward and dgrad launches) for gfx950.
//
// Output-stationary, accumulators in registers, no scatter: a wave owns 16 destination rows (the M dimension of
// v_mfma_f32_16x16x4_f32); MFMA row i of every tap IS destination row i of the tile, so the fp32 accumulators stay in
// registers for the whole kernel and each output row is written once.  For tap k the A operand row i is the gathered
// source row nbr[k][row0 + i] (zero where the neighbour does not exist), read straight from the tap-major neighbour table
// the rulebook builder produces - no pair lists, no atomics.  Lane (i = l&15, g = l>>4) loads channels [16cb+4g, 16cb+4g+4)
// of row i's neighbour with one 16-byte load (whole 64-byte pieces of each gathered row); the 4 channels feed 4
// consecutive MFMA steps and the matching K-permutation is baked into the packed weights.  The fp32 MFMA is exact and
// the summation order is fixed, so results are deterministic.
//
// Three kernels here (layers of >= 4096 row tiles take the masked-tile kernel of spconv_tiles.hip instead, round 3):
//  * spconv_fwd_direct_kernel (below, second half of the file) - k = 27 / 8 / 1 layers with 16 .. 4095 row tiles; with fewer
//    than 12000 (tile, column tile) units in its tap-split form spconv_fwd_split_kernel (4 waves per unit, partial sums added
//    through LDS in wave order).  The direct kernel: one wave per (16-row tile, 16-column tile), weight fragments straight from L2, no LDS, no
//    barrier, no partial outputs.  It replaced round 1's persistent LDS-slab streaming kernel (L0 / L1 / L2 launches
//    32 / 72 / 74 us -> 27 / 56 / 42 us; that kernel, its tile-order rulebook option and its input-block partial sums are gone).
//  * spconv_fwd_kernel - the deepest levels (< 16 tiles) and widths the direct kernel is not instantiated for: WPB waves (4..16) walk stages (tap k, chunk of
//    CW 16-channel blocks) in lock-step, a stage's weight slab (CW x NTW 1 KiB MFMA-B fragments) goes through a
//    double-buffered LDS slab, one barrier per stage; loads are software-pipelined with compile-time ring slots and are
//    unconditional / branch-free so that the compiler emits counted s_waitcnt vmcnt(N); tiny layers split their taps over
//    grid.z into partial outputs that a fixed-order kernel sums.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "bn_stats.h"
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
                                  int taps_per_split, int accumulate, float* __restrict__ out) {
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
        if (nt < ntw) {
          float* o = outz + row * cout + (nt0 + nt) * 16 + i16;
          *o = accumulate ? *o + acc[nt][r] : acc[nt][r];  // (accumulate: a second gradient of the same rows, added in place)
        }
    }
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, int splits, int64_t elems4, int accumulate,
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
  f32x4* o = reinterpret_cast<f32x4*>(out) + t;
  *o = accumulate ? *o + acc : acc;
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
               int cin, int nt_total, int accumulate, float* out, hipStream_t stream) {
  const int64_t tiles = gpn::cdiv(n_dst, 16);
  const dim3 grid((unsigned)gpn::cdiv(tiles, p.wpb), (unsigned)gpn::cdiv(nt_total, NTW), (unsigned)p.splits);
  const size_t lds = FwdCfg<NTW, CW>::lds_bytes;
  hipLaunchKernelGGL((spconv_fwd_kernel<NTW, CW, NS>), grid, dim3(p.wpb * 64), lds, stream, in, packed, nbr, K, n_dst,
                     cin, nt_total, p.taps_per_split, accumulate, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int NTW, int CW>
int dispatch_ns(const FwdPlan& p, const float* in, const float* packed, const int32_t* nbr, int K, int64_t n_dst,
                int cin, int nt_total, int accumulate, float* out, hipStream_t stream) {
  constexpr int SLAB_V4 = FwdCfg<NTW, CW>::SLAB_V4;
  const int ns = (int)gpn::cdiv(SLAB_V4, p.wpb * 64);
  switch (ns) {
    case 1: return launch_fwd<NTW, CW, 1>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
    case 2: return launch_fwd<NTW, CW, 2>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
    case 3: return launch_fwd<NTW, CW, 3>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
    default: return launch_fwd<NTW, CW, 4>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
  }
}

template <int NTW>
int dispatch_cw(const FwdPlan& p, const float* in, const float* packed, const int32_t* nbr, int K, int64_t n_dst,
                int cin, int nt_total, int accumulate, float* out, hipStream_t stream) {
  switch (p.cw) {
    case 1: return dispatch_ns<NTW, 1>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
    case 2: return dispatch_ns<NTW, 2>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
    default: return dispatch_ns<NTW, 4>(p, in, packed, nbr, K, n_dst, cin, nt_total, accumulate, out, stream);
  }
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
// GPN_ABL (default 0) builds measurement variants of the direct kernel for tools/conv_ablation.sh: 1 = no MFMA, 2 = no
// gather / weight loads, 3 = no loads at all (synthetic neighbour pattern), 4 = 3 without MFMA.  Results are wrong by design.
#ifndef GPN_ABL
#define GPN_ABL 0
#endif
#ifndef GPN_DIRECT_D
#define GPN_DIRECT_D 4
#endif
// EP: an inference pass's BatchNorm in the epilogue (gpn::ConvAffine; instantiated for the k = 1 layers, which have no other kernel)
template <int KT, int CB, bool DEV, bool EP>  // DEV: the row count is a device counter (gpn::DevRows), units walked with a grid stride
__global__ __launch_bounds__(256) void spconv_fwd_direct_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                                const int32_t* __restrict__ nbr, int64_t n_dst, int nt_total,
                                                                int64_t units, size_t packed_bytes,
                                                                const int32_t* __restrict__ perm, int accumulate,
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
  constexpr int S = KT * CB;
  constexpr int D = S >= GPN_DIRECT_D ? GPN_DIRECT_D : S;  // prefetch depth in stages
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  if constexpr (DEV) {  // the row count is a device counter (gpn::DevRows): n_dst was the buffers' bound, the grid a guess
    n_dst = gpn::live_rows(n_dev, n_dst);
    units = ((n_dst + 15) >> 4) * nt_total;
  }
  // (the unit's code as a lambda with ONE call site per instantiation: the exactly-sized form keeps its straight-line shape and
  // register count, the DEV form wraps it in a grid-stride loop - see spconv_tiles.hip)
  auto run_unit = [&](const int64_t unit) {
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
#if GPN_ABL >= 3
    return (int32_t)((rc * 2654435761u + (uint32_t)tap * 40503u) % (5u * (uint32_t)n_dst)) < (int32_t)n_dst ? (int32_t)rc : -1;
#else
    return __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, (int)(rc * 4u), (int)(tap * col_bytes), 0));
#endif
  };
#pragma unroll
  for (int u = 0; u < DI; ++u) ireg[u] = load_idx(u);
  f32x4 areg[D], breg[D];
  auto issue = [&](int s, int slot) {  // s = tap * CB + cb: compile-time after unrolling
    const int tap = s / CB, cb = s - tap * CB;
    const int32_t idx = row_ok ? ireg[tap % DI] : -1;
    const bool live = __builtin_amdgcn_sicmp(idx, -1, 38 /* ICMP_SGT */) != 0;  // wave-uniform: some row of the tile has the tap
    const uint32_t off = idx < 0 ? 0x80000000u : ((uint32_t)idx * (uint32_t)cin + 4u * (uint32_t)g) * 4u;
    const uint32_t boff = live ? (uint32_t)((s * nt_total + nt) * 1024 + lane * 16) : 0x80000000u;
#if GPN_ABL >= 2
    areg[slot] = (f32x4){(float)off, 1.f, 2.f, 3.f};
    breg[slot] = (f32x4){(float)boff, 1.f, 2.f, 3.f};
#else
    areg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)off, cb * 64, 0));
    breg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)boff, 0, 0));
#endif
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
  // a dgrad launch carrying a BatchNorm's backward sums (bn_stats.h): that BatchNorm's x, y at this wave's output elements and
  // the channel statistics are requested now and arrive during the contraction (read in the epilogue they were a round trip
  // of pure latency at the end of every wave)
  const bool st_bwd = stats.slab != nullptr && stats.x != nullptr;
  const uint32_t col = (uint32_t)(nt * 16 + i16);
  float bx[4] = {0.f, 0.f, 0.f, 0.f}, by[4] = {1.f, 1.f, 1.f, 1.f}, mu = 0.f, is = 1.f;
  if (st_bwd) {
    mu = stats.mean[col], is = stats.invstd[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (row0 + 4 * g + r < n_dst) {
        const uint32_t e = (uint32_t)orow[r] * (uint32_t)cout + col;
        bx[r] = stats.x[e];
        if (stats.relu) by[r] = stats.y[e];
      }
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
#if GPN_ABL == 1 || GPN_ABL == 4
        part += a * b;
#else
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, part, 0, 0, 0);
#endif
      }
      // (keep this add in the block of the MFMAs: moved behind the join of the two paths, hipcc 7.2 does not insert the wait
      // states an accumulator read needs after an MFMA and the add reads stale registers - seen as wrong rows, fixed by s_nop)
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
  // store; BatchNorm column sums of the tile when the launch carries a slab (bn_stats.h)
  const bool st_fwd = stats.slab != nullptr && stats.x == nullptr;
  double s0 = 0.0, s1 = 0.0;
  gpn::AffineCol ac;
  if constexpr (EP) ac = gpn::affine_col(stats.ep, col);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + 4 * g + r;
    if (row < n_dst) {
      const uint32_t e = (uint32_t)orow[r] * (uint32_t)cout + col;
      float v = acc[r];
      if (accumulate) v += out[e];  // (a second gradient of the same rows, added in place)
      if constexpr (EP) v = gpn::affine_apply(stats.ep, ac, v, e);  // an inference pass's BatchNorm [+ residual] [+ ReLU]
      out[e] = v;
      if (st_fwd) {
        s0 += (double)v;
        s1 += (double)v * (double)v;
      } else if (st_bwd) {
        const float gm = (stats.relu && !(by[r] > 0.f)) ? 0.f : v;
        s0 += (double)gm;
        s1 += (double)gm * (double)((bx[r] - mu) * is);
      }
    }
  }
  if (st_fwd) gpn::stat_add<false>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
  else if (st_bwd) gpn::stat_add<true>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
  };
  // workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give every XCD one contiguous eighth of the
  // tiles, so that the source rows its waves gather (spatial neighbours = nearby rows) are fetched into ONE L2 instead of
  // all eight (time per launch unchanged; fetched bytes per launch, averaged over the bench's conv launches: 16.2 -> 8.9 MB)
  if constexpr (!DEV) {
    const int64_t wg = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t unit = wg * 4 + wave;
    if (unit >= units) return;  // whole wave; no barrier in this kernel
    run_unit(unit);
  } else {  // an XCD's workgroups walk its eighth with a grid stride (one round unless the launch outgrew its plan)
    const int64_t per8 = (((units + 3) >> 2) + 7) >> 3;
    for (int64_t wj = blockIdx.x >> 3; wj < per8; wj += gridDim.x >> 3) {
      const int64_t unit = ((int64_t)(blockIdx.x & 7) * per8 + wj) * 4 + wave;
      if (unit < units) run_unit(unit);
    }
  }
}

// Tap-split form of the direct kernel for layers with FEW (tile, column tile) units (round 3).  A level of a few thousand
// rows gives a launch of the direct kernel fewer waves than the chip has SIMDs (1.8k rows x 80 channels: 575), and each wave
// then is one serial chain of KT x CB dependent (gather, weight fragment, 4 MFMA) stages - 17k cycles of matrix pipe for
// that wave alone while three quarters of the SIMDs idle.  Here the SP (2 or 4) waves of a unit each walk a contiguous
// SP-th of the taps with the same register rings, leave their accumulators in LDS, and the unit's first wave adds them in
// wave order and stores (plus the BatchNorm sums): SP x the waves, chains 1 / SP as long, one launch, fixed summation order
// (per wave: taps ascending, two-level as in the direct kernel; then (((p0 + p1) + p2) + p3)) => deterministic.
// GPN_SPLIT_TRACE (tools/probes/msplit_trace.py; off in the product): every wave of the tap-split kernel records when it started, had
// its table entries, finished its stage loop, passed the barrier and finished (s_memrealtime, 100 MHz), and where it ran
#ifndef GPN_SPLIT_TRACE
#define GPN_SPLIT_TRACE 0
#endif
#if GPN_SPLIT_TRACE
__device__ unsigned long long* g_split_trace = nullptr;  // [waves][10]
#endif
template <int KT, int CB, int SP, bool DEV>
__global__ __launch_bounds__(256) void spconv_fwd_split_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                               const int32_t* __restrict__ nbr, int64_t n_dst, int nt_total,
                                                               int64_t units, size_t packed_bytes,
                                                               const int32_t* __restrict__ perm, int accumulate,
                                                               gpn::ConvStats stats, float* __restrict__ out,
                                                               const int64_t* __restrict__ n_dev) {
  constexpr int TP = (KT + SP - 1) / SP;  // taps per wave
  constexpr int S = TP * CB;              // stages per wave
  constexpr int D = S >= GPN_DIRECT_D ? GPN_DIRECT_D : S;
  constexpr int UPW = 4 / SP;             // units per workgroup
  __shared__ f32x4 red[4][64];
  if (blockIdx.y) {  // the launch's second problem (gpn::ConvTwin)
    in = stats.twin.in, packed = stats.twin.packed, out = stats.twin.out;
    stats.slab = stats.twin.slab, stats.x = stats.twin.x, stats.y = stats.twin.y, stats.mean = stats.twin.mean,
    stats.invstd = stats.twin.invstd;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  if constexpr (DEV) {  // the row count is a device counter (gpn::DevRows): n_dst was the buffers' bound, the grid a guess
    n_dst = gpn::live_rows(n_dev, n_dst);
    units = ((n_dst + 15) >> 4) * nt_total;
  }
  // (one workgroup's units as a lambda with ONE call site per instantiation, see the direct kernel; `more` = another round
  // follows, uniform per workgroup)
  auto run_wg = [&](const int64_t wg, const bool more) {
#if GPN_SPLIT_TRACE
  const unsigned long long tr0 = wall_clock64();
#endif
  const int64_t unit = wg * UPW + wave / SP;
  const int part = wave % SP;
  const int tap0 = part * TP;
  const bool active = unit < units;  // (no early return: every wave reaches the barrier below)
  const int64_t tile = active ? unit / nt_total : 0;
  const int nt = active ? (int)(unit - tile * nt_total) : 0;
  const int cin = CB * 16, cout = nt_total * 16;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(nbr), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)packed_bytes, 0x00020000);
  const uint32_t col_bytes = (uint32_t)n_dst * 4u;
  const int64_t row0 = tile * 16;
  const bool row_ok = active && row0 + i16 < n_dst;
  const uint32_t rc = (uint32_t)(row_ok ? row0 + i16 : n_dst - 1);

  constexpr int DI = TP < 10 ? TP : 10;
  static_assert(DI == TP || DI >= D + 2, "the index of a stage issued D stages ahead must already be in the ring");
  int32_t ireg[DI];
  auto load_idx = [&](int j) -> int32_t {  // local tap j = global tap tap0 + j (past the last tap: tap KT - 1 re-read, masked below)
    const int tap = tap0 + j < KT ? tap0 + j : KT - 1;
    return __builtin_bit_cast(int32_t, __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, (int)(rc * 4u), (int)((uint32_t)tap * col_bytes), 0));
  };
  auto tap_idx = [&](int j) -> int32_t { return (row_ok && tap0 + j < KT) ? ireg[j % DI] : -1; };
#pragma unroll
  for (int u = 0; u < DI; ++u) ireg[u] = load_idx(u);
#if GPN_SPLIT_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long tr1 = wall_clock64();
#endif
  f32x4 areg[D], breg[D];
  auto issue = [&](int s, int slot) {  // s = local tap * CB + cb: compile-time after unrolling
    const int j = s / CB, cb = s - j * CB;
    const int32_t idx = tap_idx(j);
    const bool live = __builtin_amdgcn_sicmp(idx, -1, 38 /* ICMP_SGT */) != 0;  // wave-uniform: some row of the tile has the tap
    const uint32_t off = idx < 0 ? 0x80000000u : ((uint32_t)idx * (uint32_t)cin + 4u * (uint32_t)g) * 4u;
    const int tap = tap0 + j < KT ? tap0 + j : KT - 1;
    const uint32_t boff = live ? (uint32_t)(((tap * CB + cb) * nt_total + nt) * 1024 + lane * 16) : 0x80000000u;
    areg[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)off, cb * 64, 0));
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
  // a dgrad launch carrying a BatchNorm's backward sums (bn_stats.h): that BatchNorm's x, y at this wave's output elements and
  // the channel statistics are requested now and arrive during the contraction (read in the epilogue they were a round trip
  // of pure latency at the end of every wave)
  const bool st_bwd = stats.slab != nullptr && stats.x != nullptr;
  const uint32_t col = (uint32_t)(nt * 16 + i16);
  float bx[4] = {0.f, 0.f, 0.f, 0.f}, by[4] = {1.f, 1.f, 1.f, 1.f}, mu = 0.f, is = 1.f;
  if (st_bwd && part == 0 && active) {
    mu = stats.mean[col], is = stats.invstd[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (row0 + 4 * g + r < n_dst) {
        const uint32_t e = (uint32_t)orow[r] * (uint32_t)cout + col;
        bx[r] = stats.x[e];
        if (stats.relu) by[r] = stats.y[e];
      }
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    if (__builtin_amdgcn_sicmp(tap_idx(j), -1, 38 /* ICMP_SGT */) != 0) {  // some row of the tile has this tap
      f32x4 part_sum = zero;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int s = j * CB + cb;
        const f32x4 a = areg[s % D], b = breg[s % D];
        if (s + D < S) issue(s + D, s % D);
        part_sum = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, part_sum, 0, 0, 0);
        part_sum = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, part_sum, 0, 0, 0);
        part_sum = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, part_sum, 0, 0, 0);
        part_sum = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, part_sum, 0, 0, 0);
      }
      acc += part_sum;  // (in the MFMAs' block: see the compiler trap noted at the direct kernel)
    } else {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int s = j * CB + cb;
        if (s + D < S) issue(s + D, s % D);
      }
    }
    if (j + DI < TP) ireg[j % DI] = load_idx(j + DI);
    asm volatile("" ::: "memory");
  }
#if GPN_SPLIT_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long tr2 = wall_clock64();
#endif
  // the unit's partial sums -> its first wave, added in wave order
  if (part != 0) red[wave][lane] = acc;
  __syncthreads();
#if GPN_SPLIT_TRACE
  const unsigned long long tr3 = wall_clock64();
#endif
  if (part == 0 && active) {
#pragma unroll
  for (int q = 1; q < SP; ++q) acc += red[wave + q][lane];

  const bool st_fwd = stats.slab != nullptr && stats.x == nullptr;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + 4 * g + r;
    if (row < n_dst) {
      const uint32_t e = (uint32_t)orow[r] * (uint32_t)cout + col;
      float v = acc[r];
      if (accumulate) v += out[e];
      out[e] = v;
      if (st_fwd) {
        s0 += (double)v;
        s1 += (double)v * (double)v;
      } else if (st_bwd) {
        const float gm = (stats.relu && !(by[r] > 0.f)) ? 0.f : v;
        s0 += (double)gm;
        s1 += (double)gm * (double)((bx[r] - mu) * is);
      }
    }
  }
  if (st_fwd) gpn::stat_add<false>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
  else if (st_bwd) gpn::stat_add<true>(stats.slab, cout, (int)(tile & stats.slot_mask), (int)col, g, gpn::stat_reduce_g(s0), gpn::stat_reduce_g(s1));
  }
#if GPN_SPLIT_TRACE
  if (g_split_trace && blockIdx.y == 0 && active) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tr4 = wall_clock64();
    if (lane == 0) {
      unsigned long long* t = g_split_trace + ((size_t)unit * SP + part) * 10;
      t[0] = tr0, t[1] = tr1, t[2] = tr1, t[3] = tr2, t[4] = tr3, t[5] = tr4, t[6] = (unsigned long long)TP;
      t[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      t[8] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
      t[9] = (unsigned long long)blockIdx.x;
    }
  }
#endif
  if (more) __syncthreads();  // another round: `red` is rewritten
  };
  if constexpr (!DEV) {
    run_wg((int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3), false);
  } else {
    const int64_t per8 = (((units + UPW - 1) / UPW) + 7) >> 3;  // workgroups of an XCD that have a unit
    for (int64_t wj = blockIdx.x >> 3; wj < per8; wj += gridDim.x >> 3)
      run_wg((int64_t)(blockIdx.x & 7) * per8 + wj, wj + (int64_t)(gridDim.x >> 3) < per8);
  }
}

// units below which a layer takes the 4-way / 2-way tap-split form (0 = never).  tools/conv_split_sweep.py
// (profiles/r03_conv_split.txt, us per launch unsplit / 2-way / 4-way): 487 rows 192->96 48 / 18 / 11, 1.8k rows 160->80
// 67 / 28 / 22, 6.9k rows 64->64 23.2 / 23.4 / 21.1, 25k rows 96->48 75 / 75 / 71, 80k rows 32->32 58.8 / 59.0 / 57.5: the
// 4-way form never loses, the 2-way form never wins over it => 4-way for every layer the direct kernel takes.
// gpn_spconv_direct_split() changes the thresholds (tests, measurements).
constexpr int64_t kSplit4Units = 12000, kSplit2Units = 0;
std::atomic<int64_t> g_split4_units{kSplit4Units};
std::atomic<int64_t> g_split2_units{kSplit2Units};

template <int KT, int CB, int SP>
int launch_split(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int64_t n_dst, int nt_total,
                 int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream, const gpn::DevRows& rows) {
  if (stats.ep.mean) {
    gpn::set_error("gpn_spconv_fwd: the direct kernel's tap-split form has no BatchNorm epilogue");
    return GPN_ERR_ARG;
  }
  const int64_t units = gpn::cdiv(n_dst, 16) * nt_total;
  const int64_t plan_units = gpn::cdiv(gpn::plan_rows(n_dst, rows), 16) * nt_total;
  const size_t packed_bytes = (size_t)KT * CB * nt_total * 1024;
  const dim3 grid(gpn::dev_grid(gpn::cdiv(units, 4 / SP), gpn::cdiv(plan_units, 4 / SP), rows.dev != nullptr, 8), stats.twin.in ? 2 : 1);
  if (rows.dev)
    hipLaunchKernelGGL((spconv_fwd_split_kernel<KT, CB, SP, true>), grid, dim3(256), 0, stream, in, packed, nbr, n_dst, nt_total, units,
                       packed_bytes, perm, accumulate, stats, out, rows.dev);
  else
    hipLaunchKernelGGL((spconv_fwd_split_kernel<KT, CB, SP, false>), grid, dim3(256), 0, stream, in, packed, nbr, n_dst, nt_total, units,
                       packed_bytes, perm, accumulate, stats, out, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int KT, int CB>
int launch_direct(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int64_t n_dst, int nt_total,
                  int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream, const gpn::DevRows& rows) {
  const int64_t units = gpn::cdiv(n_dst, 16) * nt_total;
  const int64_t plan_units = gpn::cdiv(gpn::plan_rows(n_dst, rows), 16) * nt_total;  // (the form is picked from the planned count)
  if constexpr (KT >= 8) {  // (a k = 1 layer has no taps to split)
    if (plan_units < g_split4_units.load(std::memory_order_relaxed))
      return launch_split<KT, CB, 4>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    if (plan_units < g_split2_units.load(std::memory_order_relaxed))
      return launch_split<KT, CB, 2>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
  }
  const size_t packed_bytes = (size_t)KT * CB * nt_total * 1024;
  const dim3 grid(gpn::dev_grid(gpn::cdiv(units, 4), gpn::cdiv(plan_units, 4), rows.dev != nullptr, 8), stats.twin.in ? 2 : 1);
  if constexpr (KT == 1) {
    if (stats.ep.mean) {  // (an inference pass: the BatchNorm behind the k = 1 conv in the epilogue)
      if (rows.dev)
        hipLaunchKernelGGL((spconv_fwd_direct_kernel<KT, CB, true, true>), grid, dim3(256), 0, stream, in, packed, nbr, n_dst, nt_total, units,
                           packed_bytes, perm, accumulate, stats, out, rows.dev);
      else
        hipLaunchKernelGGL((spconv_fwd_direct_kernel<KT, CB, false, true>), grid, dim3(256), 0, stream, in, packed, nbr, n_dst, nt_total, units,
                           packed_bytes, perm, accumulate, stats, out, rows.dev);
      GPN_CHECK_LAUNCH();
      return GPN_OK;
    }
  }
  if (stats.ep.mean) {
    gpn::set_error("gpn_spconv_fwd: the direct kernel applies a BatchNorm in its epilogue for k = 1 layers only");
    return GPN_ERR_ARG;
  }
  if (rows.dev)
    hipLaunchKernelGGL((spconv_fwd_direct_kernel<KT, CB, true, false>), grid, dim3(256), 0, stream, in, packed, nbr, n_dst, nt_total, units,
                       packed_bytes, perm, accumulate, stats, out, rows.dev);
  else
    hipLaunchKernelGGL((spconv_fwd_direct_kernel<KT, CB, false, false>), grid, dim3(256), 0, stream, in, packed, nbr, n_dst, nt_total, units,
                       packed_bytes, perm, accumulate, stats, out, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// which shapes take the direct variant.  Measured on the bench shapes (8 x 20k-point scenes; tools/kernel_rooflines.py, us per
// launch, streaming / lock-step kernel -> direct): L0 16->16 144k rows 32.5 -> 27, L1 32->32 80k rows 71.6 -> 56,
// L2 48->48 25k rows 74.0 -> 42; conv family per training step 5.23 -> 4.1 ms.  Below 16 tiles (the two deepest levels)
// a layer has too few (tile, column) units to fill the chip and the tap-split lock-step kernel stays in use; so do the
// k = 1 layers and input widths the unrolled stage loop is not instantiated for.
bool use_direct(int K, int64_t n_dst, int cin, int cout) {
  const int CB = cin / 16;
  // k = 1 layers (the residual blocks' shortcut convs, linear heads) take it too since round 3: its epilogue carries the
  // BatchNorm sums (bn_stats.h), the lock-step kernel's does not
  if (!(K == 27 || K == 8 || K == 1)) return false;
  if (!(CB >= 1 && (CB <= 8 || CB == 10 || CB == 12))) return false;
  // 32-bit byte offsets: source rows (at most 8 n_dst of them, for a stride-2 conv), output rows, the neighbour table
  if (n_dst * (int64_t)8 * std::max(cin, cout) * 4 >= ((int64_t)1 << 31) || (int64_t)K * n_dst * 4 >= ((int64_t)1 << 31)) return false;
  return gpn::cdiv(n_dst, 16) >= 16;
}

template <int KT>
int dispatch_direct(const float* in, const float* packed, const int32_t* nbr, const int32_t* perm, int64_t n_dst, int cin,
                    int nt_total, int accumulate, const gpn::ConvStats& stats, float* out, hipStream_t stream,
                    const gpn::DevRows& rows) {
  switch (cin / 16) {
    case 1: return launch_direct<KT, 1>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 2: return launch_direct<KT, 2>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 3: return launch_direct<KT, 3>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 4: return launch_direct<KT, 4>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 5: return launch_direct<KT, 5>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 6: return launch_direct<KT, 6>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 7: return launch_direct<KT, 7>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 8: return launch_direct<KT, 8>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    case 10: return launch_direct<KT, 10>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
    default: return launch_direct<KT, 12>(in, packed, nbr, perm, n_dst, nt_total, accumulate, stats, out, stream, rows);
  }
}

}  // namespace

// (tile, column tile) unit counts below which the direct kernel runs in its 4-way / 2-way tap-split form; a negative
// argument leaves that threshold unchanged; 0 disables the form.  Results of the split forms differ from the unsplit kernel
// in the order of the last additions only (both deterministic).
extern "C" int gpn_spconv_direct_split(int64_t split4_below_units, int64_t split2_below_units) {
  if (split4_below_units >= 0) g_split4_units.store(split4_below_units, std::memory_order_relaxed);
  if (split2_below_units >= 0) g_split2_units.store(split2_below_units, std::memory_order_relaxed);
  return GPN_OK;
}

#if GPN_SPLIT_TRACE
extern "C" int gpn_probe_split_trace(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_split_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t gpn_spconv_fwd_ws_bytes(int K, int64_t n_dst, int cin, int cout) {
  if (n_dst <= 0 || cin < 16 || cout < 16) return 0;
  if (gpn::spconv_tiles_supported(K, n_dst, cin, cout) || gpn::spconv_msplit_supported(K, n_dst, cin, cout) || use_direct(K, n_dst, cin, cout))
    return 0;
  const FwdPlan p = plan_fwd(K, n_dst, cin, cout);
  return p.splits > 1 ? gpn::align_up((size_t)p.splits * n_dst * cout * sizeof(float)) : 0;
}

extern "C" int gpn_spconv_fwd_ordered(const float* in, const float* packed_w, const int32_t* nbr, const int32_t* nbr_p,
                                      const int32_t* perm, int K, int64_t n_dst, int cin, int cout, float* out,
                                      void* ws, size_t ws_bytes, gpn_stream_t stream) {
  return gpn::spconv_fwd_into(in, packed_w, nbr, nbr_p, perm, K, n_dst, cin, cout, out, 0, gpn::ConvStats(), ws, ws_bytes,
                              (hipStream_t)stream);
}

// out = conv (accumulate == 0) or out += conv (the network executor's second gradient of a slot: same value as staging the
// conv's result and adding it with a separate launch, which is what it replaces)
bool gpn::spconv_fwd_accumulates_stats(int K, int64_t n_dst, int cin, int cout) {
  return n_dst > 0 && (gpn::spconv_tiles_supported(K, n_dst, cin, cout) || gpn::spconv_msplit_supported(K, n_dst, cin, cout) ||
                       use_direct(K, n_dst, cin, cout));
}

// true if a conv of this shape runs on a kernel whose epilogue can apply a gpn::ConvAffine (an inference pass's BatchNorm): the
// masked-tile and the masked tap-split kernel, and the direct kernel for k = 1
bool gpn::spconv_fwd_applies_affine(int K, int64_t n_dst, int cin, int cout, const gpn::DevRows& rows) {
  if (n_dst <= 0) return false;
  const bool tiles = rows.dev ? (gpn::spconv_tiles_supported(K, n_dst, cin, cout) && gpn::spconv_tiles_supported(K, gpn::plan_rows(n_dst, rows), cin, cout))
                              : gpn::spconv_tiles_supported(K, n_dst, cin, cout);
  if (tiles || gpn::spconv_msplit_supported(K, n_dst, cin, cout)) return true;
  return K == 1 && (rows.dev ? use_direct(K, std::max<int64_t>(n_dst, 256), cin, cout) : use_direct(K, n_dst, cin, cout));
}

// which kernel a layer takes when its row count is a device counter: decided from the host's plan (the layer must fit the
// 32-bit offsets at its bound), never the lock-step kernel - that one sizes partial outputs from the row count
static bool dev_rows_take_tiles(int K, int64_t n_bound, int64_t n_plan, int cin, int cout) {
  return gpn::spconv_tiles_supported(K, n_bound, cin, cout) && gpn::spconv_tiles_supported(K, n_plan, cin, cout);
}
static bool dev_rows_take_direct(int K, int64_t n_bound, int cin, int cout) { return use_direct(K, std::max<int64_t>(n_bound, 256), cin, cout); }

bool gpn::spconv_fwd_accumulates_stats(int K, int64_t n_dst, int cin, int cout, const gpn::DevRows& rows) {
  if (!rows.dev) return gpn::spconv_fwd_accumulates_stats(K, n_dst, cin, cout);
  return n_dst > 0 && (dev_rows_take_tiles(K, n_dst, gpn::plan_rows(n_dst, rows), cin, cout) ||
                       gpn::spconv_msplit_supported(K, n_dst, cin, cout) || dev_rows_take_direct(K, n_dst, cin, cout));
}

int gpn::spconv_fwd_into(const float* in, const float* packed_w, const int32_t* nbr, const int32_t* nbr_p, const int32_t* perm,
                         int K, int64_t n_dst, int cin, int cout, float* out, int accumulate, const gpn::ConvStats& stats,
                         void* ws, size_t ws_bytes, hipStream_t stream, const gpn::DevRows& rows) {
  GPN_CHECK_ARG((nbr_p == nullptr) == (perm == nullptr));
  GPN_CHECK_ARG(K >= 1 && n_dst >= 0);
  GPN_CHECK_ARG(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0);
  if (n_dst == 0) return GPN_OK;
  GPN_CHECK_ARG(in && packed_w && nbr && out);
  GPN_CHECK_ARG(!stats.twin.in || (stats.twin.packed && stats.twin.out && (stats.slab == nullptr) == (stats.twin.slab == nullptr)));
  const int nt = cout / 16;
  const int64_t n_plan = gpn::plan_rows(n_dst, rows);
  const bool tiles = rows.dev ? dev_rows_take_tiles(K, n_dst, n_plan, cin, cout) : gpn::spconv_tiles_supported(K, n_dst, cin, cout);
  if (tiles) {  // the masked-tile kernel (spconv_tiles.hip): every layer of >= 16 tiles
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout, gpn::prof_shape_tag(K, n_dst, cin, cout, stats.twin.in != nullptr));
    return gpn::spconv_tiles_launch(in, packed_w, nbr_p ? nbr_p : nbr, perm, K, n_dst, cin, cout, accumulate, stats, out, stream, rows);
  }
  if (gpn::spconv_msplit_supported(K, n_dst, cin, cout)) {  // the masked tap-split kernel (spconv_msplit.hip): every k = 27 / 8 layer below that
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout, gpn::prof_shape_tag(K, n_dst, cin, cout, stats.twin.in != nullptr));
    return gpn::spconv_msplit_launch(in, packed_w, nbr_p ? nbr_p : nbr, perm, K, n_dst, cin, cout, accumulate, stats, out, stream, rows);
  }
  if (rows.dev ? dev_rows_take_direct(K, n_dst, cin, cout) : use_direct(K, n_dst, cin, cout)) {
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout, gpn::prof_shape_tag(K, n_dst, cin, cout, stats.twin.in != nullptr));
    const int32_t* table = nbr_p ? nbr_p : nbr;
    return K == 27 ? dispatch_direct<27>(in, packed_w, table, perm, n_dst, cin, nt, accumulate, stats, out, stream, rows)
           : K == 8 ? dispatch_direct<8>(in, packed_w, table, perm, n_dst, cin, nt, accumulate, stats, out, stream, rows)
                    : dispatch_direct<1>(in, packed_w, table, perm, n_dst, cin, nt, accumulate, stats, out, stream, rows);
  }
  if (rows.dev) {
    gpn::set_error("gpn_spconv_fwd: a layer whose row count is a device counter must fit the masked-tile / direct kernels (K = %d, %d -> %d channels)", K, cin, cout);
    return GPN_ERR_ARG;
  }
  if (stats.slab || stats.twin.slab || stats.ep.mean) {
    gpn::set_error("gpn_spconv_fwd: this shape runs on a kernel without a BatchNorm epilogue (see spconv_fwd_accumulates_stats / spconv_fwd_applies_affine)");
    return GPN_ERR_ARG;
  }
  if (stats.twin.in) {  // the lock-step kernel takes one problem per launch
    const gpn::ConvTwin tw = stats.twin;
    int rc2 = gpn::spconv_fwd_into(in, packed_w, nbr, nbr_p, perm, K, n_dst, cin, cout, out, accumulate, gpn::ConvStats(), ws, ws_bytes, stream);
    if (rc2) return rc2;
    return gpn::spconv_fwd_into(tw.in, tw.packed, nbr, nbr_p, perm, K, n_dst, cin, cout, tw.out, accumulate, gpn::ConvStats(), ws, ws_bytes, stream);
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
    gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout, gpn::prof_shape_tag(K, n_dst, cin, cout, stats.twin.in != nullptr));
    switch (p.ntw) {
      case 1: rc = dispatch_cw<1>(p, in, packed_w, nbr, K, n_dst, cin, nt, p.splits > 1 ? 0 : accumulate, target, stream); break;
      case 2: rc = dispatch_cw<2>(p, in, packed_w, nbr, K, n_dst, cin, nt, p.splits > 1 ? 0 : accumulate, target, stream); break;
      case 3: rc = dispatch_cw<3>(p, in, packed_w, nbr, K, n_dst, cin, nt, p.splits > 1 ? 0 : accumulate, target, stream); break;
      default: rc = dispatch_cw<4>(p, in, packed_w, nbr, K, n_dst, cin, nt, p.splits > 1 ? 0 : accumulate, target, stream); break;
    }
    if (rc == GPN_OK && p.splits > 1) {
      const int64_t elems4 = n_dst * cout / 4;
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)gpn::cdiv(elems4, 256)), dim3(256), 0, stream, target,
                         p.splits, elems4, accumulate, out);
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
