// spconv.hip — kernel C: sparse convolution forward / dgrad / wgrad for gfx950.
// Replaces the gather-GEMM-scatter conv inside spconv (network/backbone.py:19-36,74-90,149-152).
//
// This file: weight packing, the weight-gradient contraction (+ its slice reductions) and the row gather / ordered scatter.
// The forward / dgrad kernels live in spconv_tiles.hip (masked-tile kernel, layers of >= 4096 row tiles) and spconv_fwd.hip
// (direct kernel, its tap-split form, the lock-step kernel for the tiniest layers).
//
// Design (MI355X / CDNA4, wave64):
//  * packed weights: [K][Cin/16][Cout/16][64 lanes][4] = the MFMA B fragment of one 16x16 tile per 1 KiB, with transpose /
//    tap-reverse flags for dgrad (gpn_spconv_pack_weights; the executor packs a whole network per pass, net.hip).
//  * wgrad contracts over pair lists: A = in[src]^T, B = dout[dst], 4 pairs per v_mfma_f32_16x16x4_f32 k-step, grid
//    (tap, pair slice, Cin group); a workgroup stages 64 pairs at a time through LDS (16-byte gathers, conflict-free pitch)
//    while the next tile's gathers and the one after's indices are in flight; fixed-order reductions over the 4 waves (LDS)
//    and over the slices (a second launch: 16 lanes per element for many slices, a thread per element - coalesced - for
//    <= 32) => deterministic.  The kernel is bound by gather bandwidth out of L2 (each row is re-read once per pair).
//    One launch contracts up to kWgradSets layers of one shape (gpn::wgrad_contract) and one launch sums the slices of up
//    to kWgradReduceJobs layers (gpn::wgrad_reduce_many): the executor (net.hip) defers and batches, gpn_spconv_wgrad is
//    the one-layer case of both.
//  * gather_rows / scatter_rows_csr: features[pc_voxel_id] and its transpose as an ordered CSR sum (deterministic).
#include <cstdlib>

#include "gpn_common.h"
#include "spconv_pack.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// weight packing:  packed[k][cb][nt][lane][s] = Wop[k][16cb + 4(lane>>4) + s][16nt + (lane&15)]
__global__ void pack_weights_kernel(const float* __restrict__ W, int K, int cin_w, int cout_w, int flags,
                                    float* __restrict__ packed) {
  const int64_t total = (int64_t)K * cin_w * cout_w;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  packed[t] = gpn::packed_weight_element(W, K, cin_w, cout_w, flags, t);
}

// ------------------------------------------------------------------------------------------------
// wgrad: partial[s][k][ci][co] = sum over the s-th slice of pair list k of in[src][ci] * dout[dst][co]
// grid = (K, S, CIG); 256 threads = 4 waves; fixed-order LDS reduction over the waves, fixed-order reduce over slices.
// LDS-staged: feeding every MFMA with per-lane 4-byte gathers in the MFMA's own operand layout (lane (i,g) = channel
// i of pair g) costs ~7 issued instructions per pair and is bound by instruction issue, not by the matrix pipe
// (10-18 TFLOP/s measured).  Here a workgroup stages 64 pairs at a time: thread (row = t/4, q = t%4) gathers
// 16-byte pieces of in[src[row]] and dout[dst[row]] (whole 64-byte pieces per 4 threads), the rows go through registers
// into LDS tiles ([pair][channel], pitch chosen so that the strided fragment reads below are bank-conflict free), and
// each wave contracts 16 of the 64 pairs: per k-step (4 pairs) CT + NT ds_read_b32 feed CT x NT MFMAs.  The gathers of
// tile i+1 and the pair indices of tile i+2 are in flight while tile i is contracted.
// Fixed assignment of pairs to waves and fixed-order reductions: deterministic.
template <int W>
struct LdsPitch {  // floats per LDS row for W payload floats: pitch % 64 in {16, 48} => rows g, g+1, g+2, g+3 hit distinct banks
  static constexpr int value = (W % 64 == 16 || W % 64 == 48) ? W : ((W + 16) % 64 == 16 || (W + 16) % 64 == 48) ? W + 16 : W + 32;
};

// (Round 6 measured TWO tiles of rows in flight per workgroup - a second register set, indices through a ring of four LDS slots,
// bit-equal results: slower, 21.9 -> 23.3 us per 16 -> 16 layer at 136k rows, 35.8 -> 38.5 at 32 -> 32, 7.65 - 7.67 -> 7.68 - 7.70 ms
// per step in three interleaved pairs, profiles/r06_findings.md: the extra 4 (CT + NT) registers cost more residency than the
// longer prefetch gains.)
template <int CT, int NT>
__global__ __launch_bounds__(256) void spconv_wgrad_lds_kernel(
    const gpn::WgradSets sets, int64_t n_tiles, int cin, int K, int S, int cigs, int n_z, int dealt, const int64_t* __restrict__ n_dst_dev) {
  // (device-counted rows, gpn::DevRows: the offset table's leading dimension is that of the LIVE row count)
  if (n_dst_dev) n_tiles = (gpn::live_rows(n_dst_dev, n_tiles * GPN_TILE_ROWS) + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS;
  constexpr int T = 64;  // pairs per tile
  constexpr int COUT = NT * 16;
  constexpr int PA = LdsPitch<CT * 16>::value, PB = LdsPitch<COUT>::value;
  constexpr int TILE_FLOATS = T * (PA + PB);
  constexpr int RED_FLOATS = CT * NT * 256;
  __shared__ __attribute__((aligned(16))) float smem[TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS];
  __shared__ int32_t sidx[2][2][T];  // [tile parity][src / dst][pair]
  float* sA = smem;
  float* sB = smem + T * PA;

  // Work item (tap k, pair slice s, z = (layer of this launch, Cin group)).  A slice of a tap's pair list (ordered by
  // destination row) covers about the same stretch of rows for every tap, so all taps and layers of a slice go to ONE XCD
  // (workgroups are dealt round-robin to the 8 XCDs: XCD x takes the slices [x S/8, (x+1) S/8), k fastest): its rows are
  // fetched into one L2 once and re-read there by the other taps.  With (k, s, z) = blockIdx every XCD streamed the whole
  // level - 20 MB at the 80k-row level against 4 MB of L2 - once per tap, i.e. every gathered row came from the Infinity Cache.
  int k, s, z;
  if (dealt) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, s8 = S >> 3;  // (S is a multiple of 8 here)
    const int kz = K * n_z;
    s = x * s8 + j / kz;
    const int rem = j - (j / kz) * kz;
    z = rem / K, k = rem - z * K;
  } else {
    k = blockIdx.x, s = blockIdx.y, z = blockIdx.z;
  }
  // z = (layers in this launch) x (Cin groups): layers of one shape - the same layer of the two networks of a paired
  // pass, consecutive layers of a level - are contracted by one launch (uniform per workgroup: scalar loads of the set)
  const int set = z / cigs, cig = z - set * cigs;
  const float* __restrict__ in = sets.s[set].in;
  const float* __restrict__ dout = sets.s[set].dout;
  const int32_t* __restrict__ pair_src = sets.s[set].pair_src;
  const int32_t* __restrict__ pair_dst = sets.s[set].pair_dst;
  const int32_t* __restrict__ tile_off = sets.s[set].tile_off;
  float* __restrict__ partial = sets.s[set].partial;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int row = tid >> 2, q = tid & 3;  // gather role: pair `row` of the tile, 16-byte piece q of each 64-byte block
  const int ct_tiles = cin >> 4;
  const int ct0 = cig * CT;

  const int32_t l_begin = tile_off[(int64_t)k * (n_tiles + 1)];
  const int32_t l_end = tile_off[(int64_t)k * (n_tiles + 1) + n_tiles];
  const int32_t len = l_end - l_begin;
  int32_t chunk = (len + S - 1) / S;
  chunk = (chunk + 3) & ~3;
  const int32_t a = l_begin + s * chunk;
  int32_t b = a + chunk;
  if (b > l_end) b = l_end;

  f32x4 acc[CT][NT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (a < b) {  // uniform per workgroup
    const int32_t last = b - 1;
    const int n_pair_tiles = (b - a + T - 1) / T;
    auto load_idx = [&](int tile, int32_t& s_, int32_t& d_) {  // threads 0..63: pair `tid` of the tile (clamped)
      int32_t p = a + tile * T + (tid & (T - 1));
      p = p < last ? p : last;
      s_ = pair_src[p];
      d_ = pair_dst[p];
    };
    auto gather = [&](int parity, f32x4 (&ra)[CT], f32x4 (&rb)[NT]) {
      const int32_t src = sidx[parity][0][row], dst = sidx[parity][1][row];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int t = (ct0 + ct < ct_tiles) ? (ct0 + ct) : (ct_tiles - 1);
        ra[ct] = *reinterpret_cast<const f32x4*>(in + (int64_t)src * cin + t * 16 + 4 * q);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rb[nt] = *reinterpret_cast<const f32x4*>(dout + (int64_t)dst * COUT + nt * 16 + 4 * q);
    };

    // prologue: indices of tiles 0 and 1, rows of tile 0
    int32_t is_, id_;
    load_idx(0, is_, id_);
    if (tid < T) { sidx[0][0][tid] = is_; sidx[0][1][tid] = id_; }
    load_idx(1, is_, id_);
    __syncthreads();
    f32x4 ra[CT], rb[NT];
    gather(0, ra, rb);
    if (tid < T) { sidx[1][0][tid] = is_; sidx[1][1][tid] = id_; }

    for (int i = 0; i < n_pair_tiles; ++i) {
      __syncthreads();  // the previous tile's fragment reads are done; sidx[(i+1)&1] is visible
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<f32x4*>(sA + row * PA + ct * 16 + 4 * q) = ra[ct];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(sB + row * PB + nt * 16 + 4 * q) = rb[nt];
      gather((i + 1) & 1, ra, rb);  // rows of tile i+1 (clamped duplicates past the end): in flight during the contraction
      load_idx(i + 2, is_, id_);
      __syncthreads();  // tile i is in LDS
      const int32_t p_base = a + i * T + wave * 16;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int pl = wave * 16 + st * 4 + g;  // pair of this lane in this k-step
        const bool valid = p_base + st * 4 + g < b;
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = sB[pl * PB + nt * 16 + i16];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          float av = sA[pl * PA + ct * 16 + i16];
          av = (valid && ct0 + ct < ct_tiles) ? av : 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[ct][nt], 0, 0, 0);
        }
      }
      if (tid < T) { sidx[i & 1][0][tid] = is_; sidx[i & 1][1][tid] = id_; }  // indices of tile i+2
    }
    __syncthreads();  // the tiles alias the reduction buffer below
  }

  // fixed-order reduction over the 4 waves through LDS (element (ct,nt,r,lane) -> red[((ct*NT+nt)*4+r)*64+lane])
  float* red = smem;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* qd = red + ((ct * NT + nt) * 4 + r) * 64 + lane;
            if (w == 0) *qd = acc[ct][nt][r];
            else *qd += acc[ct][nt][r];
          }
    }
    __syncthreads();
  }
  float* pbase = partial + ((int64_t)s * K + k) * (int64_t)cin * COUT;
  for (int e = tid; e < CT * NT * 256; e += 256) {
    const int l = e & 63, r = (e >> 6) & 3, tn = e >> 8;
    const int nt = tn % NT, ct = tn / NT;
    if (ct0 + ct >= ct_tiles) continue;
    const int ci = (ct0 + ct) * 16 + 4 * (l >> 4) + r;
    const int co = nt * 16 + (l & 15);
    pbase[(int64_t)ci * COUT + co] = red[e];
  }
}

// dW[e] = sum_s partial[s][e]: 16 lanes per element stride over the slices, then a fixed-order shuffle tree
// (block `blk` of cdiv(elems, 16) blocks of 256 threads)
__device__ __forceinline__ void wgrad_reduce_tree(const float* __restrict__ partial, int S, int64_t elems, int K, int cin,
                                                  int cout, int oki, float* __restrict__ dW, uint32_t blk) {
  const int part = threadIdx.x & 15;
  const int64_t e = (int64_t)blk * 16 + (threadIdx.x >> 4);
  float acc = 0.f;
  if (e < elems) {
    // four slices in flight per lane (same order of additions): with one load per iteration the loop was a chain of
    // S / 16 memory round trips
    const float* __restrict__ p = partial + e;
    int s = part;
    for (; s + 48 < S; s += 64) {
      const float v0 = p[(int64_t)s * elems], v1 = p[(int64_t)(s + 16) * elems];
      const float v2 = p[(int64_t)(s + 32) * elems], v3 = p[(int64_t)(s + 48) * elems];
      acc += v0;
      acc += v1;
      acc += v2;
      acc += v3;
    }
    for (; s < S; s += 16) acc += p[(int64_t)s * elems];
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 16);
  if (part == 0 && e < elems) {
    int64_t o = e;  // e = (k * cin + ci) * cout + co
    if (oki) {
      const int co = (int)(e % cout);
      const int64_t r = e / cout;
      const int ci = (int)(r % cin), k = (int)(r / cin);
      o = ((int64_t)co * K + k) * cin + ci;  // gradient in the parameter's own [Cout][K][Cin] layout
    }
    dW[o] = acc;
  }
}

// the same sum for FEW slices (the deep levels: many weight elements, S <= 32): a thread per element, the slices read in
// slice order with every load in flight together - coalesced across the threads of a wave.  The 16-lane form above reads
// 16 slices per element at once, i.e. sixteen 16-byte pieces per wave load: 22 us for the 8 MB of a 64 -> 64 layer's
// partials (0.36 TB/s).  Sum order: slice 0, 1, 2, ... (deterministic; differs from the tree above in the last bits).
// (block `blk` of cdiv(elems, 256) blocks of 256 threads)
__device__ __forceinline__ void wgrad_reduce_few(const float* __restrict__ partial, int S, int64_t elems, int K, int cin,
                                                 int cout, int oki, float* __restrict__ dW, uint32_t blk) {
  const int64_t e = (int64_t)blk * 256 + threadIdx.x;
  if (e >= elems) return;
  const float* __restrict__ p = partial + e;
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= S; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(s + u) * elems];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; s < S; ++s) acc += p[(int64_t)s * elems];
  int64_t o = e;  // e = (k * cin + ci) * cout + co
  if (oki) {
    const int co = (int)(e % cout);
    const int64_t r = e / cout;
    const int ci = (int)(r % cin), k = (int)(r / cin);
    o = ((int64_t)co * K + k) * cin + ci;  // gradient in the parameter's own [Cout][K][Cin] layout
  }
  dW[o] = acc;
}

// The slice sums of up to kWgradReduceJobs layers in ONE launch (the executor defers them: a contraction writes its partials
// to its own piece of the workspace and the sums of a batch of layers run together - ~100 reduce launches of 3-8 us per
// training step, each a dependent launch on the weight-gradient stream, become ~15).  The job table travels as a kernel
// argument; a workgroup finds its job by a scan of the (uniform) block offsets.
struct ReduceBatch {
  int n;
  uint32_t block_end[gpn::kWgradReduceJobs];  // exclusive prefix of the jobs' block counts
  gpn::WgradReduceJob job[gpn::kWgradReduceJobs];
};
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(const ReduceBatch b) {
  int j = 0;
  while (j + 1 < b.n && blockIdx.x >= b.block_end[j]) ++j;
  const uint32_t blk = blockIdx.x - (j ? b.block_end[j - 1] : 0u);
  const gpn::WgradReduceJob& q = b.job[j];
  if (q.few) wgrad_reduce_few(q.partial, q.S, q.elems, q.K, q.cin, q.cout, q.oki, q.dW, blk);
  else wgrad_reduce_tree(q.partial, q.S, q.elems, q.K, q.cin, q.cout, q.oki, q.dW, blk);
}

int wgrad_splits(int K, int cin, int cout, int64_t n_dst) {
  const int ct_tiles = cin / 16;
  const int CT = ct_tiles < 4 ? ct_tiles : 4;
  const int cig = (ct_tiles + CT - 1) / CT;
  // enough pair slices for ~4096 workgroups (each slice is a latency-bound gather loop), at least ~128 dst rows per slice,
  // and at most 4 MB of partials (they are written and re-read by the slice sums, through the L2s the dgrad chain's gathers
  // live in: with the sums deferred and batched, interleaved runs on one box gave 9.42 / 9.36 / 9.08 ms per step at 8 / 6 / 4 MB,
  // 8.63 / 9.10 / 9.70 at 4 / 3 / 2 on another, 8.90 / 9.29 / 9.36 at 8 / 16 / 32 on a third).  Earlier sweeps, round 3
  // (profiles/r03_findings.md): rows per slice 32 / 64 / 128 / 384 / 768 / 1536 -> 9.43 / 9.5 / 8.9 / 9.0 / 9.4 / 9.95 ms per
  // step; partial cap 2 / 3 / 4 / 6 / 8 / 16 MB -> 10.3 / 9.8 / 9.1 / 9.7 / 9.3 / 9.5 ms in single runs (box noise +-0.3), and
  // 4 MB (with the thread-per-element reduce up to 64 slices) 9.32 vs 9.16 ms for 8 MB in an interleaved same-box A/B.
  // (the three constants were environment switches until round 6; the sweeps above are their record)
  constexpr int64_t target_wgs = 4096, partial_mb = 4, rows_per_slice = 128;
  int64_t S = target_wgs / ((int64_t)K * cig);
  const int64_t cap = n_dst / rows_per_slice;
  if (S > cap) S = cap;
  const int64_t mem_cap = (partial_mb << 20) / ((int64_t)K * cin * cout * 4);
  if (S > mem_cap) S = mem_cap;
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  if (S >= 8) S &= ~(int64_t)7;  // whole eighths: the slices of a launch are dealt to the 8 XCDs (spconv_wgrad_lds_kernel)
  return (int)S;
}

template <int CT, int NT>
int launch_wgrad(const gpn::WgradSets& sets, int K, int64_t n_dst, int cin, int S, hipStream_t stream, const int64_t* n_dev) {
  const int64_t n_tiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const int ct_tiles = cin / 16;
  const int cig = (ct_tiles + CT - 1) / CT;
  const int n_z = sets.n * cig;
  const int dealt = (S >= 8 && S % 8 == 0) ? 1 : 0;  // slices dealt to the XCDs (round 4: -10 % per layer at the two large levels)
  const dim3 grid = dealt ? dim3((unsigned)(K * S * n_z)) : dim3(K, S, n_z);
  hipLaunchKernelGGL((spconv_wgrad_lds_kernel<CT, NT>), grid, dim3(256), 0, stream, sets, n_tiles, cin, K, S, cig, n_z, dealt, n_dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int CT>
int dispatch_wgrad_nt(int nt, const gpn::WgradSets& sets, int K, int64_t n_dst, int cin, int S, hipStream_t stream,
                      const int64_t* n_dev) {
  switch (nt) {
#define GPN_CASE(N) \
  case N: return launch_wgrad<CT, N>(sets, K, n_dst, cin, S, stream, n_dev);
    GPN_CASE(1) GPN_CASE(2) GPN_CASE(3) GPN_CASE(4) GPN_CASE(5) GPN_CASE(6) GPN_CASE(7) GPN_CASE(8)
#undef GPN_CASE
    default:
      gpn::set_error("gpn_spconv_wgrad: cout=%d not supported (multiple of 16, <= 128)", nt * 16);
      return GPN_ERR_ARG;
  }
}

// ------------------------------------------------------------------------------------------------ G
// U float4 elements per thread, the U index loads issued together and then the U row loads together: two memory round
// trips per U elements (one element per thread ran at half the bandwidth of a copy of the same size - every wave was a
// serial idx -> row chain and the launch needed two rounds of waves); 32-bit element arithmetic
// (n_dev != nullptr, gpn::DevRows: the row count is a device counter; the launch then walks its elements in rounds of
// U x (threads of the grid) - one round when the grid was sized from the count)
template <int U>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx,
                                                          uint32_t total4, uint32_t C4, float* __restrict__ out,
                                                          const int64_t* __restrict__ n_dev) {
  if (n_dev) total4 = (uint32_t)gpn::live_rows(n_dev, total4 / C4) * C4;
  const uint32_t T = gridDim.x * blockDim.x;
  for (uint32_t base = 0; base < total4; base += (uint32_t)U * T) {
  const uint32_t t0 = base + blockIdx.x * blockDim.x + threadIdx.x;
  int32_t r[U];
  uint32_t c[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const uint32_t t = t0 + (uint32_t)k * T;
    const uint32_t row = t < total4 ? t / C4 : 0u;
    c[k] = t - row * C4;
    r[k] = t < total4 ? idx[row] : -1;
  }
  f32x4 v[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (r[k] >= 0) v[k] = reinterpret_cast<const f32x4*>(table)[(uint64_t)(uint32_t)r[k] * C4 + c[k]];
  }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const uint32_t t = t0 + (uint32_t)k * T;
    if (t < total4) reinterpret_cast<f32x4*>(out)[t] = v[k];
  }
  }
}
__global__ void gather_rows_scalar_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx,
                                          int64_t n, int C, float* __restrict__ out, const int64_t* __restrict__ n_dev) {
  n = gpn::live_rows(n_dev, n);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    const int32_t r = idx[i];
    out[t] = r >= 0 ? table[(int64_t)r * C + c] : 0.f;
  }
}
// dtable[r][c] = ordered sum over the points of row r
__global__ void scatter_rows_csr_kernel(const float* __restrict__ dout, const int32_t* __restrict__ order,
                                        const int32_t* __restrict__ starts, int64_t n_rows, int C,
                                        float* __restrict__ dtable, const int64_t* __restrict__ n_dev) {
  n_rows = gpn::live_rows(n_dev, n_rows);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_rows * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    float acc = 0.f;
    for (int32_t j = starts[r]; j < starts[r + 1]; ++j) acc += dout[(int64_t)order[j] * C + c];
    dtable[t] = acc;
  }
}
// float4 form (C % 4 == 0): same order of additions per channel; the first two points of a row are fetched together
// (rows hold 1.1 points on average: the common case is one order -> row chain, not a loop)
__global__ __launch_bounds__(256) void scatter_rows_csr_v4_kernel(const float* __restrict__ dout,
                                                                   const int32_t* __restrict__ order,
                                                                   const int32_t* __restrict__ starts, uint32_t total4,
                                                                   uint32_t C4, float* __restrict__ dtable,
                                                                   const int64_t* __restrict__ n_dev) {
  if (n_dev) total4 = (uint32_t)gpn::live_rows(n_dev, total4 / C4) * C4;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += gridDim.x * blockDim.x) {
  const uint32_t r = t / C4, c = t - r * C4;
  const int32_t b = starts[r], e = starts[r + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (b < e) {
    const int32_t p0 = order[b];
    const int32_t p1 = b + 1 < e ? order[b + 1] : -1;
    const f32x4 v0 = reinterpret_cast<const f32x4*>(dout)[(uint64_t)(uint32_t)p0 * C4 + c];
    f32x4 v1 = {0.f, 0.f, 0.f, 0.f};
    if (p1 >= 0) v1 = reinterpret_cast<const f32x4*>(dout)[(uint64_t)(uint32_t)p1 * C4 + c];
    acc = acc + v0;
    if (p1 >= 0) acc = acc + v1;
    for (int32_t j = b + 2; j < e; ++j) acc = acc + reinterpret_cast<const f32x4*>(dout)[(uint64_t)(uint32_t)order[j] * C4 + c];
  }
  reinterpret_cast<f32x4*>(dtable)[t] = acc;
  }
}

}  // namespace

// ================================================================================================
extern "C" int gpn_spconv_pack_weights(const float* W, int K, int cin_w, int cout_w, int flags, float* packed,
                                       gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(W && packed && K >= 1);
  GPN_CHECK_ARG(cin_w >= 16 && cout_w >= 16 && cin_w % 16 == 0 && cout_w % 16 == 0);
  const int64_t total = (int64_t)K * cin_w * cout_w;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((int)gpn::cdiv(total, 256)), dim3(256), 0, stream, W, K, cin_w,
                     cout_w, flags, packed);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" size_t gpn_spconv_wgrad_ws_bytes(int K, int cin, int cout, int64_t n_dst) {
  const int S = wgrad_splits(K, cin, cout, n_dst);
  return gpn::align_up((size_t)S * K * cin * cout * sizeof(float));
}

namespace gpn {

int wgrad_slices(int K, int cin, int cout, int64_t n_dst) { return wgrad_splits(K, cin, cout, n_dst); }

// the contraction of sets.n layers of ONE shape (K, n_dst, cin, cout) into their partial[S][K][cin][cout] buffers
int wgrad_contract(const WgradSets& sets, int K, int64_t n_dst, int cin, int cout, int S, hipStream_t stream,
                   const int64_t* n_dst_dev) {
  GPN_CHECK_ARG(sets.n >= 1 && sets.n <= kWgradSets);
  const int ct_tiles = cin / 16;
  const int CT = ct_tiles < 4 ? ct_tiles : 4;
  const int nt = cout / 16;
  gpn::ProfScope prof(GPN_K_SPCONV_WGRAD, stream, 0.0, 0.0);
  switch (CT) {
    case 1: return dispatch_wgrad_nt<1>(nt, sets, K, n_dst, cin, S, stream, n_dst_dev);
    case 2: return dispatch_wgrad_nt<2>(nt, sets, K, n_dst, cin, S, stream, n_dst_dev);
    case 3: return dispatch_wgrad_nt<3>(nt, sets, K, n_dst, cin, S, stream, n_dst_dev);
    default: return dispatch_wgrad_nt<4>(nt, sets, K, n_dst, cin, S, stream, n_dst_dev);
  }
}

WgradReduceJob wgrad_reduce_job(const float* partial, int S, int K, int cin, int cout, int flags, float* dW) {
  WgradReduceJob q;
  q.partial = partial, q.dW = dW;
  q.elems = (int64_t)K * cin * cout;
  q.S = S, q.K = K, q.cin = cin, q.cout = cout;
  q.oki = (flags & GPN_LAYOUT_OKI) ? 1 : 0;
  q.few = (S <= 32 && q.elems >= 16384) ? 1 : 0;
  return q;
}

int wgrad_reduce_many(const WgradReduceJob* jobs, int n, hipStream_t stream) {
  if (n <= 0) return GPN_OK;
  GPN_CHECK_ARG(n <= kWgradReduceJobs);
  ReduceBatch b;
  b.n = n;
  uint32_t blocks = 0;
  for (int j = 0; j < n; ++j) {
    b.job[j] = jobs[j];
    blocks += (uint32_t)gpn::cdiv(jobs[j].elems, jobs[j].few ? 256 : 16);
    b.block_end[j] = blocks;
  }
  hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(blocks), dim3(256), 0, stream, b);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

}  // namespace gpn

extern "C" int gpn_spconv_wgrad(const float* in, const float* dout, const int32_t* pair_src,
                                const int32_t* pair_dst, const int32_t* tile_off, int K, int64_t n_dst, int cin,
                                int cout, int flags, float* dW, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(K >= 1 && n_dst >= 0 && dW);
  GPN_CHECK_ARG(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0);
  const int64_t elems = (int64_t)K * cin * cout;
  if (n_dst == 0) {
    GPN_CHECK_HIP(hipMemsetAsync(dW, 0, sizeof(float) * elems, stream));
    return GPN_OK;
  }
  GPN_CHECK_ARG(in && dout && pair_src && pair_dst && tile_off);
  const int S = wgrad_splits(K, cin, cout, n_dst);
  if (ws_bytes < (size_t)S * elems * sizeof(float) || !ws) {
    gpn::set_error("gpn_spconv_wgrad: workspace too small");
    return GPN_ERR_WS;
  }
  float* partial = static_cast<float*>(ws);
  gpn::WgradSets sets;
  sets.n = 1;
  sets.s[0] = gpn::WgradSet{in, dout, pair_src, pair_dst, tile_off, partial};
  int rc = gpn::wgrad_contract(sets, K, n_dst, cin, cout, S, stream);
  if (rc != GPN_OK) return rc;
  const gpn::WgradReduceJob job = gpn::wgrad_reduce_job(partial, S, K, cin, cout, flags, dW);
  return gpn::wgrad_reduce_many(&job, 1, stream);
}

static int gather_rows_impl(const float* table, const int32_t* idx, int64_t n, const gpn::DevRows& rows, int C, float* out,
                            hipStream_t stream) {
  GPN_CHECK_ARG(n >= 0 && C >= 1);
  if (n == 0) return GPN_OK;
  GPN_CHECK_ARG(table && idx && out);
  const int64_t np = gpn::plan_rows(n, rows);
  const bool dev = rows.dev != nullptr;
  if (C % 4 == 0 && n * (C / 4) < (int64_t)0x7fffffff) {
    const int64_t total4 = n * (C / 4), plan4 = np * (C / 4);
    if (plan4 >= 4 * 256 * 256)  // enough elements to keep every CU busy with 4 per thread
      hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(gpn::dev_grid(gpn::cdiv(total4, 4 * 256), gpn::cdiv(plan4, 4 * 256), dev)), dim3(256), 0,
                         stream, table, idx, (uint32_t)total4, (uint32_t)(C / 4), out, rows.dev);
    else
      hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(gpn::dev_grid(gpn::cdiv(total4, 256), gpn::cdiv(plan4, 256), dev)), dim3(256), 0, stream,
                         table, idx, (uint32_t)total4, (uint32_t)(C / 4), out, rows.dev);
  } else {
    hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3(gpn::dev_grid(gpn::cdiv(n * C, 256), gpn::cdiv(np * C, 256), dev)), dim3(256), 0, stream,
                       table, idx, n, C, out, rows.dev);
  }
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_gather_rows(const float* table, const int32_t* idx, int64_t n, int C, float* out,
                               gpn_stream_t stream_) {
  return gather_rows_impl(table, idx, n, gpn::DevRows(), C, out, (hipStream_t)stream_);
}
// row count on the device: n = the bound of idx / out, *n_dev the live count, n_plan the host's estimate (grid size only)
extern "C" int gpn_gather_rows_dev(const float* table, const int32_t* idx, int64_t n, const int64_t* n_dev, int64_t n_plan, int C,
                                   float* out, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev != nullptr);
  return gather_rows_impl(table, idx, n, gpn::DevRows{n_dev, n_plan}, C, out, (hipStream_t)stream_);
}

static int scatter_rows_csr_impl(const float* dout, const int32_t* order, const int32_t* starts, int64_t n_rows,
                                 const gpn::DevRows& rows, int C, float* dtable, hipStream_t stream) {
  GPN_CHECK_ARG(n_rows >= 0 && C >= 1);
  if (n_rows == 0) return GPN_OK;
  GPN_CHECK_ARG(starts && dtable);  // dout / order may be NULL when no point exists (every starts[r] is then 0)
  const int64_t np = gpn::plan_rows(n_rows, rows);
  const bool dev = rows.dev != nullptr;
  if (C % 4 == 0 && n_rows * (C / 4) < (int64_t)0x7fffffff)
    hipLaunchKernelGGL(scatter_rows_csr_v4_kernel, dim3(gpn::dev_grid(gpn::cdiv(n_rows * (C / 4), 256), gpn::cdiv(np * (C / 4), 256), dev)),
                       dim3(256), 0, stream, dout, order, starts, (uint32_t)(n_rows * (C / 4)), (uint32_t)(C / 4), dtable, rows.dev);
  else
    hipLaunchKernelGGL(scatter_rows_csr_kernel, dim3(gpn::dev_grid(gpn::cdiv(n_rows * C, 256), gpn::cdiv(np * C, 256), dev)), dim3(256), 0,
                       stream, dout, order, starts, n_rows, C, dtable, rows.dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

extern "C" int gpn_scatter_rows_csr(const float* dout, const int32_t* order, const int32_t* starts,
                                    int64_t n_rows, int C, float* dtable, gpn_stream_t stream_) {
  return scatter_rows_csr_impl(dout, order, starts, n_rows, gpn::DevRows(), C, dtable, (hipStream_t)stream_);
}
extern "C" int gpn_scatter_rows_csr_dev(const float* dout, const int32_t* order, const int32_t* starts, int64_t n_rows,
                                        const int64_t* n_dev, int64_t n_plan, int C, float* dtable, gpn_stream_t stream_) {
  GPN_CHECK_ARG(n_dev != nullptr);
  return scatter_rows_csr_impl(dout, order, starts, n_rows, gpn::DevRows{n_dev, n_plan}, C, dtable, (hipStream_t)stream_);
}
