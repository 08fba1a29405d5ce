// spconv_wgrad.hip — a second form of the weight-gradient contraction of the sparse convolutions for gfx950 (round 4): no LDS
// staging.  MEASURED AND NOT THE DEFAULT: the LDS-staged kernel of spconv.hip runs unless GPN_WGRAD_ROWS / gpn_spconv_wgrad_rows()
// say otherwise (tools/wgrad_bench.py, profiles/r04_wgrad_bench.txt: faster per layer from 64 channels up - 26.7 against 35.6 us
// for 64 -> 64 at 6.9k rows, 43.9 against 63.8 for 128 -> 64 - slower below, 96 against 29 us at 16 channels; in the training
// step, on the weight-gradient stream beside the dgrad chain, 7.87 - 8.17 ms against 7.81 - 8.00 with it on the >= 64-channel
// layers and 8.7 - 9.2 ms with it everywhere, four interleaved rounds).  Both kernels read every pair's two rows from L2 / the
// Infinity Cache and that, not the staging, is what they wait for.  Kept as the tested alternative
// (tests/test_gpu_ops.py::test_weight_gradient_kernels_agree runs both on every channel pair).
//
//   partial[s][k][ci][co] = sum over the s-th slice of tap k's pair list of in[src][ci] * dout[dst][co]
// (network/backbone.py:19-36,74-90,149-152: the gradient of every spconv.SubMConv3d / SparseConv3d / SparseInverseConv3d weight)
//
// The contraction runs over PAIRS, so pairs must sit on the K index of v_mfma_f32_16x16x4_f32 - lane l = (i16 = l % 16,
// g = l / 16) supplies A[m = i16][k = g] and B[k = g][n = i16] - while a row of `in` / `dout` is contiguous along the channels,
// i.e. along m / n.  The LDS kernel gathered 64-byte pieces of rows [pair][channel] into LDS and read them back transposed
// (4-byte ds_reads): two barriers, 256 x 16 B of LDS writes and 16 ds_read_b32 per 16 MFMAs and wave; 0.08 - 0.18 of the MFMA
// peak (level 0 ... level 1), 0.09 - 0.12 at the deeper levels (profiles/r04_kernel_stats.csv).
//
// Here the rows go from L2 straight into MFMA operands.  Lane (i16, g) loads 16 bytes - channels 4 q .. 4 q + 3 - of the row
// of pair (4 rg + g) of the step, where (rg, q) = (i16 / W, i16 % W) and W = lanes per row (quads of 4 channels: W = 16 for a
// 64-channel block, 8 for 32, 4 for 16).  Register j of that load IS an A (or B) operand: A_j[m = i16][k = g] =
// in[src(rg, g)][4 q + j].  The 16 products a_ja x b_jb, ja, jb in 0..3, cover the block's channel pairs (4 qa + ja, 4 qb + jb);
// element D[m][n] of a product is a valid sum over the step's rows where m and n belong to the same row group (the diagonal
// W x W blocks of D), and a meaningless cross term elsewhere, which is never stored.  MFMA rows doing useful work:
// qa qb / (16 W) - all of them for 64 -> 64 channels, 1/2 for 32 -> 32, 1/4 for 16 -> 16 - against no LDS traffic, no barrier
// and 2 loads + 2 ds_bpermute (the pair indices, read 64 at a time, coalesced) per 16 MFMAs.  Operands are requested 3 steps
// ahead through a ring of 4 slots (the tap loop of spconv_tiles.hip has the why and the how).
// Channels beyond 64 take more blocks (grid.z); a workgroup = 4 waves, each contracting a quarter of the slice; fixed-order
// sums over row groups and waves through LDS, over the slices by wgrad_reduce_many (spconv.hip) => deterministic.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  static_for_impl(f, std::make_integer_sequence<int, N>());
}
constexpr int kBlock = 64;  // channels per block (one 16-byte quad per lane of a 16-lane row)
constexpr int kRing = 4;    // operand slots

template <int W>  // lanes per row: 4 (16-channel blocks), 8 (32), 16 (48 / 64)
__global__ __launch_bounds__(256) void spconv_wgrad_rows_kernel(const gpn::WgradSets sets, int64_t n_tiles, int cin, int cout, int S,
                                                                int blocks_a, int blocks_b,
                                                                const int64_t* __restrict__ n_dst_dev) {
  // (device-counted rows, gpn::DevRows: the offset table's leading dimension is that of the LIVE row count)
  if (n_dst_dev) n_tiles = (gpn::live_rows(n_dst_dev, n_tiles * GPN_TILE_ROWS) + GPN_TILE_ROWS - 1) / GPN_TILE_ROWS;
  constexpr int RG = 16 / W;        // row groups of a 16-lane group
  constexpr int RPS = 4 * RG;       // pairs per step
  constexpr int BP = kRing * RPS;   // pairs per batch = per iteration of the main loop (kRing steps): 16 (W = 16), 32, 64
  constexpr uint32_t kOob = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float red[kBlock * kBlock];
  __shared__ int32_t sidx[4][2][2][BP];  // per wave: two batches of pair indices [src / dst][pair]

  const int k = blockIdx.x, s = blockIdx.y;
  const int per_set = blocks_a * blocks_b;
  const int set = blockIdx.z / per_set, bz = blockIdx.z - set * per_set;
  const int ba = bz / blocks_b, bb = bz - ba * blocks_b;
  const float* __restrict__ in = sets.s[set].in;
  const float* __restrict__ dout = sets.s[set].dout;
  const int32_t* __restrict__ pair_src = sets.s[set].pair_src;
  const int32_t* __restrict__ pair_dst = sets.s[set].pair_dst;
  const int32_t* __restrict__ tile_off = sets.s[set].tile_off;
  float* __restrict__ partial = sets.s[set].partial;
  const int K = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int rg = i16 / W, q = i16 % W;
  const int ca0 = ba * kBlock, cb0 = bb * kBlock;
  const int qa = min(cin - ca0, kBlock) >> 2, qb = min(cout - cb0, kBlock) >> 2;  // valid quads of this block pair

  const int32_t l_begin = tile_off[(int64_t)k * (n_tiles + 1)];
  const int32_t l_end = tile_off[(int64_t)k * (n_tiles + 1) + n_tiles];
  const int32_t len = l_end - l_begin;
  int32_t chunk = (len + S - 1) / S;
  chunk = (chunk + 3) & ~3;
  const int32_t sa = l_begin + s * chunk;
  const int32_t sb = min(sa + chunk, l_end);
  // the wave's quarter of the slice (whole steps)
  int32_t quarter = (max(sb - sa, 0) + 3) >> 2;
  quarter = (quarter + RPS - 1) / RPS * RPS;
  const int32_t pa = __builtin_amdgcn_readfirstlane(min(sa + wave * quarter, sb));
  const int32_t pb = __builtin_amdgcn_readfirstlane(min(pa + quarter, sb));

  f32x4 acc[4][4];
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[ja][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (pa < pb) {  // uniform per wave
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dout), 0, 0x7fffffff, 0x00020000);
    const int n_iters = (pb - pa + BP - 1) / BP;
    const int32_t last = pb - 1;
    const uint32_t a_row = (uint32_t)cin * 4u, b_row = (uint32_t)cout * 4u;
    const uint32_t a_col = q < qa ? (uint32_t)(ca0 + 4 * q) * 4u : kOob, b_col = q < qb ? (uint32_t)(cb0 + 4 * q) * 4u : kOob;
    const int my = rg * 4 + g;  // the lane's pair inside a step

    // pair indices: batch i = pairs pa + i BP ... of the wave, read coalesced by the first BP lanes (clamped: duplicates past
    // the wave's last pair, never multiplied) one iteration before they are stored to the wave's LDS slab and two before the
    // steps read them back (16 lanes per address: broadcast)
    const int il = lane < BP ? lane : BP - 1;
    auto load_idx = [&](int i, int32_t& is, int32_t& id) {
      const int32_t p = min(pa + i * BP + il, last);
      is = pair_src[p], id = pair_dst[p];
    };
    auto store_idx = [&](int i, int32_t is, int32_t id) {
      if (lane < BP) sidx[wave][i & 1][0][lane] = is, sidx[wave][i & 1][1][lane] = id;
    };
    f32x4 ra[kRing], rb[kRing];
    // operands of step u of iteration i (u >= kRing: step u - kRing of iteration i + 1)
    auto issue = [&](auto slot_tag, auto u_tag, int i) {
      constexpr int sl = decltype(slot_tag)::value, u = decltype(u_tag)::value;
      constexpr int st = u % kRing;
      const int bi = (i + u / kRing) & 1;
      const int32_t src = sidx[wave][bi][0][st * RPS + my], dst = sidx[wave][bi][1][st * RPS + my];
      const bool ok = pa + (i * kRing + u) * RPS + my < pb;
      const uint32_t ao = ok ? (uint32_t)src * a_row + a_col : kOob;  // (a_col = kOob keeps the sum out of range: rows * 4 C < 2^31)
      const uint32_t bo = ok ? (uint32_t)dst * b_row + b_col : kOob;
      ra[sl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)ao, 0, 0));
      rb[sl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)bo, 0, 0));
    };
    auto consume = [&](auto slot_tag) {
      constexpr int sl = decltype(slot_tag)::value;
#pragma unroll
      for (int ja = 0; ja < 4; ++ja)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
          acc[ja][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[sl][ja], rb[sl][jb], acc[ja][jb], 0, 0, 0);
    };
    int32_t is, id;
    load_idx(0, is, id);
    store_idx(0, is, id);
    load_idx(1, is, id);
    static_for<kRing - 1>([&](auto u) { issue(u, u, 0); });
    __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < n_iters; ++i) {  // (the last iteration may run up to kRing - 1 steps of zeros)
      store_idx(i + 1, is, id);
      load_idx(i + 2, is, id);
      static_for<kRing>([&](auto t) {
        constexpr int st = decltype(t)::value;
        issue(std::integral_constant<int, (st + kRing - 1) % kRing>(), std::integral_constant<int, st + kRing - 1>(), i);
        __builtin_amdgcn_sched_barrier(0);
        consume(t);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }

  // ---- D[m = 4 g + r][n = i16] of product (ja, jb): valid where m and n are lanes of the same row group; fixed-order sums over
  // the row groups and the waves in LDS (red[ci][co] of this block pair), then the block of partial[s][k] -----------------------
  const int na = qa * 4, nb = qb * 4;
  const int rg_n = i16 / W, qb_n = i16 % W;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;
        const int rg_m = m / W, qa_m = m % W;
        for (int pass = 0; pass < RG; ++pass) {  // (one LDS instruction sequence per row group: in order within the wave)
          if (rg_m == pass && rg_n == pass && qa_m < qa && qb_n < qb) {
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
              for (int jb = 0; jb < 4; ++jb) {
                float* e = red + (4 * qa_m + ja) * kBlock + 4 * qb_n + jb;
                if (w == 0 && pass == 0) *e = acc[ja][jb][r];
                else *e += acc[ja][jb][r];
              }
          }
        }
      }
    }
    __syncthreads();
  }
  float* pbase = partial + ((int64_t)s * K + k) * (int64_t)cin * cout;
  for (int e = tid; e < na * nb; e += 256) {
    const int ci = e / nb, co = e - ci * nb;
    pbase[(int64_t)(ca0 + ci) * cout + cb0 + co] = red[ci * kBlock + co];
  }
}

// GPN_WGRAD_ROWS / gpn_spconv_wgrad_rows(): 0 = never (the LDS-staged kernel of spconv.hip; default), 1 = layers with a side of
// >= 64 channels (where it is the faster one in isolation), 2 = every shape (tests, tools)
std::atomic<int> g_rows_mode{[] {
  const char* e = getenv("GPN_WGRAD_ROWS");
  return e ? atoi(e) : 0;
}()};

}  // namespace

namespace gpn {

bool wgrad_rows_supported(int64_t n_rows_bound, int cin, int cout) {
  // Both kernels read every pair's two rows from L2 and are bound by that (~4.5 TB/s of 64 - 256-byte pieces at the 80k-row
  // level); this one wins where the LDS kernel's staging overhead shows - blocks of 64 channels, few long slices - and loses
  // where 3/4 of its MFMA rows are cross terms and a wave has 3 steps to amortise its 64-register epilogue (16-channel
  // layers: 96 us against 29).  tools/wgrad_bench.py, profiles/r04_wgrad_bench.txt.
  const int mode = g_rows_mode.load(std::memory_order_relaxed);
  if (mode == 0 || (mode == 1 && std::max(cin, cout) < 64)) return false;
  // 32-bit byte offsets into the operands (source rows: at most 8 x the destination rows, for a stride-2 conv)
  return cin % 16 == 0 && cout % 16 == 0 && n_rows_bound * 8 * (int64_t)std::max(cin, cout) * 4 < ((int64_t)1 << 31);
}

int wgrad_rows_contract(const WgradSets& sets, int K, int64_t n_dst, int cin, int cout, int S, hipStream_t stream,
                        const int64_t* n_dst_dev) {
  const int64_t n_tiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const int blocks_a = (int)gpn::cdiv(cin, kBlock), blocks_b = (int)gpn::cdiv(cout, kBlock);
  const int widest = std::max(std::min(cin, kBlock), std::min(cout, kBlock)) / 4;  // quads of the wider operand block
  const dim3 grid(K, S, sets.n * blocks_a * blocks_b);
  if (widest <= 4)
    hipLaunchKernelGGL((spconv_wgrad_rows_kernel<4>), grid, dim3(256), 0, stream, sets, n_tiles, cin, cout, S, blocks_a, blocks_b, n_dst_dev);
  else if (widest <= 8)
    hipLaunchKernelGGL((spconv_wgrad_rows_kernel<8>), grid, dim3(256), 0, stream, sets, n_tiles, cin, cout, S, blocks_a, blocks_b, n_dst_dev);
  else
    hipLaunchKernelGGL((spconv_wgrad_rows_kernel<16>), grid, dim3(256), 0, stream, sets, n_tiles, cin, cout, S, blocks_a, blocks_b, n_dst_dev);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

}  // namespace gpn

// which weight-gradient contraction runs: 2 = rows straight into MFMA operands (this file) for every shape, 1 = for layers with
// a side of >= 64 channels, 0 = the LDS-staged kernel of spconv.hip everywhere (default); mode < 0 only queries.  Returns the previous value.  (env GPN_WGRAD_ROWS; tests and tools compare the two.)
extern "C" int gpn_spconv_wgrad_rows(int mode) {
  return mode < 0 ? g_rows_mode.load(std::memory_order_relaxed) : g_rows_mode.exchange(mode, std::memory_order_relaxed);
}
