// spconv_fwd.hip — the fused gather-MFMA-scatter sparse convolution (forward and dgrad launches) for gfx950.
//
// Workgroup = 256 threads = 4 waves.  Wave w of workgroup (x, y, z) owns
//     32 destination rows (tile 4x + w)  x  NTW 16-wide output-column tiles (group y)  x  taps [z*TS, (z+1)*TS).
// Its fp32 accumulators live in a private LDS tile; the output is written once (or, when taps are split across
// workgroups for small layers, once per split into a partial buffer that a fixed-order reduction sums).
//
// Memory-latency plan (the layers of this network are small: what limits them is dependent-load latency, not flops):
//   * the wave's block entries (source row, local destination row, tap) for ALL its taps are copied into LDS with
//     coalesced loads at kernel start, so a gathered row's address never waits on a global load;
//   * gathered rows (MFMA A operands) are prefetched PD (block, 64-channel chunk) steps ahead into a register ring:
//     lane (i = l&15, g = l>>4) loads channels [16cb+4g, 16cb+4g+4) of pair i's row with one 16-byte load, so every
//     gathered row is read as whole contiguous 64-byte pieces;
//   * weights (MFMA B operands, pre-packed per (tap, channel block, column tile) as 1 KiB wave fragments) reach the
//     waves through an LDS slab shared by the workgroup that holds as many whole taps as fit in 20 KiB, so there are
//     only ceil(taps / taps_per_slab) barrier-separated stages and B reads are conflict-free ds_read_b128;
//   * v_mfma_f32_16x16x4_f32 does the per-rule dense contraction (exact fp32 == an fmaf chain); summation order is
//     fixed (tap-major), so results are deterministic.
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlabBytes = 20 * 1024;
constexpr int kMaxBlocks = 56;  // <= 2 blocks per (32-row tile, tap), 27 taps -> 54

template <int NTW, int PD>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ in, const float* __restrict__ packed, const int32_t* __restrict__ blk_src,
    const int32_t* __restrict__ blk_meta, const int32_t* __restrict__ blk_off, int K, int64_t n_dst, int64_t n_wtiles,
    int cin, int nt_total, int taps_per_split, int taps_per_slab, float* __restrict__ out) {
  constexpr int LDW = NTW * 16 + 16;  // +16 floats: rows an odd distance apart land on disjoint bank halves
  constexpr int ROWS = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* slab = reinterpret_cast<f32x4*>(smem);                                   // kSlabBytes
  float* acc_all = smem + kSlabBytes / 4;                                          // [4][ROWS][LDW]
  int32_t* ent_src_all = reinterpret_cast<int32_t*>(acc_all + 4 * ROWS * LDW);     // [4][kMaxBlocks*16]
  uint8_t* ent_dst_all = reinterpret_cast<uint8_t*>(ent_src_all + 4 * kMaxBlocks * 16);  // [4][kMaxBlocks*16]
  uint8_t* ent_tap_all = ent_dst_all + 4 * kMaxBlocks * 16;                        // [4][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  float* acc_lds = acc_all + (size_t)wave * ROWS * LDW;
  int32_t* ent_src = ent_src_all + wave * kMaxBlocks * 16;
  uint8_t* ent_dst = ent_dst_all + wave * kMaxBlocks * 16;
  uint8_t* ent_tap = ent_tap_all + wave * 64;

  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  const bool active = w < n_wtiles;
  const int nt0 = blockIdx.y * NTW;
  const int ntw = (nt_total - nt0 < NTW) ? (nt_total - nt0) : NTW;
  const int k_lo = blockIdx.z * taps_per_split;
  const int k_hi = (k_lo + taps_per_split < K) ? (k_lo + taps_per_split) : K;
  const int64_t row0 = w * ROWS;
  const int cout = nt_total * 16;
  const int CB = cin >> 4;
  const int NCH = (CB + 3) >> 2;
  const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(packed);

  for (int e = lane * 4; e < ROWS * LDW; e += 64 * 4) *reinterpret_cast<f32x4*>(acc_lds + e) = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- stage this wave's block entries in LDS (coalesced) -----------------------------------------------------
  int32_t b_lo = 0, nblk = 0;
  if (active) {
    b_lo = blk_off[w * K + k_lo];
    nblk = blk_off[w * K + k_hi] - b_lo;
  }
  for (int e = lane; e < nblk * 16; e += 64) {
    const int32_t s = blk_src[(int64_t)b_lo * 16 + e];
    const int32_t m = blk_meta[(int64_t)b_lo * 16 + e];
    ent_src[e] = s;
    ent_dst[e] = m < 0 ? (uint8_t)255 : (uint8_t)(m & 0xff);
    if ((e & 15) == 0) ent_tap[e >> 4] = (uint8_t)(m >> 8);  // entry 0 of a block is always valid
  }
  __syncthreads();

  const int nsteps = nblk * NCH;
  // gathered-row prefetch ring: slot u holds the A chunk of step t with t % PD == u
  f32x4 a[PD][4];
  auto issue_a = [&](int t, f32x4 (&dst)[4]) {
    int32_t src = -1;
    int ch = 0;
    if (t < nsteps) {
      const int blk = t / NCH;
      ch = t - blk * NCH;
      src = ent_src[blk * 16 + i16];
    }
    const float* arow = in + (int64_t)src * cin + ch * 64 + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dst[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (src >= 0 && ch * 4 + c < CB) dst[c] = *reinterpret_cast<const f32x4*>(arow + c * 16);
    }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) issue_a(u, a[u]);

  f32x4 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int t_begin = 0;
  for (int ks = k_lo; ks < k_hi; ks += taps_per_slab) {
    const int ke = (ks + taps_per_slab < k_hi) ? (ks + taps_per_slab) : k_hi;
    // ---- weights of taps [ks, ke) -> LDS slab ([tap][cb][nt][lane] float4) ----
    if (ks != k_lo) __syncthreads();
    const int pieces = (ke - ks) * CB * NTW;  // 1 KiB fragments
    for (int q = tid; q < pieces * 64; q += 256) {
      const int p = q >> 6;
      const int nt = p % NTW;
      const int rest = p / NTW;  // (tap - ks) * CB + cb
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (nt < ntw) v = pw[((int64_t)(ks * CB + rest) * nt_total + nt0 + nt) * 64 + (q & 63)];
      slab[q] = v;
    }
    __syncthreads();

    // ---- this wave's steps whose tap lies in [ks, ke) ----
    int t_end = 0;
    if (active) t_end = (blk_off[w * K + ke] - b_lo) * NCH;
    for (int base = (t_begin / PD) * PD; base < t_end; base += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int t = base + u;
        if (t >= t_begin && t < t_end) {
          const int blk = t / NCH;
          const int ch = t - blk * NCH;
          const int tap = ent_tap[blk];
          const f32x4* sb = slab + ((int64_t)((tap - ks) * CB + ch * 4) * NTW) * 64 + lane;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (ch * 4 + c < CB) {
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt) {
                if (nt < ntw) {
                  const f32x4 bf = sb[(c * NTW + nt) * 64];
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c].x, bf.x, acc[nt], 0, 0, 0);
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c].y, bf.y, acc[nt], 0, 0, 0);
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c].z, bf.z, acc[nt], 0, 0, 0);
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c].w, bf.w, acc[nt], 0, 0, 0);
                }
              }
            }
          }
          issue_a(t + PD, a[u]);
          if (ch == NCH - 1) {
            // D[row = 4g + r][col = i16] belongs to pair 4g + r of the block
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = ent_dst[blk * 16 + 4 * g + r];
              if (row != 255) {
                float* dstp = acc_lds + row * LDW + i16;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                  if (nt < ntw) dstp[nt * 16] += acc[nt][r];
              }
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
    t_begin = t_end;
  }

  // ---- write this wave's columns of its tile (split z writes partial z) ---------------------------------------
  if (!active) return;
  float* outz = out + (int64_t)blockIdx.z * n_dst * cout;
  const int64_t rows_here = (n_dst - row0 < ROWS) ? (n_dst - row0) : ROWS;
  const int v4 = ntw * 4;
  for (int e = lane; e < (int)rows_here * v4; e += 64) {
    const int r = e / v4, c4 = e - r * v4;
    *reinterpret_cast<f32x4*>(outz + (row0 + r) * cout + nt0 * 16 + c4 * 4) =
        *reinterpret_cast<const f32x4*>(acc_lds + r * LDW + c4 * 4);
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, int splits, int64_t elems4,
                                       float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= elems4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(partial);
  f32x4 acc = p[t];
  for (int s = 1; s < splits; ++s) acc += p[(int64_t)s * elems4 + t];
  reinterpret_cast<f32x4*>(out)[t] = acc;
}

struct FwdPlan {
  int ntw, splits, taps_per_split, taps_per_slab;
};

FwdPlan plan_fwd(int K, int64_t n_dst, int cin, int cout) {
  const int nt = cout / 16, CB = cin / 16;
  const int64_t row_wgs = gpn::cdiv(gpn::cdiv(n_dst, GPN_TILE_ROWS), 4);
  FwdPlan p;
  // column tiles per workgroup: as many as possible (rows are re-gathered once per column group) with >= 512
  // workgroups in flight, and one tap's weights (CB * ntw KiB) must fit the slab
  p.ntw = 1;
  for (int ntw = 4; ntw > 1; --ntw) {
    if (ntw > nt || CB * ntw * 1024 > kSlabBytes) continue;
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
  int tps = kSlabBytes / (CB * p.ntw * 1024);
  if (tps < 1) tps = 1;
  p.taps_per_slab = tps;
  return p;
}

template <int NTW, int PD>
int launch_fwd(const FwdPlan& p, const float* in, const float* packed, const int32_t* blk_src, const int32_t* blk_meta,
               const int32_t* blk_off, int K, int64_t n_dst, int cin, int nt_total, float* out, hipStream_t stream) {
  const int64_t n_wtiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const dim3 grid((unsigned)gpn::cdiv(n_wtiles, 4), (unsigned)gpn::cdiv(nt_total, NTW), (unsigned)p.splits);
  const size_t lds = (size_t)kSlabBytes + (size_t)4 * 32 * (NTW * 16 + 16) * sizeof(float) +
                     (size_t)4 * kMaxBlocks * 16 * (sizeof(int32_t) + 1) + 4 * 64;
  static bool attr_set = false;
  if (!attr_set) {
    GPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_fwd_kernel<NTW, PD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((spconv_fwd_kernel<NTW, PD>), grid, dim3(256), lds, stream, in, packed, blk_src, blk_meta, blk_off,
                     K, n_dst, n_wtiles, cin, nt_total, p.taps_per_split, p.taps_per_slab, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

}  // namespace

extern "C" size_t gpn_spconv_fwd_ws_bytes(int K, int64_t n_dst, int cin, int cout) {
  if (n_dst <= 0 || cin < 16 || cout < 16) return 0;
  const FwdPlan p = plan_fwd(K, n_dst, cin, cout);
  return p.splits > 1 ? gpn::align_up((size_t)p.splits * n_dst * cout * sizeof(float)) : 0;
}

extern "C" int gpn_spconv_fwd(const float* in, const float* packed_w, const int32_t* blk_src,
                              const int32_t* blk_meta, const int32_t* blk_off, int K, int64_t n_dst, int tm, int cin,
                              int cout, float* out, void* ws, size_t ws_bytes, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(K >= 1 && 2 * K <= kMaxBlocks && n_dst >= 0 && tm == 1);
  GPN_CHECK_ARG(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0);
  if (n_dst == 0) return GPN_OK;
  GPN_CHECK_ARG(in && packed_w && blk_src && blk_meta && blk_off && out);
  const int nt = cout / 16;
  const FwdPlan p = plan_fwd(K, n_dst, cin, cout);
  GPN_CHECK_ARG((cin / 16) * p.ntw * 1024 <= kSlabBytes);
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
    // prefetch depth: deep for narrow layers (little work per step), shallower when a step is 64 channels wide
    const bool narrow = cin <= 32;
#define GPN_FWD(NTW) \
  rc = narrow ? launch_fwd<NTW, 8>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, target, stream) \
              : launch_fwd<NTW, 4>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, target, stream)
    switch (p.ntw) {
      case 1: GPN_FWD(1); break;
      case 2: GPN_FWD(2); break;
      case 3: GPN_FWD(3); break;
      default: GPN_FWD(4); break;
    }
#undef GPN_FWD
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
