// spconv_fwd.hip — the fused gather-MFMA-scatter sparse convolution (forward and dgrad launches) for gfx950.
//
// Workgroup = 256 threads = 4 waves.  Wave w of workgroup (x, y, z) owns
//     32 destination rows (tile 4x + w)  x  NTW 16-wide output-column tiles (group y)  x  taps [z*TS, (z+1)*TS).
// Its fp32 accumulators live in a private LDS tile; the output is written once (or, when taps are split across
// workgroups for small layers, once per split into a partial buffer that a fixed-order reduction sums).
//
// The layers of this network are small, so what limits them is dependent-load latency, not flops.  The kernel is
// therefore organised around keeping gathers in flight:
//   * taps are processed in stages of as many whole taps as fit a 16 KiB LDS weight slab (shared by the 4 waves;
//     B operands are conflict-free ds_read_b128 of pre-packed 1 KiB wave fragments);
//   * per stage a wave copies its block entries (source row, local destination row, tap) into LDS with coalesced
//     loads, so a gathered row's address never waits on a global load inside the pipeline;
//   * gathered rows (MFMA A operands) are streamed global -> LDS with the CDNA LDS-DMA (global_load_lds_dwordx4,
//     per-lane source address = a whole 64-byte piece of a gathered row per 4 lanes) into a PD-deep ring, PD steps
//     ahead of their use; the ring is drained with COUNTED s_waitcnt vmcnt((PD-1)*CW), so PD-1 steps of gathers
//     stay in flight under every MFMA group (the compiler cannot count loads it placed behind lane-divergent
//     branches and falls back to vmcnt(0) — measured: 2 us per step — hence the explicit DMA + explicit counts);
//   * v_mfma_f32_16x16x4_f32 does the per-rule dense contraction (exact fp32 == an fmaf chain); summation order is
//     fixed (tap-major), so results are deterministic.
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlabBytes = 16 * 1024;
constexpr int kMaxTapsPerSlab = 16;
constexpr int kMaxBlocks = 2 * kMaxTapsPerSlab;  // <= 2 blocks per (32-row tile, tap)

#define GPN_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 0xF) | ((((N) >> 4) & 0x3) << 14) | (0x7 << 4) | (0xF << 8))

template <int NTW, int CW, int PD>
struct FwdLds {
  static constexpr int LDW = NTW * 16 + 16;  // +16 floats: rows an odd distance apart land on disjoint bank halves
  static constexpr size_t slab = 0;
  static constexpr size_t acc = slab + kSlabBytes;
  static constexpr size_t ring = acc + (size_t)4 * 32 * LDW * 4;
  static constexpr size_t ent_src = ring + (size_t)4 * PD * CW * 1024;
  static constexpr size_t ent_dst = ent_src + (size_t)4 * kMaxBlocks * 16 * 4;
  static constexpr size_t ent_tap = ent_dst + (size_t)4 * kMaxBlocks * 16;
  static constexpr size_t total = ent_tap + (size_t)4 * kMaxBlocks;
};

template <int NTW, int CW, int PD>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ in, const float* __restrict__ packed, const int32_t* __restrict__ blk_src,
    const int32_t* __restrict__ blk_meta, const int32_t* __restrict__ blk_off, int K, int64_t n_dst, int64_t n_wtiles,
    int cin, int nt_total, int taps_per_split, int taps_per_slab, float* __restrict__ out) {
  using L = FwdLds<NTW, CW, PD>;
  constexpr int LDW = L::LDW;
  constexpr int ROWS = 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* slab = reinterpret_cast<f32x4*>(smem_raw + L::slab);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  float* acc_lds = reinterpret_cast<float*>(smem_raw + L::acc) + (size_t)wave * ROWS * LDW;
  f32x4* ring = reinterpret_cast<f32x4*>(smem_raw + L::ring) + (size_t)wave * PD * CW * 64;
  int32_t* ent_src = reinterpret_cast<int32_t*>(smem_raw + L::ent_src) + wave * kMaxBlocks * 16;
  uint8_t* ent_dst = reinterpret_cast<uint8_t*>(smem_raw + L::ent_dst) + wave * kMaxBlocks * 16;
  uint8_t* ent_tap = reinterpret_cast<uint8_t*>(smem_raw + L::ent_tap) + wave * kMaxBlocks;

  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  const bool active = w < n_wtiles;
  const int nt0 = blockIdx.y * NTW;
  const int ntw = (nt_total - nt0 < NTW) ? (nt_total - nt0) : NTW;
  const int k_lo = blockIdx.z * taps_per_split;
  const int k_hi = (k_lo + taps_per_split < K) ? (k_lo + taps_per_split) : K;
  const int64_t row0 = w * ROWS;
  const int cout = nt_total * 16;
  const int CB = cin >> 4;
  const int NCH = CB / CW;  // CW divides CB (host guarantees)
  const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(packed);

  for (int e = lane * 4; e < ROWS * LDW; e += 64 * 4) *reinterpret_cast<f32x4*>(acc_lds + e) = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int ks = k_lo; ks < k_hi; ks += taps_per_slab) {
    const int ke = (ks + taps_per_slab < k_hi) ? (ks + taps_per_slab) : k_hi;
    if (ks != k_lo) __syncthreads();  // every wave is done with the previous slab

    // ---- (a) this wave's block entries of taps [ks, ke) -> LDS -------------------------------------------------
    int32_t b_lo = 0, nblk = 0;
    if (active) {
      b_lo = blk_off[w * K + ks];
      nblk = blk_off[w * K + ke] - b_lo;
    }
    for (int e = lane; e < nblk * 16; e += 64) {
      const int32_t s = blk_src[(int64_t)b_lo * 16 + e];
      const int32_t m = blk_meta[(int64_t)b_lo * 16 + e];
      ent_src[e] = s < 0 ? 0 : s;  // padding lanes gather row 0: their D rows are never accumulated
      ent_dst[e] = m < 0 ? (uint8_t)255 : (uint8_t)(m & 0xff);
      if ((e & 15) == 0) ent_tap[e >> 4] = (uint8_t)((m >> 8) - ks);  // entry 0 of a block is always valid
    }

    // ---- (b) weights of taps [ks, ke) -> LDS slab ([tap][cb][nt][lane] float4), <= 4 float4 per thread ----------
    {
      const int limit = (ke - ks) * CB * NTW * 64;  // float4 count
      f32x4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int q = j * 256 + tid;
        q = q < limit ? q : limit - 1;
        const int p = q >> 6;
        int nt = p % NTW;
        const int rest = p / NTW;  // (tap - ks) * CB + cb
        nt = nt < ntw ? nt : 0;
        r[j] = pw[((int64_t)(ks * CB + rest) * nt_total + nt0 + nt) * 64 + (q & 63)];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = j * 256 + tid;
        if (q < limit) slab[q] = r[j];
      }
    }
    __syncthreads();

    const int nsteps = nblk * NCH;
    if (nsteps > 0) {
      // ---- (c) prime the gathered-row ring ------------------------------------------------------------------
      auto issue_a = [&](int t) {
        const int tt = t < nsteps ? t : nsteps - 1;  // past the end: harmless duplicate, keeps the DMA count uniform
        const int blk = tt / NCH;
        const int ch = tt - blk * NCH;
        const int32_t src = ent_src[blk * 16 + i16];
        const float* gp = in + (int64_t)src * cin + ch * (CW * 16) + 4 * g;
        f32x4* slot = ring + (t % PD) * (CW * 64);
#pragma unroll
        for (int c = 0; c < CW; ++c)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + c * 16),
                                           (__attribute__((address_space(3))) void*)(slot + c * 64), 16, 0, 0);
      };
      for (int u = 0; u < PD; ++u) issue_a(u);

      // ---- (d) steps ------------------------------------------------------------------------------------------
      for (int t = 0; t < nsteps; ++t) {
        const int blk = t / NCH;
        const int ch = t - blk * NCH;
        const int tap = ent_tap[blk];
        GPN_WAIT_VMCNT((PD - 1) * CW);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* slot = ring + (t % PD) * (CW * 64) + lane;
        f32x4 a[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) a[c] = slot[c * 64];
        const f32x4* sb = slab + ((int64_t)(tap * CB + ch * CW) * NTW) * 64 + lane;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            if (nt < ntw) {
              const f32x4 bf = sb[(c * NTW + nt) * 64];
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, bf.x, acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, bf.y, acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, bf.z, acc[nt], 0, 0, 0);
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, bf.w, acc[nt], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_a(t + PD);  // refill the slot just consumed
        if (ch == NCH - 1) {
          // D[row = 4g + r][col = i16] belongs to pair 4g + r of the block; the 4 rows are distinct destinations
          int row[4];
          float v[4][NTW];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            row[r] = ent_dst[blk * 16 + 4 * g + r];
            const int rr = row[r] == 255 ? 0 : row[r];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) v[r][nt] = acc_lds[rr * LDW + nt * 16 + i16];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (row[r] != 255) {
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                if (nt < ntw) acc_lds[row[r] * LDW + nt * 16 + i16] = v[r][nt] + acc[nt][r];
            }
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      // ---- (e) drain the tail DMAs before the ring / entries are reused --------------------------------------
      GPN_WAIT_VMCNT(0);
    }
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
  int ntw, cw, splits, taps_per_split, taps_per_slab;
};

FwdPlan plan_fwd(int K, int64_t n_dst, int cin, int cout) {
  const int nt = cout / 16, CB = cin / 16;
  const int64_t row_wgs = gpn::cdiv(gpn::cdiv(n_dst, GPN_TILE_ROWS), 4);
  FwdPlan p;
  p.cw = (CB % 4 == 0) ? 4 : (CB % 2 == 0) ? 2 : 1;
  // column tiles per workgroup: as many as possible (rows are re-gathered once per column group) with >= 512
  // workgroups in flight; one tap's weights (CB * ntw KiB) must fit the slab
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
  if (tps > kMaxTapsPerSlab) tps = kMaxTapsPerSlab;
  p.taps_per_slab = tps;
  return p;
}

template <int NTW, int CW, int PD>
int launch_fwd(const FwdPlan& p, const float* in, const float* packed, const int32_t* blk_src, const int32_t* blk_meta,
               const int32_t* blk_off, int K, int64_t n_dst, int cin, int nt_total, float* out, hipStream_t stream) {
  const int64_t n_wtiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const dim3 grid((unsigned)gpn::cdiv(n_wtiles, 4), (unsigned)gpn::cdiv(nt_total, NTW), (unsigned)p.splits);
  const size_t lds = FwdLds<NTW, CW, PD>::total;
  static bool attr_set = false;
  if (!attr_set) {
    GPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_fwd_kernel<NTW, CW, PD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((spconv_fwd_kernel<NTW, CW, PD>), grid, dim3(256), lds, stream, in, packed, blk_src, blk_meta,
                     blk_off, K, n_dst, n_wtiles, cin, nt_total, p.taps_per_split, p.taps_per_slab, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int NTW>
int dispatch_cw(const FwdPlan& p, const float* in, const float* packed, const int32_t* blk_src, const int32_t* blk_meta,
                const int32_t* blk_off, int K, int64_t n_dst, int cin, int nt_total, float* out, hipStream_t stream) {
  // ring depth: 6 / 3 / 2 steps of 1 / 2 / 4 KiB per wave
  switch (p.cw) {
    case 1: return launch_fwd<NTW, 1, 6>(p, in, packed, blk_src, blk_meta, blk_off, K, n_dst, cin, nt_total, out, stream);
    case 2: return launch_fwd<NTW, 2, 3>(p, in, packed, blk_src, blk_meta, blk_off, K, n_dst, cin, nt_total, out, stream);
    default: return launch_fwd<NTW, 4, 2>(p, in, packed, blk_src, blk_meta, blk_off, K, n_dst, cin, nt_total, out, stream);
  }
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
  GPN_CHECK_ARG(K >= 1 && K <= 255 && n_dst >= 0 && tm == 1);
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
    switch (p.ntw) {
      case 1: rc = dispatch_cw<1>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, target, stream); break;
      case 2: rc = dispatch_cw<2>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, target, stream); break;
      case 3: rc = dispatch_cw<3>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, target, stream); break;
      default: rc = dispatch_cw<4>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, target, stream); break;
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
