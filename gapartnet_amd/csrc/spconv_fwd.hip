// spconv_fwd.hip — the fused gather-MFMA-scatter sparse convolution (forward and dgrad launches) for gfx950.
//
// Workgroup = 256 threads = 4 waves = 4 adjacent 32-row destination tiles x one column group (NTW 16-wide tiles).
// The 4 waves walk the taps in lock-step so that the tap's weight slab is fetched from L2 ONCE per workgroup and
// shared through LDS (double-buffered, one barrier per stage) instead of once per 16-pair block:
//
//   stage (tap k, 64-channel chunk ch):
//     - all threads: global -> registers of the NEXT stage's slab  (<= 4 channel blocks x NTW tiles x 1 KiB)
//     - each wave  : global -> registers of its NEXT stage's gathered rows (A operands; <= 2 blocks x 4 x 16 B / lane)
//                    and, three taps ahead, its block entries (src row, local dst row)
//     - each wave  : contraction of the CURRENT stage: B fragments by ds_read_b128 from the slab, A from registers,
//                    v_mfma_f32_16x16x4_f32; the 16 x (16 NTW) result is added to the wave's private fp32 tile in LDS
//     - all threads: registers -> LDS of the next slab; __syncthreads()
//
// Every dependent global load (entries -> rows) is issued at least one full stage before its use, the output tile
// is written to HBM exactly once, and the summation order is fixed (tap-major), so results are deterministic.
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NTW>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ in, const float* __restrict__ packed, const int32_t* __restrict__ blk_src,
    const int32_t* __restrict__ blk_meta, const int32_t* __restrict__ blk_off, int K, int64_t n_dst, int64_t n_wtiles,
    int cin, int nt_total, float* __restrict__ out) {
  constexpr int LDW = NTW * 16 + 16;   // +16 floats: rows an odd distance apart land on disjoint bank halves
  constexpr int ROWS = 32;
  constexpr int SLAB_V4 = 4 * NTW * 64;  // float4 per slab buffer: [4 channel blocks][NTW tiles][64 lanes]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* slab = reinterpret_cast<f32x4*>(smem);                 // [2][SLAB_V4]
  float* acc_all = smem + 2 * SLAB_V4 * 4;                       // [4 waves][ROWS][LDW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  float* acc_lds = acc_all + (size_t)wave * ROWS * LDW;

  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  const bool active = w < n_wtiles;
  const int nt0 = blockIdx.y * NTW;
  const int ntw = (nt_total - nt0 < NTW) ? (nt_total - nt0) : NTW;
  const int64_t row0 = w * ROWS;
  const int cout = nt_total * 16;
  const int CB = cin >> 4;
  const int NCH = (CB + 3) >> 2;
  const int n_stages = K * NCH;
  const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(packed);

  for (int e = lane * 4; e < ROWS * LDW; e += 64 * 4) *reinterpret_cast<f32x4*>(acc_lds + e) = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-tap block ranges of this wave's tile, one per lane (K <= 63)
  int32_t boff = 0;
  if (active) boff = blk_off[w * K + (lane < K ? lane : K)];

  // ---- helpers -------------------------------------------------------------------------------------------------
  auto load_slab = [&](int stage, f32x4 (&r)[NTW]) {
    const int k = stage / NCH, ch = stage - k * NCH;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int q = j * 256 + tid;  // float4 index inside the slab: piece p = q >> 6 -> (c = p / NTW, nt = p % NTW)
      const int p = q >> 6;
      const int c = p / NTW, nt = p - c * NTW;
      const int cb = ch * 4 + c;
      r[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (cb < CB && nt < ntw) r[j] = pw[((int64_t)(k * CB + cb) * nt_total + nt0 + nt) * 64 + (q & 63)];
    }
  };
  auto store_slab = [&](int buf, const f32x4 (&r)[NTW]) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) slab[buf * SLAB_V4 + j * 256 + tid] = r[j];
  };
  auto load_ent = [&](int k, int32_t (&src)[2], int32_t (&meta)[2]) {
    src[0] = src[1] = -1;
    meta[0] = meta[1] = -1;
    if (k < K) {
      const int32_t b0 = __shfl(boff, k, 64), b1 = __shfl(boff, k + 1, 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (b0 + j < b1) {
          src[j] = blk_src[(int64_t)(b0 + j) * 16 + i16];
          meta[j] = blk_meta[(int64_t)(b0 + j) * 16 + i16];
        }
      }
    }
  };
  auto load_a = [&](const int32_t (&src)[2], int ch, f32x4 (&a)[2][4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* arow = in + (int64_t)src[j] * cin + ch * 64 + 4 * g;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (src[j] >= 0 && ch * 4 + c < CB) a[j][c] = *reinterpret_cast<const f32x4*>(arow + c * 16);
      }
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  f32x4 slab_r[NTW];
  load_slab(0, slab_r);
  int32_t src_c[2], meta_c[2], src_n[2], meta_n[2], src_nn[2], meta_nn[2];
  load_ent(0, src_c, meta_c);
  load_ent(1, src_n, meta_n);
  load_ent(2, src_nn, meta_nn);
  f32x4 a_cur[2][4], a_nxt[2][4];
  load_a(src_c, 0, a_cur);
  store_slab(0, slab_r);
  __syncthreads();

  // ---- stages --------------------------------------------------------------------------------------------------
  int stage = 0;
  for (int k = 0; k < K; ++k) {
    for (int ch = 0; ch < NCH; ++ch, ++stage) {
      const bool last_ch = (ch == NCH - 1);
      const bool has_next = stage + 1 < n_stages;
      int32_t src_t[2], meta_t[2];
      if (has_next) load_slab(stage + 1, slab_r);
      if (!last_ch) load_a(src_c, ch + 1, a_nxt);
      else load_a(src_n, 0, a_nxt);
      if (last_ch) load_ent(k + 3, src_t, meta_t);

      // contraction of the current stage
      const bool have0 = __builtin_amdgcn_readfirstlane(meta_c[0]) >= 0;  // entry 0 of an existing block is valid
      const bool have1 = __builtin_amdgcn_readfirstlane(meta_c[1]) >= 0;
      if (have0) {
        f32x4 acc[2][NTW];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4* sb = slab + (stage & 1) * SLAB_V4 + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (ch * 4 + c < CB) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              if (nt < ntw) {
                const f32x4 bf = sb[(c * NTW + nt) * 64];
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0][c].x, bf.x, acc[0][nt], 0, 0, 0);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0][c].y, bf.y, acc[0][nt], 0, 0, 0);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0][c].z, bf.z, acc[0][nt], 0, 0, 0);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0][c].w, bf.w, acc[0][nt], 0, 0, 0);
                if (have1) {
                  acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1][c].x, bf.x, acc[1][nt], 0, 0, 0);
                  acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1][c].y, bf.y, acc[1][nt], 0, 0, 0);
                  acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1][c].z, bf.z, acc[1][nt], 0, 0, 0);
                  acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1][c].w, bf.w, acc[1][nt], 0, 0, 0);
                }
              }
            }
          }
        }
        // D[row = 4g + r][col = i16] of block j belongs to its pair 4g + r, whose local dst row lane (4g + r) holds
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j == 1 && !have1) break;
          const int dstl = meta_c[j] >= 0 ? (meta_c[j] & 0xff) : -1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = __shfl(dstl, 4 * g + r, 64);
            if (row >= 0) {
              float* dstp = acc_lds + row * LDW + i16;
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                if (nt < ntw) dstp[nt * 16] += acc[j][nt][r];
            }
          }
        }
      }

      if (has_next) store_slab((stage + 1) & 1, slab_r);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) a_cur[j][c] = a_nxt[j][c];
      if (last_ch) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          src_c[j] = src_n[j]; meta_c[j] = meta_n[j];
          src_n[j] = src_nn[j]; meta_n[j] = meta_nn[j];
          src_nn[j] = src_t[j]; meta_nn[j] = meta_t[j];
        }
      }
    }
  }

  // ---- write this wave's columns of its tile -------------------------------------------------------------------
  if (!active) return;
  const int64_t rows_here = (n_dst - row0 < ROWS) ? (n_dst - row0) : ROWS;
  const int v4 = ntw * 4;
  for (int e = lane; e < (int)rows_here * v4; e += 64) {
    const int r = e / v4, c4 = e - r * v4;
    *reinterpret_cast<f32x4*>(out + (row0 + r) * cout + nt0 * 16 + c4 * 4) =
        *reinterpret_cast<const f32x4*>(acc_lds + r * LDW + c4 * 4);
  }
}

template <int NTW>
int launch_fwd(const float* in, const float* packed, const int32_t* blk_src, const int32_t* blk_meta,
               const int32_t* blk_off, int K, int64_t n_dst, int cin, int nt_total, float* out, hipStream_t stream) {
  const int64_t n_wtiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const dim3 grid((unsigned)gpn::cdiv(n_wtiles, 4), (unsigned)gpn::cdiv(nt_total, NTW));
  const size_t lds = (size_t)(2 * 4 * NTW * 64 * 4 + 4 * 32 * (NTW * 16 + 16)) * sizeof(float);
  hipLaunchKernelGGL((spconv_fwd_kernel<NTW>), grid, dim3(256), lds, stream, in, packed, blk_src, blk_meta, blk_off, K,
                     n_dst, n_wtiles, cin, nt_total, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

// column tiles per workgroup: as many as possible (gathered rows are re-read once per column group) while keeping
// >= ~512 workgroups (256 CUs x 2) in flight
int pick_ntw(int nt_total, int64_t n_wtiles) {
  const int64_t row_wgs = gpn::cdiv(n_wtiles, 4);
  for (int ntw = 4; ntw > 1; --ntw) {
    if (ntw > nt_total) continue;
    if (row_wgs * gpn::cdiv(nt_total, ntw) >= 512) return ntw;
  }
  return 1;
}

}  // namespace

extern "C" int gpn_spconv_fwd(const float* in, const float* packed_w, const int32_t* blk_src,
                              const int32_t* blk_meta, const int32_t* blk_off, int K, int64_t n_dst, int tm, int cin,
                              int cout, float* out, gpn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPN_CHECK_ARG(K >= 1 && K <= 63 && n_dst >= 0 && tm == 1);
  GPN_CHECK_ARG(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0);
  if (n_dst == 0) return GPN_OK;
  GPN_CHECK_ARG(in && packed_w && blk_src && blk_meta && blk_off && out);
  const int nt = cout / 16;
  const int64_t n_wtiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const int ntw = pick_ntw(nt, n_wtiles);
  gpn::ProfScope prof(GPN_K_SPCONV_FWD, stream, 0.0, 4.0 * (double)n_dst * cout);
  switch (ntw) {
    case 1: return launch_fwd<1>(in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, out, stream);
    case 2: return launch_fwd<2>(in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, out, stream);
    case 3: return launch_fwd<3>(in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, out, stream);
    default: return launch_fwd<4>(in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, out, stream);
  }
}
