// spconv_fwd.hip — the fused gather-MFMA-scatter sparse convolution (forward and dgrad launches) for gfx950.
//
// Workgroup = 256 threads = 4 waves.  Wave w of workgroup (x, y, z) owns
//     32 destination rows (tile 4x + w)  x  NTW 16-wide output-column tiles (group y)  x  taps [z*TS, (z+1)*TS).
// Its fp32 accumulators live in a private LDS tile; the output is written once (or, when taps are split across
// workgroups for small layers, once per split into a partial buffer that a fixed-order reduction sums).
//
// The 4 waves walk stages (tap k, chunk of CW 16-channel blocks) in lock-step, so a stage's weight slab
// (CW x NTW pre-packed 1 KiB MFMA-B fragments) is fetched from L2 once per workgroup and shared through a
// double-buffered LDS slab (conflict-free ds_read_b128), one barrier per stage.  Everything a stage needs from global
// memory is requested one or more stages earlier and lands in registers while the previous stage's MFMAs run:
//     next stage's slab (global -> regs -> LDS after this stage's compute),
//     next stage's gathered rows = MFMA A operands: lane (i = l&15, g = l>>4) loads channels [16cb+4g, 16cb+4g+4) of
//         pair i's source row with one 16-byte load, i.e. whole contiguous 64-byte pieces of each gathered row,
//     the block entries (source row, local destination row) of the tap three taps ahead.
// All of those loads are unconditional and branch-free (indices are clamped, padding lanes gather row 0 and their MFMA
// output rows are simply never accumulated): the compiler can then count them and emits partial s_waitcnt vmcnt(N)
// instead of vmcnt(0) — with lane-divergent guards around the loads every stage exposed a full memory latency.
// v_mfma_f32_16x16x4_f32 does the per-rule dense contraction (exact fp32 == an fmaf chain); summation order is fixed
// (tap-major), so results are deterministic.
#include "gpn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NTW, int CW>
struct FwdCfg {
  static constexpr int LDW = NTW * 16 + 16;       // +16 floats: rows an odd distance apart land on disjoint bank halves
  static constexpr int SLAB_V4 = CW * NTW * 64;   // float4 per slab buffer: [CW][NTW][64 lanes]
  static constexpr int NS = (SLAB_V4 + 255) / 256;  // slab float4 per thread
  static constexpr size_t lds_bytes = (size_t)2 * SLAB_V4 * 16 + (size_t)4 * 32 * LDW * 4;
};

template <int NTW, int CW>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ in, const float* __restrict__ packed, const int32_t* __restrict__ blk_src,
    const int32_t* __restrict__ blk_meta, const int32_t* __restrict__ blk_off, int K, int64_t n_dst, int64_t n_wtiles,
    int cin, int nt_total, int taps_per_split, int64_t entry_cap, float* __restrict__ out) {
  using C = FwdCfg<NTW, CW>;
  constexpr int LDW = C::LDW, SLAB_V4 = C::SLAB_V4, NS = C::NS;
  constexpr int ROWS = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* slab = reinterpret_cast<f32x4*>(smem);          // [2][SLAB_V4]
  float* acc_all = smem + 2 * SLAB_V4 * 4;               // [4 waves][ROWS][LDW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  float* acc_lds = acc_all + (size_t)wave * ROWS * LDW;

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
  const int n_stages = (k_hi - k_lo) * NCH;
  const f32x4* __restrict__ pw = reinterpret_cast<const f32x4*>(packed);

  for (int e = lane * 4; e < ROWS * LDW; e += 64 * 4) *reinterpret_cast<f32x4*>(acc_lds + e) = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-tap block ranges of this wave's tile: lane l holds blk_off[w*K + min(l, K)]  (K <= 63)
  int32_t boff = 0;
  if (active) boff = blk_off[w * K + (lane < K ? lane : K)];

  // ---- branch-free loaders ---------------------------------------------------------------------------------------
  auto load_slab = [&](int stage, f32x4 (&r)[NS]) {
    int st = stage < n_stages ? stage : n_stages - 1;  // past the end: harmless duplicate
    const int k = k_lo + st / NCH, ch = st % NCH;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      int q = j * 256 + tid;
      q = q < SLAB_V4 ? q : SLAB_V4 - 1;
      const int p = q >> 6;                 // piece = c * NTW + nt
      const int c = p / NTW;
      int nt = p - c * NTW;
      nt = nt < ntw ? nt : 0;
      r[j] = pw[((int64_t)(k * CB + ch * CW + c) * nt_total + nt0 + nt) * 64 + (q & 63)];
    }
  };
  auto store_slab = [&](int buf, const f32x4 (&r)[NS]) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int q = j * 256 + tid;
      if (q < SLAB_V4) slab[buf * SLAB_V4 + q] = r[j];
    }
  };
  // entries of tap k: 2 blocks x (src, dst) per 16-lane group; src = -1 / dst = 255 where there is no pair
  auto load_ent = [&](int k, int32_t (&src)[2], int32_t (&dst)[2]) {
    const int kk = k < K ? k : K;  // k >= K: empty range [boff[K], boff[K])
    const int32_t b0 = __shfl(boff, kk, 64);
    const int32_t b1 = __shfl(boff, (kk + 1 < K ? kk + 1 : K), 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t e = (int64_t)(b0 + j) * 16 + i16;
      e = e < entry_cap ? e : entry_cap - 1;
      const int32_t s = blk_src[e];
      const int32_t m = blk_meta[e];
      const bool ok = (b0 + j < b1) && (m >= 0);
      src[j] = ok ? s : -1;
      dst[j] = ok ? (m & 0xff) : 255;
    }
  };
  auto load_a = [&](const int32_t (&src)[2], int ch, f32x4 (&a)[2][CW]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int32_t s = src[j] < 0 ? 0 : src[j];  // padding lanes gather row 0; their output rows are never used
      const f32x4* arow = reinterpret_cast<const f32x4*>(in + (int64_t)s * cin + ch * (CW * 16) + 4 * g);
#pragma unroll
      for (int c = 0; c < CW; ++c) a[j][c] = arow[c * 4];
    }
  };

  // ---- software pipeline: prefetch distance D stages for slabs and gathered rows, 2D stages for entries ----------
  // (a stage is shorter than a memory round trip, and the 4 waves of a workgroup advance in lock-step, so the only
  //  way to keep the workgroup from paying one full latency per stage is to have several stages of loads in flight.
  //  Ring slots are compile-time: the stage loop is unrolled by E = 2D and the stage count padded to a multiple of E
  //  with empty stages, which keeps every load unconditional.)
  constexpr int D = (CW == 4) ? 2 : 4;
  constexpr int E = 2 * D;
  const int n_pad = (n_stages + E - 1) / E * E;
  auto tap_of = [&](int st) { return st < n_stages ? k_lo + st / NCH : K; };  // K -> empty entry range
  auto ch_of = [&](int st) { return st < n_stages ? st % NCH : 0; };

  int32_t esrc[E][2], edst[E][2];
  f32x4 areg[D][2][CW];
  f32x4 sreg[D][NS];
#pragma unroll
  for (int u = 0; u < E; ++u) load_ent(tap_of(u), esrc[u], edst[u]);
#pragma unroll
  for (int u = 0; u < D; ++u) load_slab(u, sreg[u]);
#pragma unroll
  for (int u = 0; u < D; ++u) load_a(esrc[u], ch_of(u), areg[u]);
  store_slab(0, sreg[0]);
  load_slab(D, sreg[0]);
  __syncthreads();

  f32x4 acc[2][NTW];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int s0 = 0; s0 < n_pad; s0 += E) {
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int stage = s0 + u;
      const bool last_ch = (ch_of(stage) == NCH - 1);
      // contraction of the current stage (blocks that do not exist are skipped: uniform branches, no memory ops inside)
      const bool have0 = __builtin_amdgcn_readfirstlane(edst[u][0]) != 255;  // entry 0 of an existing block is valid
      const bool have1 = __builtin_amdgcn_readfirstlane(edst[u][1]) != 255;
      if (have0) {
        const f32x4* sb = slab + (stage & 1) * SLAB_V4 + lane;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            if (nt < ntw) {
              const f32x4 bf = sb[(c * NTW + nt) * 64];
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][0][c].x, bf.x, acc[0][nt], 0, 0, 0);
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][0][c].y, bf.y, acc[0][nt], 0, 0, 0);
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][0][c].z, bf.z, acc[0][nt], 0, 0, 0);
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][0][c].w, bf.w, acc[0][nt], 0, 0, 0);
              if (have1) {
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][1][c].x, bf.x, acc[1][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][1][c].y, bf.y, acc[1][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][1][c].z, bf.z, acc[1][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[u % D][1][c].w, bf.w, acc[1][nt], 0, 0, 0);
              }
            }
          }
        }
        if (last_ch) {
          // D[row = 4g + r][col = i16] of block j belongs to its pair 4g + r (lane 4g + r holds that pair's local
          // destination row); the rows of one tap are distinct, so reads can be batched ahead of the writes
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (j == 1 && !have1) break;
            int row[4];
            float v[4][NTW];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              row[r] = __shfl(edst[u][j], 4 * g + r, 64);
              const int rr = row[r] == 255 ? 0 : row[r];
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt) v[r][nt] = acc_lds[rr * LDW + nt * 16 + i16];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (row[r] != 255) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                  if (nt < ntw) acc_lds[row[r] * LDW + nt * 16 + i16] = v[r][nt] + acc[j][nt][r];
              }
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      // hand the next stage its slab, then refill the ring slots this stage freed (all loads unconditional)
      store_slab((stage + 1) & 1, sreg[(u + 1) % D]);
      load_slab(stage + 1 + D, sreg[(u + 1) % D]);
      load_a(esrc[(u + D) % E], ch_of(stage + D), areg[u % D]);
      load_ent(tap_of(stage + E), esrc[u], edst[u]);
      __syncthreads();
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
  int ntw, cw, splits, taps_per_split;
};

FwdPlan plan_fwd(int K, int64_t n_dst, int cin, int cout) {
  const int nt = cout / 16, CB = cin / 16;
  const int64_t row_wgs = gpn::cdiv(gpn::cdiv(n_dst, GPN_TILE_ROWS), 4);
  FwdPlan p;
  p.cw = (CB % 4 == 0) ? 4 : (CB % 2 == 0) ? 2 : 1;
  // column tiles per workgroup: as many as possible (rows are re-gathered once per column group) while keeping
  // >= 512 workgroups in flight
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

template <int NTW, int CW>
int launch_fwd(const FwdPlan& p, const float* in, const float* packed, const int32_t* blk_src, const int32_t* blk_meta,
               const int32_t* blk_off, int K, int64_t n_dst, int cin, int nt_total, int64_t entry_cap, float* out,
               hipStream_t stream) {
  const int64_t n_wtiles = gpn::cdiv(n_dst, GPN_TILE_ROWS);
  const dim3 grid((unsigned)gpn::cdiv(n_wtiles, 4), (unsigned)gpn::cdiv(nt_total, NTW), (unsigned)p.splits);
  const size_t lds = FwdCfg<NTW, CW>::lds_bytes;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    GPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spconv_fwd_kernel<NTW, CW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((spconv_fwd_kernel<NTW, CW>), grid, dim3(256), lds, stream, in, packed, blk_src, blk_meta, blk_off,
                     K, n_dst, n_wtiles, cin, nt_total, p.taps_per_split, entry_cap, out);
  GPN_CHECK_LAUNCH();
  return GPN_OK;
}

template <int NTW>
int dispatch_cw(const FwdPlan& p, const float* in, const float* packed, const int32_t* blk_src, const int32_t* blk_meta,
                const int32_t* blk_off, int K, int64_t n_dst, int cin, int nt_total, int64_t entry_cap, float* out,
                hipStream_t stream) {
  switch (p.cw) {
    case 1: return launch_fwd<NTW, 1>(p, in, packed, blk_src, blk_meta, blk_off, K, n_dst, cin, nt_total, entry_cap, out, stream);
    case 2: return launch_fwd<NTW, 2>(p, in, packed, blk_src, blk_meta, blk_off, K, n_dst, cin, nt_total, entry_cap, out, stream);
    default: return launch_fwd<NTW, 4>(p, in, packed, blk_src, blk_meta, blk_off, K, n_dst, cin, nt_total, entry_cap, out, stream);
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
  GPN_CHECK_ARG(K >= 1 && K <= 63 && n_dst >= 0 && tm == 1);
  GPN_CHECK_ARG(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0);
  if (n_dst == 0) return GPN_OK;
  GPN_CHECK_ARG(in && packed_w && blk_src && blk_meta && blk_off && out);
  const int nt = cout / 16;
  const FwdPlan p = plan_fwd(K, n_dst, cin, cout);
  const int64_t entry_cap = gpn_rulebook_blocks_capacity(K, n_dst, 1) * 16;
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
      case 1: rc = dispatch_cw<1>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, entry_cap, target, stream); break;
      case 2: rc = dispatch_cw<2>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, entry_cap, target, stream); break;
      case 3: rc = dispatch_cw<3>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, entry_cap, target, stream); break;
      default: rc = dispatch_cw<4>(p, in, packed_w, blk_src, blk_meta, blk_off, K, n_dst, cin, nt, entry_cap, target, stream); break;
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
