// bn_stats.h — BatchNorm column sums accumulated by the PRODUCING conv launch (round 3).
//
// network/backbone.py:40-49 is conv -> BatchNorm1d -> ReLU everywhere; as separate launches the statistics of a BatchNorm
// are a full extra pass over the conv's output (forward: sum x, sum x^2; backward: sum g, sum g * xhat over the gradient the
// following dgrad conv has just written).  The conv kernels own those rows in registers when they store them, so their
// epilogue adds the column sums of its 16-row tile to a small slab instead, and the BatchNorm keeps only its apply pass
// (bn.hip: gpn::bn_fwd_train_fused / bn_bwd_fused): 2 launches per conv + BatchNorm instead of 3, in both directions.
//
// Thousands of waves contribute to one channel, so the sums must not depend on arrival order: they are accumulated as
// 64-bit FIXED-POINT integers with global atomics (integer addition is associative: bitwise deterministic, no ordering
// protocol between workgroups, no fences).  A value v is split into a coarse word hi = rint(v 2^H) and a fine word
// lo = rint((v - hi 2^-H) 2^L), |lo| <= 2^(L-H-1):
//   sum x, sum g, sum g*xhat : H = 20, L = 50 -> |total| < 8.8e12, resolution 9e-16 per contribution
//   sum x^2                  : H = 10, L = 40 -> total < 9.0e15 (rms 2.5e5 over 144k rows), resolution 9e-13
// (a contribution is the sum over the 16 rows of a tile, accumulated in double; the fine words of <= 2^20 contributions
// cannot overflow).
// To keep same-address contention low the contributions of a launch are spread over kStatSlots slot sets
// (slot = wave index mod 32); the apply pass folds the 32 x 4 words per channel (exact integer sums, then two int -> double
// conversions).  Slabs live in the executor's workspace and are zeroed by one memset per pass.
#pragma once
#include "gpn_common.h"

namespace gpn {

struct StatScale {
  int h0, l0, h1, l1;  // (H, L) of the first and of the second sum
};
__host__ __device__ constexpr StatScale kStatScaleFwd() { return StatScale{20, 50, 10, 40}; }  // sum x, sum x^2
__host__ __device__ constexpr StatScale kStatScaleBwd() { return StatScale{20, 50, 20, 50}; }  // sum g, sum g * xhat

// sum over the four lanes (g = lane >> 4) that hold the same column: fixed order ((g0 + g1) + (g2 + g3)).  The epilogues
// accumulate a tile's column sums in DOUBLE (products of fp32 values are exact there): with fp32 tile sums the variance
// E[x^2] - mean^2 of a small level (338 rows) lost 3-4 digits on channels whose mean is large against their spread, and the
// gradients of the levels above moved by 1e-2 (tools: GPN_BN_FUSE_MIN_ROWS A/B against the oracle)
__device__ __forceinline__ double stat_reduce_g(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// all 64 lanes call this with the two column totals (identical in the 4 lanes g of a column): lane g adds one of the four
// fixed-point words of column `col` to slot set `slot`.  slab [kStatSlots][4][C]: words (sum0 hi, sum0 lo, sum1 hi, sum1 lo).
template <bool BWD>
__device__ __forceinline__ void stat_add(unsigned long long* __restrict__ slab, int C, int slot, int col, int g, double v0,
                                         double v1) {
  constexpr StatScale sc = BWD ? kStatScaleBwd() : kStatScaleFwd();
  const double v = g < 2 ? v0 : v1;
  const int H = g < 2 ? sc.h0 : sc.h1, L = g < 2 ? sc.l0 : sc.l1;
  const long long hi = __double2ll_rn(ldexp(v, H));
  const long long lo = __double2ll_rn(ldexp(v - ldexp((double)hi, -H), L));
  atomicAdd(slab + ((size_t)slot * 4 + g) * C + col, (unsigned long long)((g & 1) ? lo : hi));
}

// the two totals of channel c from a slab (exact integer sums over the slot sets)
template <bool BWD>
__device__ __forceinline__ void stat_fold(const unsigned long long* __restrict__ slab, int C, int c, double& t0, double& t1) {
  constexpr StatScale sc = BWD ? kStatScaleBwd() : kStatScaleFwd();
  long long w[4] = {0, 0, 0, 0};
  for (int s = 0; s < kStatSlots; ++s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] += (long long)slab[((size_t)s * 4 + q) * C + c];
  }
  t0 = ldexp((double)w[0], -sc.h0) + ldexp((double)w[1], -sc.l0);
  t1 = ldexp((double)w[2], -sc.h1) + ldexp((double)w[3], -sc.l1);
}

}  // namespace gpn
