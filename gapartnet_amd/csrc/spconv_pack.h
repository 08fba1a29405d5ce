// spconv_pack.h — element mapping of the packed weight layout the fused conv kernel reads (shared by the single-weight
// packer in spconv.hip and the batched packer of the layer-program executor in net.hip).
//   packed[k][cb][nt][lane][s] = Wop[k][16cb + 4(lane>>4) + s][16nt + (lane&15)]
// where Wop is the stored weight, optionally transposed (dgrad) and tap-reversed (SubM dgrad); the stored weight is
// canonical [K][Cin][Cout] or, with GPN_LAYOUT_OKI, the spconv-2.x parameter [Cout][K][Cin].
#pragma once
#include "gpn_common.h"

namespace gpn {

__device__ __forceinline__ float packed_weight_element(const float* __restrict__ W, int K, int cin_w, int cout_w,
                                                       int flags, int64_t t) {
  const int cin = (flags & GPN_PACK_TRANSPOSE) ? cout_w : cin_w;
  const int cout = (flags & GPN_PACK_TRANSPOSE) ? cin_w : cout_w;
  const int s = (int)(t & 3);
  const int lane = (int)((t >> 2) & 63);
  int64_t r = t >> 8;
  const int NT = cout / 16, CB = cin / 16;
  const int nt = (int)(r % NT);
  r /= NT;
  const int cb = (int)(r % CB);
  r /= CB;
  const int k = (int)r;
  const int ci = cb * 16 + 4 * (lane >> 4) + s;
  const int co = nt * 16 + (lane & 15);
  const int kk = (flags & GPN_PACK_REVERSE) ? (K - 1 - k) : k;
  // element (tap kk, input channel wi, output channel wo) of the stored weight
  const int wi = (flags & GPN_PACK_TRANSPOSE) ? co : ci;
  const int wo = (flags & GPN_PACK_TRANSPOSE) ? ci : co;
  if (flags & GPN_LAYOUT_OKI) return W[((int64_t)wo * K + kk) * cin_w + wi];  // [Cout][K][Cin]
  return W[((int64_t)kk * cin_w + wi) * cout_w + wo];                        // [K][Cin][Cout]
}

}  // namespace gpn
