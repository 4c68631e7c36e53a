// Element-wise share arithmetic of the modulus conversions (SCI/src/BuildingBlocks/truncation.cpp:518-695), shared by
// convert.hip and by the kernels that run a conversion step inside a neighbouring device-resident step (fc_bn.hip).
#pragma once
#include "common.hpp"

namespace mmh {
constexpr int kAlice = 1;

__device__ __forceinline__ u64 bw_mask(int bw) { return bw >= 64 ? ~u64(0) : (u64(1) << bw) - 1; }
// truncation.cpp:533-541.  x may exceed p (it is an l-bit ring value): keep the reference's
// non-canonical results bit for bit.
__device__ __forceinline__ u64 secure_add(u64 x, u64 y, u64 p) { return x >= (p - y) ? x - (p - y) : x + y; }
__device__ __forceinline__ u64 secure_sub(u64 x, u64 y, u64 p) { return x >= y ? x - y : x + (p - y); }

// ring_to_field after the COT (truncation.cpp:551-554, :571 / :576): cv = (2^bw - 1) % p + 1, big = 2^(bw-2) % p
__device__ __forceinline__ u64 r2f_post_elem(int party, u64 tmp, uint8_t msb, u64 yy, u64 p, u64 cv, u64 big_mod_p) {
  if (party == kAlice) {
    const u64 t = secure_sub(0, yy, p);
    yy = msb ? secure_add(cv, t, p) : t;
  }
  u64 o = secure_sub(tmp, yy, p);
  if (party == kAlice) o = secure_sub(o, big_mod_p, p);
  return o;
}
// field -> ring before the COT (truncation.cpp:601-607 / :657-665): + p/4 on ALICE, msb = v >= p/2 + 1,
// correlation corr_value where the msb is clear
__device__ __forceinline__ u64 f2r_pre_elem(int party, u64 v, u64 p, bool &m) {
  const u64 big = p / 4;
  if (party == kAlice) v = v >= p - big ? v - p + big : v + big;
  m = v >= p / 2 + 1;
  return v;
}
}  // namespace mmh
