// Device-side BFV helpers shared by bfv.hip, conv.hip and fc.hip.
#pragma once
#include "common.hpp"

namespace mmh {
__device__ __forceinline__ u64 reduce128(u64 hi, u64 lo, const ModC &m) {
  return barrett128(hi, lo, m.q, m.barrett_hi, m.barrett_lo);
}
__device__ __forceinline__ u64 mulmod_c(u64 a, u64 b, const ModC &m) {
  return reduce128(mulhi64(a, b), a * b, m);
}
// Barrett for sums of products that may approach 2^128: two corrections
__device__ __forceinline__ u64 reduce128_wide(u64 hi, u64 lo, const DevMod &m) {
  return csub(barrett128(hi, lo, m.q, m.barrett_hi, m.barrett_lo), m.q);
}
__device__ __forceinline__ u64 signed_to_mod(int64_t v, u64 q) {
  return v >= 0 ? (u64)v : q - (u64)(-v);
}

// round(Q m / t) mod q_l, SEAL multiply_add_plain_with_scaling_variant
// floor(x / t) for x = hi:lo < t^2 and an odd t below 2^50, from a double-precision estimate: x and 1/t carry relative
// errors of 2^-52, the quotient is below 2^50, so the estimate is off by less than one before the floor and by at most
// one after it; the remainder, exact in wrapping 64-bit arithmetic, says which.  A dozen full-rate instructions and
// one 64-bit product where the Barrett form (reduce128, then the exact division) takes five 64-bit high products on
// the quarter-rate multiplier: what a transform workgroup that adds a plaintext on load waits for (SRC_CT_PLAIN).
__device__ __forceinline__ u64 floor_div_f64(u64 hi, u64 lo, u64 t) {
  const double xd = (double)hi * 0x1p64 + (double)lo;
  u64 q = (u64)(xd * (1.0 / (double)t));
  const int64_t r = (int64_t)(lo - q * t);
  q -= r < 0 ? 1 : 0;
  q += r >= (int64_t)t ? 1 : 0;
  return q;
}
__device__ __forceinline__ u64 scale_fix(u64 m, const BfvConsts &b, const LevelConsts &c) {
  u64 lo = m * c.q_mod_t, hi = mulhi64(m, c.q_mod_t);
  lo += b.thr;
  hi += lo < b.thr;
  if (b.t_log2 >= 0) return b.t_log2 == 0 ? lo : (lo >> b.t_log2) | (hi << (64 - b.t_log2));
  if ((b.t.q >> 50) == 0) return floor_div_f64(hi, lo, b.t.q);   // m, Q mod t < t: the sum is below t^2
  const u64 r = reduce128(hi, lo, b.t);
  return (lo - r) * b.t_inv64;  // exact division of a multiple of the odd t
}
// (m delta + fix) mod q, canonical.  Under a modulus of the 2^b - c shape and with m, fix < q (a plaintext and the
// correction above are below t) the product takes the word folds; same value either way.
__device__ __forceinline__ u64 scaled_plain(u64 m, u64 fix, u64 delta, const DevMod &q) {
  if (mul_mod_fold_ok(q) && m < q.q && fix < q.q) return add_mod(mul_mod_fold(m, delta, q), fix, q.q);
  u64 lo = m * delta, hi = mulhi64(m, delta);
  lo += fix;
  hi += lo < fix;
  return barrett128(hi, lo, q.q, q.barrett_hi, q.barrett_lo);
}

// RNSTool::decrypt_scale_and_round on x[l] (one coefficient, all limbs of the level)
__device__ __forceinline__ u64 behz_round(const u64 *x, const BfvConsts &b, const LevelConsts &c,
                                          const DevMod *__restrict__ mods) {
  u64 t_lo = 0, t_hi = 0, g_lo = 0, g_hi = 0;
  for (int l = 0; l < c.limbs; ++l) {
    const DevMod &q = mods[l];
    u64 v = mul_mod(x[l], c.prod_t_gamma_mod_q[l], q);
    v = mul_mod(v, c.inv_punct_mod_q[l], q);
    u64 p = v * c.punct_mod_t[l];
    t_lo += p;
    t_hi += mulhi64(v, c.punct_mod_t[l]) + (t_lo < p);
    p = v * c.punct_mod_gamma[l];
    g_lo += p;
    g_hi += mulhi64(v, c.punct_mod_gamma[l]) + (g_lo < p);
  }
  const u64 vt = mulmod_c(reduce128(t_hi, t_lo, b.t), c.neg_inv_q_mod_t, b.t);
  const u64 vg = mulmod_c(reduce128(g_hi, g_lo, b.gamma), c.neg_inv_q_mod_gamma, b.gamma);
  const u64 t = b.t.q, g = b.gamma.q;
  u64 r;
  if (vg > (g >> 1)) {
    r = add_mod(vt, reduce128(0, g - vg, b.t), t);
  } else {
    r = sub_mod(vt, reduce128(0, vg, b.t), t);
  }
  return r ? mulmod_c(r, b.inv_gamma_mod_t, b.t) : 0;
}

}  // namespace mmh
