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
__device__ __forceinline__ u64 scale_fix(u64 m, const BfvConsts &b, const LevelConsts &c) {
  u64 lo = m * c.q_mod_t, hi = mulhi64(m, c.q_mod_t);
  lo += b.thr;
  hi += lo < b.thr;
  if (b.t_log2 >= 0) return b.t_log2 == 0 ? lo : (lo >> b.t_log2) | (hi << (64 - b.t_log2));
  const u64 r = reduce128(hi, lo, b.t);
  return (lo - r) * b.t_inv64;  // exact division of a multiple of the odd t
}
__device__ __forceinline__ u64 scaled_plain(u64 m, u64 fix, u64 delta, const DevMod &q) {
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
