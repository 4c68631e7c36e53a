// Device side of the batched negacyclic NTT (ntt.hip has the kernels and their launches): butterflies, twiddle
// loads, input sources and the forward transform of one unit as a __device__ function, shared with kernels that run
// transforms beside other work in one launch (conv.hip).
#pragma once
#include "common.hpp"
#include "bfv_device.hpp"
#include "sampler_device.hpp"

namespace mmh {

__device__ __forceinline__ int pad(int j) { return j + (j >> 4); }

// x >= m ? x - m : x
__device__ __forceinline__ u64 csub64(u64 x, u64 m) { return x >= m ? x - m : x; }

// How a kernel keeps its lazy values bounded.  FOLD (q = 2^b - c, all SEAL-generated primes):
// a 3-instruction partial reduction to < 2q every third stage.  Otherwise conditional
// subtractions (5 instructions) every other stage.
template <bool FOLD>
struct Reducer {
  u64 q, q2, q4, q8;
  u32 c, shift, mask;
  __device__ __forceinline__ explicit Reducer(const DevMod &M)
      : q(M.q), q2(2 * M.q), q4(4 * M.q), q8(8 * M.q), c(M.fold_c), shift(M.fold_shift), mask(M.fold_mask) {}
  // forward stage s: X must be small enough that X + 4q stays below 2^64 after the stage
  __device__ __forceinline__ u64 fwd_x(u64 X, int s) const {
    if (FOLD) return (s % 3 == 0 && s > 0) ? fold_mod(X, c, shift, mask) : X;  // < 4q,8q,12q -> 16q -> fold
    return (s & 1) ? csub64(X, q8) : X;                                        // < 12q / < 16q alternating
  }
  // to [0, q) from the bound the last forward stage leaves
  __device__ __forceinline__ u64 fwd_final(u64 v) const {
    if (FOLD) return csub64(fold_mod(v, c, shift, mask), q);
    return csub64(csub64(csub64(csub64(v, q8), q4), q2), q);
  }
  // inverse stage e (e-th executed).  X - Y + bias must stay below 2^64 without wrapping, so inputs
  // stay under 8q: FOLD alternates inputs < 3q (products of the previous stage; sums < 6q are left
  // alone) and inputs < 6q (sums < 12q are folded under 2q).  Otherwise inputs < 4q at every stage.
  __device__ __forceinline__ u64 inv_sum(u64 S, int e) const {
    if (FOLD) return (e & 1) ? fold_mod(S, c, shift, mask) : S;
    return csub64(S, q4);
  }
  __device__ __forceinline__ u64 inv_bias(int e) const {  // multiple of q that is >= any input Y
    if (FOLD) return (e & 1) ? q8 : q4;
    return q4;
  }
};

// The 2^R - 1 twiddles of a radix-2^R group, stage-major: entry (1<<u) - 1 + b is
// table[2^(S0+u) + (jhi<<u) + b].  Loaded ahead of the barrier that precedes their pass so that the
// L2 latency overlaps the previous pass (the kernel is LDS-limited to 4 waves per SIMD, which
// leaves 128 VGPRs per lane).
template <int R, int S0>
__device__ __forceinline__ void load_twiddles(const ulonglong2 *__restrict__ tw, u32 jhi,
                                              ulonglong2 (&w)[(1 << R) - 1]) {
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) w[(1 << u) - 1 + b] = tw[(1u << (S0 + u)) + (jhi << u) + b];
}

// R radix-2 stages (forward, Cooley-Tukey) on 2^R register-resident elements.
// Element k of the group sits at index jhi*(G<<R) + k*G + jlo; stage u pairs k with
// k + 2^(R-1-u) and uses twiddle fwd[2^(S0+u) + (jhi<<u) + (k >> (R-u))].
// A butterfly maps X < B to outputs X + T and X + 4q - T, both < B + 4q (T < 3q); Y is unconstrained.
template <int R, int S0, bool FOLD, bool UNIFORM = false>
__device__ __forceinline__ void fwd_radix(u64 (&x)[1 << R], const ulonglong2 (&tw)[(1 << R) - 1], u64 nq,
                                          const Reducer<FOLD> &rd) {
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) {
      const ulonglong2 w = tw[(1 << u) - 1 + b];
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int k0 = b * 2 * half + i, k1 = k0 + half;
        const u64 X = rd.fwd_x(x[k0], S0 + u);
        const u64 A = UNIFORM ? shoup_mad_uniform(X, x[k1], w.x, w.y, nq)  // X + T
                              : shoup_mad(X, x[k1], w.x, w.y, nq);
        x[k0] = A;
        x[k1] = 2 * X + rd.q4 - A;                        // X + 4q - T
      }
    }
  }
}

// The same stages run backwards (Gentleman-Sande) with the inverse twiddles; E0 = number of
// stages already executed.  Sums double the bound, differences go through the multiplier (< 3q).
template <int R, int S0, int E0, bool FOLD, bool UNIFORM = false>
__device__ __forceinline__ void inv_radix(u64 (&x)[1 << R], const ulonglong2 (&tw)[(1 << R) - 1], u64 nq,
                                          const Reducer<FOLD> &rd) {
#pragma unroll
  for (int u = R - 1; u >= 0; --u) {
    const int half = 1 << (R - 1 - u);
    const int e = E0 + (R - 1 - u);
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) {
      const ulonglong2 w = tw[(1 << u) - 1 + b];
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int k0 = b * 2 * half + i, k1 = k0 + half;
        const u64 X = x[k0], Y = x[k1];
        x[k0] = rd.inv_sum(X + Y, e);
        const u64 D = X + rd.inv_bias(e) - Y;
        x[k1] = UNIFORM ? shoup_mad_uniform(0, D, w.x, w.y, nq) : shoup_mad(0, D, w.x, w.y, nq);
      }
    }
  }
}


// ---------------------------------------------------------------- double-precision butterflies
// Moduli below 2^49 (SEAL's 49-bit q1, every BatchEncoder prime of constants.h:45-50) run their
// butterflies on the FP64 pipe: v_fma_f64 / v_mul_f64 / v_add_f64 / v_rndne_f64 issue at the same
// 4 cycles per wave64 as v_mad_u64_u32 (profiles/r01_f64_probe.txt) and a modular product needs 5 of
// them instead of 12 integer instructions.  Values are signed integers held exactly in doubles:
//   h = fl(w*y), l = w*y - h (exact, FMA), c = rint(h/q), d = h - c*q (exact: an integer below 2^53),
//   w*y = c*q + (d + l) with |d + l| <= (1/2 + 3*B/16) q when |y| <= B*q, since q < 2^49.
// Bound bookkeeping (B = magnitude bound in units of q, every intermediate must stay below 16q <= 2^53):
//   forward  inputs centred to |v| <= 2; a stage maps B to B + 1/2 + 3B/16; the X operands are reduced
//            to 1/2 at the first stage of passes B and C, so B peaks at 8.4 (N = 8192, end of pass A);
//   inverse  inputs centred to |v| <= 1; sums are reduced every other stage, the fixed point of the
//            recurrence is B = 2 at even stages, differences stay <= 8.
// The final value is reduced to [0, q), so the visible result is that of the integer kernels.
// Twiddles of such a modulus are a compact table of doubles at the start of its table half (context.hip).
#pragma clang fp contract(off)
struct F64Mod { double q, qinv; };
__device__ __forceinline__ double f64_mulmod(double w, double y, const F64Mod &m) {
  const double h = w * y;
  const double l = __builtin_fma(w, y, -h);
  const double c = __builtin_rint(h * m.qinv);
  const double d = __builtin_fma(-c, m.q, h);
  return d + l;
}
__device__ __forceinline__ double f64_reduce(double x, const F64Mod &m) {  // |result| <= q/2 (+1)
  return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x);
}
// x < 2^52 as a double, minus `shift`: bias = 2^52 + shift (exact while shift < 2^52)
__device__ __forceinline__ double f64_from_u64(u64 x, double bias) {
  return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - bias;
}
__device__ __forceinline__ u64 f64_canonical(double v, const F64Mod &m) {  // v mod q in [0, q)
  double r = f64_reduce(v, m);
  r += r < 0.0 ? m.q : 0.0;
  return (u64)__double_as_longlong(r + 0x1p52) & 0x000FFFFFFFFFFFFFull;
}
template <int R, int S0>
__device__ __forceinline__ void load_twiddles_f64(const ulonglong2 *__restrict__ tw, u32 jhi,
                                                  double (&w)[(1 << R) - 1]) {
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int b = 0; b < (1 << u); ++b)
      w[(1 << u) - 1 + b] = reinterpret_cast<const double *>(tw)[(1u << (S0 + u)) + (jhi << u) + b];
}
// same index pattern as fwd_radix / inv_radix
template <int R, bool REDUCE_FIRST>
__device__ __forceinline__ void fwd_radix_f64(double (&x)[1 << R], const double (&tw)[(1 << R) - 1],
                                              const F64Mod &m) {
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) {
      const double w = tw[(1 << u) - 1 + b];
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int k0 = b * 2 * half + i, k1 = k0 + half;
        const double X = (REDUCE_FIRST && u == 0) ? f64_reduce(x[k0], m) : x[k0];
        const double T = f64_mulmod(w, x[k1], m);
        x[k0] = X + T;
        x[k1] = X - T;
      }
    }
  }
}
template <int R, int E0>
__device__ __forceinline__ void inv_radix_f64(double (&x)[1 << R], const double (&tw)[(1 << R) - 1],
                                              const F64Mod &m) {
#pragma unroll
  for (int u = R - 1; u >= 0; --u) {
    const int half = 1 << (R - 1 - u);
    const int e = E0 + (R - 1 - u);
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) {
      const double w = tw[(1 << u) - 1 + b];
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int k0 = b * 2 * half + i, k1 = k0 + half;
        const double X = x[k0], Y = x[k1], S = X + Y;
        x[k0] = (e & 1) ? f64_reduce(S, m) : S;
        x[k1] = f64_mulmod(w, X - Y, m);
      }
    }
  }
}

// Where a forward transform takes its input from.
//   SRC_INPLACE  `data` itself.
//   SRC_PLAIN    plaintext `src.p` [n_polys][N] with coefficients mod t, lifted on load into every
//                limb the way multiply_plain does (m >= (t+1)/2 -> m + (q - t)); the lifted
//                polynomials never exist in memory.
//   SRC_FILTER   raw conv filters `src.p` [M][C][fh][fw]: polynomial (m, slice) is encoded on chip
//                (tensor_encoder.cc EncodeFilter: tap (cc, r, col) -> coefficient
//                origin - cc*plane - r*row - col), so neither the sparse plaintexts nor their
//                lifted images exist in memory.
//   SRC_SIGNED   small signed polynomials `src.p` [n_polys][N] (int64: ternary u, noise), mapped to
//                [0, q) per limb on load (sample_poly_ternary / SEAL_NOISE_SAMPLER outputs as
//                encrypt_zero consumes them, hom_conv2d_ss.cc:302-306).
//   SRC_TERNARY  the ternary u of an encryption of zero DRAWN IN PLACE: polynomial k is tile first_tile + k of the
//                ternary stream (key, sid) of sampler_device.hpp -- thread t's 16 samples of a tile are exactly the
//                coefficients 256 j + t it transforms -- so u never exists in memory; the same values as
//                mmh_sample_ternary / mmh_sample_reshare followed by SRC_SIGNED.
//   SRC_CT_PLAIN ciphertexts `src.p` [n][2][limbs][N] at the first level with the plaintexts `src.p2` [n][N] (mod t)
//                added on load the way Evaluator::add_plain does (round(q m / t) onto c0, every limb), so the sum
//                never exists in coefficient form (HomBNSS::bn, hom_bn_ss.cc:263-265, in front of multiply_plain).
enum { SRC_INPLACE = 0, SRC_PLAIN = 1, SRC_FILTER = 2, SRC_SIGNED = 3, SRC_TERNARY = 4, SRC_CT_PLAIN = 5 };
struct NttSource {
  const u64 *p;
  const u64 *p2;                   // inverse transform of a product: the key pair [2][limbs][N] (p = [n][limbs][N])
  // ... or, p2_each != 0, one second operand per output polynomial: p2 = [n][2][limbs][N] (ciphertexts in NTT form
  // times their own plaintext p[k], HomBNSS::bn); p3 / p4 (or null): a second pair of the same shapes whose product is
  // added before the transform (HomBNSS::bole: ex * py + ey * px)
  const u64 *p3, *p4;
  int p2_each;
  u64 t, thr;                      // plain modulus and the lift threshold (t + 1) / 2
  u64 t_bhi, t_blo;                // Barrett constants of t (SRC_FILTER reduces the raw weights)
  int C, taps, wc, n_cslices;      // channels, fh*fw, channels per slice, slices per filter
  int fw, origin, plane, row;      // coefficient = origin - cc*plane - r*row - col
  ChaChaKey key;                   // SRC_TERNARY: the sampler's key, the ternary stream's id and the tile of polynomial 0
  u64 sid, first_tile;
  const BfvConsts *bfv;            // SRC_CT_PLAIN: the context's BFV constants (device memory)
};

// The EA = N/256 stride-256 coefficients of thread t, from wherever the transform reads them.
// (t: the lane's index in its transform workgroup of 256 -- threadIdx.x, except where two transforms share a wider
// workgroup)
template <int LOGN, int SRC>
__device__ __forceinline__ void load_source(u32 unit, u64 (&x)[1 << (LOGN - 8)], const u64 *__restrict__ a,
                                            const DevMod &M, const NttSource &src, int limbs, u64 *lds, int t) {
  constexpr int EA = 1 << (LOGN - 8);
  if (SRC == SRC_CT_PLAIN) {
    const u32 poly = unit / limbs, l = unit % limbs;
    const u64 *__restrict__ c = src.p + ((size_t)unit << LOGN);
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = c[k * 256 + t];
    if ((poly & 1) == 0 && src.p2) {   // c0 of ciphertext poly / 2
      const u64 *__restrict__ m = src.p2 + ((size_t)(poly >> 1) << LOGN);
      const BfvConsts &b = *src.bfv;
      const LevelConsts &lc = b.lv[0];
      u64 mv[EA];   // all loads first: a load behind a branch on the previous one's value is not hoisted over it
#pragma unroll
      for (int k = 0; k < EA; ++k) mv[k] = m[k * 256 + t];
#pragma unroll
      for (int k = 0; k < EA; ++k)
        if (mv[k]) x[k] = add_mod(x[k], scaled_plain(mv[k], scale_fix(mv[k], b, lc), lc.delta_mod_q[l], M), M.q);
    }
  } else if (SRC == SRC_TERNARY) {
#pragma unroll
    for (int h = 0; h < EA / 16; ++h) {
      int64_t v[16];
      ternary_samples(src.key, src.sid, src.first_tile + (size_t)(unit / limbs) * (EA / 16) + h, t, v);
#pragma unroll
      for (int k = 0; k < 16; ++k) x[h * 16 + k] = v[k] >= 0 ? (u64)v[k] : M.q - (u64)(-v[k]);
    }
  } else if (SRC == SRC_PLAIN) {
    const u64 *__restrict__ m = src.p + (size_t)(unit / limbs) * (1 << LOGN);
    const u64 up = M.q - src.t;
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = m[k * 256 + t];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] += x[k] >= src.thr ? up : 0;
  } else if (SRC == SRC_SIGNED) {
    const long long *__restrict__ m =
        reinterpret_cast<const long long *>(src.p) + (size_t)(unit / limbs) * (1 << LOGN);
#pragma unroll
    for (int k = 0; k < EA; ++k) {
      const long long v = m[k * 256 + t];
      x[k] = v >= 0 ? (u64)v : M.q - (u64)(-v);
    }
  } else if (SRC == SRC_FILTER) {
    const int poly = unit / limbs, fm = poly / src.n_cslices, slice = poly % src.n_cslices;
    const int nch = min(src.wc, src.C - slice * src.wc);
    const u64 *__restrict__ taps = src.p + ((size_t)fm * src.C + (size_t)slice * src.wc) * src.taps;
    const u64 up = M.q - src.t;
#pragma unroll
    for (int k = 0; k < EA; ++k) lds[pad(k * 256 + t)] = 0;
    __syncthreads();
    for (int i = t; i < nch * src.taps; i += 256) {
      const int cc = i / src.taps, rc = i % src.taps, r = rc / src.fw, col = rc % src.fw;
      u64 v = barrett128(0, taps[i], src.t, src.t_bhi, src.t_blo);
      v += v >= src.thr ? up : 0;
      lds[pad(src.origin - cc * src.plane - r * src.row - col)] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = lds[pad(k * 256 + t)];
    __syncthreads();  // pass A of the double-precision body writes the same cells with another type
  } else {
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = a[k * 256 + t];
  }
}

template <int LOGN, int SRC>
__device__ __forceinline__ void load_source(u32 unit, u64 (&x)[1 << (LOGN - 8)], const u64 *__restrict__ a,
                                            const DevMod &M, const NttSource &src, int limbs, u64 *lds) {
  load_source<LOGN, SRC>(unit, x, a, M, src, limbs, lds, (int)threadIdx.x);
}

template <int LOGN>
__device__ __forceinline__ void store_from_lds(u64 *__restrict__ a, const u64 *lds) {
  ulonglong2 *__restrict__ out = reinterpret_cast<ulonglong2 *>(a);
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < (1 << LOGN) / 512; ++k) {
    const int j = 2 * (k * 256 + t);
    out[k * 256 + t] = make_ulonglong2(lds[pad(j)], lds[pad(j) + 1]);
  }
}

// Forward transform of one limb with q < 2^49 in double precision; same passes and LDS image.
template <int LOGN, int SRC>
__device__ __forceinline__ void ntt_fwd_f64(u32 unit, u64 *__restrict__ a, const DevMod &M,
                                            const ulonglong2 *__restrict__ tw, const NttSource &src,
                                            int limbs, u64 *lds_u) {
  constexpr int RA = LOGN - 8, EA = 1 << RA, REPS = (1 << LOGN) / 4096;
  double *lds = reinterpret_cast<double *>(lds_u);
  const int t = threadIdx.x;
  const F64Mod m{M.qd, M.qinv};
  double wB[15], wC[15];
  {
    u64 xi[EA];
    load_source<LOGN, SRC>(unit, xi, a, M, src, limbs, lds_u);
    const double bias = 0x1p52 + 2.0 * m.q;  // inputs in [0, 4q) become |v| <= 2q
    double x[EA];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = f64_from_u64(xi[k], bias);
    double wA[EA - 1];
    load_twiddles_f64<RA, 0>(tw, 0, wA);
    load_twiddles_f64<4, RA>(tw, t >> 4, wB);
    fwd_radix_f64<RA, false>(x, wA, m);
#pragma unroll
    for (int k = 0; k < EA; ++k) lds[pad(k * 256 + t)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep, jhi = g >> 4, jlo = g & 15;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    if (rep > 0) load_twiddles_f64<4, RA>(tw, jhi, wB);
    if (rep == REPS - 1) load_twiddles_f64<4, RA + 4>(tw, t, wC);
    fwd_radix_f64<4, true>(x, wB, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[17 * g + k];
    if (rep > 0) load_twiddles_f64<4, RA + 4>(tw, g, wC);
    fwd_radix_f64<4, true>(x, wC, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_u[17 * g + k] = f64_canonical(x[k], m);
  }
  __syncthreads();
  store_from_lds<LOGN>(a, lds_u);
}

// Forward transform of unit `unit` (polynomial unit / limbs, limb unit % limbs) by one 256-thread workgroup; `lds` is
// lds_bytes<LOGN>() of shared memory.  A __device__ function so that other kernels can run transforms beside their
// own work in one launch (conv_front_kernel, conv.hip).
template <int LOGN, bool FOLD, int SRC>
__device__ __forceinline__ void ntt_fwd_unit(u32 unit, u64 *__restrict__ data, const DevMod *__restrict__ mods,
                                             const ulonglong2 *__restrict__ tables, int limbs, int mod_base,
                                             const NttSource &src, u64 *lds) {
  constexpr int N = 1 << LOGN, RA = LOGN - 8, EA = 1 << RA, REPS = N / 4096;
  const int t = threadIdx.x;
  const DevMod &M = mods[mod_base + unit % limbs];
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset;
  u64 *__restrict__ a = data + (size_t)unit * N;
  if (M.f64) {  // uniform over the workgroup
    ntt_fwd_f64<LOGN, SRC>(unit, a, M, tw, src, limbs, lds);
    return;
  }
  const Reducer<FOLD> rd(M);
  const u64 nq = M.nq;

  ulonglong2 wB[15], wC[15];
  {
    u64 x[EA];
    load_source<LOGN, SRC>(unit, x, a, M, src, limbs, lds);
    ulonglong2 wA[EA - 1];
    load_twiddles<RA, 0>(tw, 0, wA);          // wave-uniform: scalar loads
    load_twiddles<4, RA>(tw, t >> 4, wB);     // for pass B (rep 0), in flight during pass A
    fwd_radix<RA, 0, FOLD, true>(x, wA, nq, rd);
#pragma unroll
    for (int k = 0; k < EA; ++k) lds[pad(k * 256 + t)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep, jhi = g >> 4, jlo = g & 15;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    if (rep > 0) load_twiddles<4, RA>(tw, jhi, wB);
    if (rep == REPS - 1) load_twiddles<4, RA + 4>(tw, t, wC);  // for pass C (rep 0)
    fwd_radix<4, RA, FOLD>(x, wB, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[17 * g + k];
    if (rep > 0) load_twiddles<4, RA + 4>(tw, g, wC);
    fwd_radix<4, RA + 4, FOLD>(x, wC, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[17 * g + k] = rd.fwd_final(x[k]);
  }
  __syncthreads();
  store_from_lds<LOGN>(a, lds);
}

// N = 4096, one integer limb of the 2^b - c shape: component j of the pk-encryption of zero of the re-sharing
// (asymmetric_encrypt_zero, hom_conv2d_ss.cc:302-316: u drawn, transformed, multiplied by pk_j, transformed back) in
// ONE workgroup, nothing but the result written.  The forward transform's last pass and the inverse transform's first
// work on the same 16 consecutive coefficients of a lane, so the product with the key happens in registers between
// them.  Same canonical values as SRC_TERNARY -> ntt_inv_kernel<PROD>: out = INTT(NTT(u) (.) key).
// `t` is the lane's index among the unit's 256 (every __syncthreads() below is reached by all lanes of the workgroup: a
// 512-thread workgroup runs two units side by side, each on its own `lds`).
template <bool FOLD>
__device__ __forceinline__ void ntt_zero_unit12(u32 poly, const DevMod &M, const ulonglong2 *__restrict__ tables,
                                                const NttSource &src, const u64 *__restrict__ key,
                                                u64 *__restrict__ out, u64 *lds, int t) {
  constexpr int LOGN = 12, N = 1 << LOGN;
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset, *__restrict__ twi = tw + N;
  const Reducer<FOLD> rd(M);
  const u64 nq = M.nq;
  {
    ulonglong2 wB[15];
    {
      u64 x[16];
      load_source<LOGN, SRC_TERNARY>(poly, x, nullptr, M, src, 1, lds, t);
      ulonglong2 wA[15];
      load_twiddles<4, 0>(tw, 0, wA);
      load_twiddles<4, 4>(tw, t >> 4, wB);
      fwd_radix<4, 0, FOLD, true>(x, wA, nq, rd);
#pragma unroll
      for (int k = 0; k < 16; ++k) lds[pad(k * 256 + t)] = x[k];
    }
    __syncthreads();
    {
      const int jhi = t >> 4, jlo = t & 15;
      u64 x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
      fwd_radix<4, 4, FOLD>(x, wB, nq, rd);
#pragma unroll
      for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
    }
  }
  __syncthreads();
  ulonglong2 iB[15];
  {
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[17 * t + k];
    {
      ulonglong2 wC[15];
      load_twiddles<4, 8>(tw, t, wC);
      fwd_radix<4, 8, FOLD>(x, wC, nq, rd);
    }
    const ulonglong2 *__restrict__ kv = reinterpret_cast<const ulonglong2 *>(key) + 8 * t;
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // coefficients 16 t + 2k, + 2k + 1 of NTT(u), times the key's
      const ulonglong2 w = kv[k];
      x[2 * k] = mul_mod_fold(rd.fwd_final(x[2 * k]), w.x, M);
      x[2 * k + 1] = mul_mod_fold(rd.fwd_final(x[2 * k + 1]), w.y, M);
    }
    ulonglong2 iC[15];
    load_twiddles<4, 8>(twi, t, iC);
    load_twiddles<4, 4>(twi, t >> 4, iB);
    inv_radix<4, 8, 0, FOLD>(x, iC, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[17 * t + k] = x[k];   // (the cells this lane just read: no barrier before)
  }
  __syncthreads();
  {
    const int jhi = t >> 4, jlo = t & 15;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    inv_radix<4, 4, 4, FOLD>(x, iB, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
  {
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(k * 256 + t)];
    ulonglong2 wA[15];
    load_twiddles<4, 0>(twi, 0, wA);
    inv_radix<4, 0, 8, FOLD, true>(x, wA, nq, rd);
    const u64 ninv = M.ninv, ninv_sh = M.ninv_sh;
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k * 256 + t] = csub64(csub64(mul_shoup_lazy4(x[k], ninv, ninv_sh, nq), rd.q2), rd.q);
  }
}

template <bool FOLD>
__device__ __forceinline__ void ntt_zero_unit12(u32 poly, const DevMod &M, const ulonglong2 *__restrict__ tables,
                                                const NttSource &src, const u64 *__restrict__ key,
                                                u64 *__restrict__ out, u64 *lds) {
  ntt_zero_unit12<FOLD>(poly, M, tables, src, key, out, lds, (int)threadIdx.x);
}

template <int LOGN>
static constexpr size_t lds_bytes() {
  return ((size_t(1) << LOGN) + (size_t(1) << (LOGN - 4))) * sizeof(u64);
}
}  // namespace mmh
