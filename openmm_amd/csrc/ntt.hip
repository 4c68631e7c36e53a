// Batched negacyclic NTT / inverse NTT for N = 4096 and 8192 on gfx950.
//
// One 256-thread workgroup transforms one polynomial limb entirely on chip:
//   pass A  radix-2^(logN-8) butterflies on registers straight from coalesced HBM loads,
//   pass B  radix-16 on 16-element groups of stride 16 exchanged through LDS,
//   pass C  radix-16 on 16 consecutive elements, then a coalesced 16-B/lane store.
// 64-bit Harvey butterflies with Shoup twiddles (w, floor(w 2^64/q)) on VALU; values stay
// lazily reduced (below 16q forward, 4q inverse) and are corrected once at the end,
// so the visible result equals seal::util::ntt_negacyclic_harvey /
// inverse_ntt_negacyclic_harvey (hom_conv2d_ss.cc:306,315,333): bit-reversed order,
// fully reduced.
//
// LDS image: element j lives at j + (j >> 4) (one 8-byte pad per 16 elements), which makes
// every access pattern of the three passes bank-conflict free for ds_read_b64/ds_write_b64.
#include "common.hpp"
#include "bfv_device.hpp"
#include "host_math.hpp"

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
enum { SRC_INPLACE = 0, SRC_PLAIN = 1, SRC_FILTER = 2, SRC_SIGNED = 3 };
struct NttSource {
  const u64 *p;
  const u64 *p2;                   // inverse transform of a product: the key pair [2][limbs][N] (p = [n][limbs][N])
  u64 t, thr;                      // plain modulus and the lift threshold (t + 1) / 2
  u64 t_bhi, t_blo;                // Barrett constants of t (SRC_FILTER reduces the raw weights)
  int C, taps, wc, n_cslices;      // channels, fh*fw, channels per slice, slices per filter
  int fw, origin, plane, row;      // coefficient = origin - cc*plane - r*row - col
};

// The EA = N/256 stride-256 coefficients of thread t, from wherever the transform reads them.
template <int LOGN, int SRC>
__device__ __forceinline__ void load_source(u64 (&x)[1 << (LOGN - 8)], const u64 *__restrict__ a,
                                            const DevMod &M, const NttSource &src, int limbs, u64 *lds) {
  constexpr int EA = 1 << (LOGN - 8);
  const int t = threadIdx.x;
  if (SRC == SRC_PLAIN) {
    const u64 *__restrict__ m = src.p + (size_t)(blockIdx.x / limbs) * (1 << LOGN);
    const u64 up = M.q - src.t;
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = m[k * 256 + t];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] += x[k] >= src.thr ? up : 0;
  } else if (SRC == SRC_SIGNED) {
    const long long *__restrict__ m =
        reinterpret_cast<const long long *>(src.p) + (size_t)(blockIdx.x / limbs) * (1 << LOGN);
#pragma unroll
    for (int k = 0; k < EA; ++k) {
      const long long v = m[k * 256 + t];
      x[k] = v >= 0 ? (u64)v : M.q - (u64)(-v);
    }
  } else if (SRC == SRC_FILTER) {
    const int poly = blockIdx.x / limbs, fm = poly / src.n_cslices, slice = poly % src.n_cslices;
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
__device__ __forceinline__ void ntt_fwd_f64(u64 *__restrict__ a, const DevMod &M,
                                            const ulonglong2 *__restrict__ tw, const NttSource &src,
                                            int limbs, u64 *lds_u) {
  constexpr int RA = LOGN - 8, EA = 1 << RA, REPS = (1 << LOGN) / 4096;
  double *lds = reinterpret_cast<double *>(lds_u);
  const int t = threadIdx.x;
  const F64Mod m{M.qd, M.qinv};
  double wB[15], wC[15];
  {
    u64 xi[EA];
    load_source<LOGN, SRC>(xi, a, M, src, limbs, lds_u);
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

template <int LOGN, bool FOLD, int SRC = SRC_INPLACE>
__global__ __launch_bounds__(256) void ntt_fwd_kernel(u64 *__restrict__ data,
                                                      const DevMod *__restrict__ mods,
                                                      const ulonglong2 *__restrict__ tables,
                                                      int limbs, int mod_base,
                                                      NttSource src) {
  constexpr int N = 1 << LOGN, RA = LOGN - 8, EA = 1 << RA, REPS = N / 4096;
  extern __shared__ __align__(16) u64 lds[];
  const int t = threadIdx.x;
  const DevMod &M = mods[mod_base + blockIdx.x % limbs];
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset;
  u64 *__restrict__ a = data + (size_t)blockIdx.x * N;
  if (M.f64) {  // uniform over the workgroup
    ntt_fwd_f64<LOGN, SRC>(a, M, tw, src, limbs, lds);
    return;
  }
  const Reducer<FOLD> rd(M);
  const u64 nq = M.nq;

  ulonglong2 wB[15], wC[15];
  {
    u64 x[EA];
    load_source<LOGN, SRC>(x, a, M, src, limbs, lds);
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

// What becomes of the coefficients an inverse transform produces.
//   SINK_STORE    written in place.
//   SINK_RESHARE  unit (p, j) is component j of the pk-encryption of zero of output ciphertext p at the last level
//                 (PROD source); its coefficients go straight into the re-sharing tail of the approximate build
//                 (hom_conv2d_ss.cc:761-788, :392-441, :108-140):
//                   out = truncate(mod_switch(acc) + zero + (0, e1) - (mask, 0)), unused coefficients of c0 cleared,
//                 and the other party's share is the mask's scale-and-round at the target coefficients.  The
//                 encryption of zero never exists in memory.  acc [n][2][2][N] coefficient form at the first level,
//                 mask [n][N] mod q0, e1 signed, out [n][2][N].
enum { SINK_STORE = 0, SINK_RESHARE = 1 };
struct ReshareSink {
  const u64 *acc;
  const int64_t *e1;
  const u64 *mask;
  u64 *out;
  u64 inv_qlast, mask0, mask1;
  const uint8_t *keep;
  size_t keep_period;
  const int *share_dst;
  size_t share_plane;
  u64 *share_out;
  BfvConsts b;
  // constants as (w, floor(w 2^64 / q)) pairs for the fixed multipliers of the tail: q1^-1 mod q0, and for the share
  // of a power-of-two ring t*gamma mod q0 and -q0^-1 mod gamma (fast = 1; else the general behz_round)
  u64 inv_qlast_sh, sh_c, sh_c_sh, sh_ng, sh_ng_sh;
  int fast, qlast_small;
};
// The tail's arithmetic on one coefficient (shared by the fused kernel's Emitter and the element-wise kernel):
// divide_and_round_q_last of the accumulator, + the encryption of zero, then e1 (c1) or - mask (c0), truncation.
struct ReshareCoeff {
  const ReshareSink &f;
  const DevMod *__restrict__ mods;
  const DevMod &q0;
  u64 q1, half, half0;
  __device__ __forceinline__ ReshareCoeff(const ReshareSink &f_, const DevMod *mods_)
      : f(f_), mods(mods_), q0(mods_[0]), q1(mods_[1].q), half(mods_[1].q >> 1), half0(reduce64(mods_[1].q >> 1, mods_[0])) {}
  // x * w mod q in [0, q) for a fixed w with its quotient wsh: the same canonical value as Barrett's mul_mod
  __device__ __forceinline__ static u64 mul_fixed(u64 x, u64 w, u64 wsh, u64 q) {
    const u64 v = shoup_mad_uniform(0, x, w, wsh, 0 - q);   // [0, 3q)
    return csub(csub(v, 2 * q), q);
  }
  __device__ __forceinline__ u64 switched(u64 a0, u64 a1, u64 zz) const {
    const u64 last = csub(a1 + half, q1);
    const u64 lastq = f.qlast_small ? last : reduce64(last, q0);   // q1 < q0: already reduced
    const u64 tmp = sub_mod(lastq, half0, q0.q);
    return add_mod(mul_fixed(sub_mod(a0, tmp, q0.q), f.inv_qlast, f.inv_qlast_sh, q0.q), zz, q0.q);
  }
  __device__ __forceinline__ u64 c1(u64 a0, u64 a1, u64 zz, int64_t e) const {
    return add_mod(switched(a0, a1, zz), signed_to_mod(e, q0.q), q0.q) & f.mask1;
  }
  __device__ __forceinline__ u64 c0(u64 a0, u64 a1, u64 zz, u64 m, bool kept) const {
    const u64 r = sub_mod(switched(a0, a1, zz), m, q0.q) & f.mask0;
    return kept ? r : 0;
  }
  // the other party's share: scale-and-round of the mask at a target coefficient (RNSTool::decrypt_scale_and_round
  // on one limb).  With t a power of two and gamma above q0 the general routine's nine Barrett reductions come down
  // to two fixed multiplications and masks: at the last level Q = q0, so v = m t gamma mod q0 enters both bases as it
  // is, the gamma side is v (-q0^-1) mod gamma, and everything modulo t is a mask (2^64 is a multiple of t).
  __device__ __forceinline__ u64 share(u64 m) const {
    if (!f.fast) {
      const u64 x[2] = {m, 0};
      return behz_round(x, f.b, f.b.lv[1], mods);
    }
    const u64 tm = f.b.t.q - 1, g = f.b.gamma.q;
    const u64 v = mul_fixed(m, f.sh_c, f.sh_c_sh, q0.q);
    const u64 vt = (v * f.b.lv[1].neg_inv_q_mod_t) & tm;
    const u64 vg = mul_fixed(v, f.sh_ng, f.sh_ng_sh, g);
    const u64 r = (vg > (g >> 1) ? vt + (g - vg) : vt - vg) & tm;
    return (r * f.b.inv_gamma_mod_t) & tm;
  }
};
template <int LOGN, int SINK>
struct Emitter {
  static constexpr int EA = 1 << (LOGN - 8), H = 4;  // coefficients per lane, and per batch of loads
  u64 *__restrict__ a;    // SINK_STORE: the polynomial itself
  const ReshareSink &f;
  const DevMod *__restrict__ mods;
  size_t p;               // output ciphertext, component
  int j;
  __device__ __forceinline__ Emitter(u64 *a_, const ReshareSink &f_, const DevMod *mods_, u32 unit)
      : a(a_), f(f_), mods(mods_), p(unit >> 1), j(unit & 1) {}
  // coefficients k*256 + t of the unit, canonical modulo this limb's q.  The tail parks them in the lane's own
  // cells of the LDS image (nobody else reads those in the last pass) and walks them in batches of H: all loads
  // of a batch ahead of its arithmetic and its stores, one memory round trip per batch instead of one per
  // coefficient, and the batch loop stays a loop.
  __device__ __forceinline__ void operator()(int t, const u64 (&zr)[EA], u64 *lds) const {
    if (SINK == SINK_STORE) {
#pragma unroll
      for (int k = 0; k < EA; ++k) a[k * 256 + t] = zr[k];
      return;
    }
#pragma unroll
    for (int k = 0; k < EA; ++k) lds[pad(k * 256 + t)] = zr[k];
    const u64 *z = lds + pad(t);  // coefficient k*256 + t at z[17 * 16 * k]
    constexpr int ZS = 256 + 16;
    const ReshareCoeff rc(f, mods);
    const u64 *__restrict__ acc0 = f.acc + (((p * 2 + j) * 2) << LOGN), *__restrict__ acc1 = acc0 + (size_t(1) << LOGN);
    u64 *__restrict__ dst = f.out + ((p * 2 + j) << LOGN);
    if (j == 1) {
      const int64_t *__restrict__ e1 = f.e1 ? f.e1 + (p << LOGN) : nullptr;
#pragma unroll 1
      for (int h = 0; h < EA; h += H) {
        u64 a0[H], a1[H];
        int64_t e[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
          const int i = (h + k) * 256 + t;
          a0[k] = acc0[i], a1[k] = acc1[i], e[k] = e1 ? e1[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < H; ++k) dst[(h + k) * 256 + t] = rc.c1(a0[k], a1[k], z[(h + k) * ZS], e[k]);
      }
      return;
    }
    const size_t kp0 = (p % f.keep_period) << LOGN;
    const u64 *__restrict__ mask = f.mask + (p << LOGN);
    u64 *__restrict__ share = f.share_out + (p / f.keep_period) * f.share_plane;
#pragma unroll 1
    for (int h = 0; h < EA; h += H) {
      bool kept[H];
      int d[H];
      u64 m[H], a0[H], a1[H];
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const int i = (h + k) * 256 + t;
        kept[k] = !f.keep || f.keep[kp0 + i] != 0;
        d[k] = f.share_dst ? f.share_dst[kp0 + i] : -1;
        m[k] = mask[i];
      }
#pragma unroll
      for (int k = 0; k < H; ++k) {  // of removed c0 coefficients nothing more is read
        const int i = (h + k) * 256 + t;
        a0[k] = kept[k] ? acc0[i] : 0, a1[k] = kept[k] ? acc1[i] : 0;
      }
#pragma unroll
      for (int k = 0; k < H; ++k) {
        if (d[k] >= 0) share[d[k]] = rc.share(m[k]);
        dst[(h + k) * 256 + t] = rc.c0(a0[k], a1[k], z[(h + k) * ZS], m[k], kept[k]);
      }
    }
  }
};

// Inverse transform of one limb with q < 2^49 in double precision (LDS already holds the input).
template <int LOGN, class Emit>
__device__ __forceinline__ void ntt_inv_f64(const Emit &emit, const DevMod &M,
                                            const ulonglong2 *__restrict__ tw, u64 *lds_u) {
  constexpr int RA = LOGN - 8, EA = 1 << RA, REPS = (1 << LOGN) / 4096;
  double *lds = reinterpret_cast<double *>(lds_u);
  const int t = threadIdx.x;
  const F64Mod m{M.qd, M.qinv};
  const double bias = 0x1p52 + m.q;  // inputs in [0, 2q) become |v| <= q
  double wB[15], wC[15];
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = f64_from_u64(lds_u[17 * g + k], bias);
    load_twiddles_f64<4, RA + 4>(tw, g, wC);
    if (rep == REPS - 1) load_twiddles_f64<4, RA>(tw, t >> 4, wB);
    inv_radix_f64<4, 0>(x, wC, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[17 * g + k] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep, jhi = g >> 4, jlo = g & 15;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    if (rep > 0) load_twiddles_f64<4, RA>(tw, jhi, wB);
    inv_radix_f64<4, 4>(x, wB, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
  {
    double x[EA];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = lds[pad(k * 256 + t)];
    double wA[EA - 1];
    load_twiddles_f64<RA, 0>(tw, 0, wA);
    inv_radix_f64<RA, 8>(x, wA, m);
    const double ninv = (double)M.ninv;  // exact: below q < 2^49
    u64 z[EA];
#pragma unroll
    for (int k = 0; k < EA; ++k) z[k] = f64_canonical(f64_mulmod(ninv, x[k], m), m);
    emit(t, z, lds_u);
  }
}

// PROD: polynomial (k, j, l) of the output is the inverse transform of src.p[k][l] (.) src.p2[j][l], the dyadic
// product formed while loading (encrypt_zero's u * pk_j without a pass of its own); else in place.
// (second launch bound: the waves per SIMD that the LDS image allows anyway, so that the tail's batched loads are
// fitted into 128 registers instead of costing a workgroup per CU)
template <int LOGN, bool FOLD, bool PROD = false, int SINK = SINK_STORE>
__global__ __launch_bounds__(256, LOGN == 12 ? 4 : 2) void ntt_inv_kernel(u64 *__restrict__ data,
                                                      const DevMod *__restrict__ mods,
                                                      const ulonglong2 *__restrict__ tables,
                                                      int limbs, int mod_base, NttSource src, ReshareSink fin) {
  constexpr int N = 1 << LOGN, RA = LOGN - 8, EA = 1 << RA, REPS = N / 4096;
  extern __shared__ __align__(16) u64 lds[];
  const int t = threadIdx.x;
  const DevMod &M = mods[mod_base + blockIdx.x % limbs];
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset + N;
  u64 *__restrict__ a = data + (size_t)blockIdx.x * N;

  if (PROD) {
    const size_t kj = blockIdx.x / limbs, l = blockIdx.x % limbs;
    const ulonglong2 *__restrict__ in =
        reinterpret_cast<const ulonglong2 *>(src.p + (((kj >> 1) * limbs + l) << LOGN));
    const ulonglong2 *__restrict__ key =
        reinterpret_cast<const ulonglong2 *>(src.p2 + (((kj & 1) * limbs + l) << LOGN));
#pragma unroll
    for (int k = 0; k < N / 512; ++k) {
      const int j = 2 * (k * 256 + t);
      const ulonglong2 v = in[k * 256 + t], w = key[k * 256 + t];
      lds[pad(j)] = mul_mod(v.x, w.x, M);
      lds[pad(j) + 1] = mul_mod(v.y, w.y, M);
    }
  } else {
    const ulonglong2 *__restrict__ in = reinterpret_cast<const ulonglong2 *>(a);
#pragma unroll
    for (int k = 0; k < N / 512; ++k) {
      const int j = 2 * (k * 256 + t);
      const ulonglong2 v = in[k * 256 + t];
      lds[pad(j)] = v.x;
      lds[pad(j) + 1] = v.y;
    }
  }
  __syncthreads();
  const Emitter<LOGN, SINK> emit(a, fin, mods, blockIdx.x);  // (SINK_RESHARE runs with limbs == 1)
  if (M.f64) {  // uniform over the workgroup
    ntt_inv_f64<LOGN>(emit, M, tw, lds);
    return;
  }
  const Reducer<FOLD> rd(M);
  const u64 nq = M.nq;
  ulonglong2 wB[15], wC[15];
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[17 * g + k];
    load_twiddles<4, RA + 4>(tw, g, wC);
    if (rep == REPS - 1) load_twiddles<4, RA>(tw, t >> 4, wB);  // for the next pass (rep 0)
    inv_radix<4, RA + 4, 0, FOLD>(x, wC, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[17 * g + k] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep, jhi = g >> 4, jlo = g & 15;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    if (rep > 0) load_twiddles<4, RA>(tw, jhi, wB);
    inv_radix<4, RA, 4, FOLD>(x, wB, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
  {
    u64 x[EA];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = lds[pad(k * 256 + t)];
    ulonglong2 wA[EA - 1];
    load_twiddles<RA, 0>(tw, 0, wA);
    inv_radix<RA, 0, 8, FOLD, true>(x, wA, nq, rd);
    const u64 ninv = M.ninv, ninv_sh = M.ninv_sh;
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = csub64(csub64(mul_shoup_lazy4(x[k], ninv, ninv_sh, nq), rd.q2), rd.q);
    emit(t, x, lds);
  }
}

// The tail alone, one thread per coefficient, for an encryption of zero that exists in memory: prepared ahead of the
// inputs (mmh_reshare_zero_prepare), or written by a separate inverse transform when the call has so few output
// ciphertexts that the fused kernel's one workgroup per component would leave most of the chip idle and pay its
// batches' memory round trips one after the other.  `zero` [n][2][N] canonical mod q0; same ReshareCoeff arithmetic as
// the fused kernel's Emitter, so the same bits.
__global__ __launch_bounds__(256) void reshare_elem_kernel(const u64 *__restrict__ zero, const DevMod *__restrict__ mods,
                                                           ReshareSink f, int logN, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t unit = idx >> logN, p = unit >> 1;
  const int i = (int)(idx & ((size_t(1) << logN) - 1)), j = (int)(unit & 1);
  const ReshareCoeff rc(f, mods);
  const size_t a = (((p * 2 + j) * 2) << logN) + i;
  const u64 zz = zero[idx];
  if (j == 1) {
    const int64_t e = f.e1 ? f.e1[(p << logN) + i] : 0;
    f.out[idx] = rc.c1(f.acc[a], f.acc[a + (size_t(1) << logN)], zz, e);
    return;
  }
  const size_t kp = ((p % f.keep_period) << logN) + i;
  const bool kept = !f.keep || f.keep[kp] != 0;
  const int d = f.share_dst ? f.share_dst[kp] : -1;
  const u64 m = f.mask[(p << logN) + i];
  if (d >= 0) f.share_out[(p / f.keep_period) * f.share_plane + d] = rc.share(m);
  const u64 a0 = kept ? f.acc[a] : 0, a1 = kept ? f.acc[a + (size_t(1) << logN)] : 0;
  f.out[idx] = rc.c0(a0, a1, zz, m, kept);
}

template <int LOGN>
static constexpr size_t lds_bytes() {
  return ((size_t(1) << LOGN) + (size_t(1) << (LOGN - 4))) * sizeof(u64);
}

template <int LOGN, class K>
static int launch_inverse(K kern, u64 *d, const mmh_ctx *ctx, size_t blocks, int limbs, int mod_base,
                          const NttSource &src, const ReshareSink &fin, hipStream_t s) {
  if (lds_bytes<LOGN>() > 48 * 1024)  // per device and cheap: set on every launch
    MMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<LOGN>()));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes<LOGN>(), s, d, ctx->d_mods,
                     ctx->d_tables, limbs, mod_base, src, fin);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

// mode: SRC_* for a forward transform, -1 for the inverse (in place), -2 for the inverse of a product, -3 for the
// inverse of a product whose coefficients feed the re-sharing tail (fin)
template <int LOGN>
static int launch(int mode, bool fold, u64 *d, const mmh_ctx *ctx, size_t blocks, int limbs, int mod_base,
                  const NttSource &src, hipStream_t s, const ReshareSink *fin = nullptr) {
  if (mode == -3)
    return fold ? launch_inverse<LOGN>(ntt_inv_kernel<LOGN, true, true, SINK_RESHARE>, d, ctx, blocks, limbs, mod_base,
                                       src, *fin, s)
                : launch_inverse<LOGN>(ntt_inv_kernel<LOGN, false, true, SINK_RESHARE>, d, ctx, blocks, limbs,
                                       mod_base, src, *fin, s);
  if (mode == -2)
    return fold ? launch_inverse<LOGN>(ntt_inv_kernel<LOGN, true, true>, d, ctx, blocks, limbs, mod_base, src,
                                       ReshareSink{}, s)
                : launch_inverse<LOGN>(ntt_inv_kernel<LOGN, false, true>, d, ctx, blocks, limbs, mod_base, src,
                                       ReshareSink{}, s);
  if (mode == -1)
    return fold ? launch_inverse<LOGN>(ntt_inv_kernel<LOGN, true>, d, ctx, blocks, limbs, mod_base, src,
                                       ReshareSink{}, s)
                : launch_inverse<LOGN>(ntt_inv_kernel<LOGN, false>, d, ctx, blocks, limbs, mod_base, src,
                                       ReshareSink{}, s);
  auto kern = 
                mode == SRC_PLAIN
                  ? (fold ? ntt_fwd_kernel<LOGN, true, SRC_PLAIN> : ntt_fwd_kernel<LOGN, false, SRC_PLAIN>)
              : mode == SRC_FILTER
                  ? (fold ? ntt_fwd_kernel<LOGN, true, SRC_FILTER> : ntt_fwd_kernel<LOGN, false, SRC_FILTER>)
              : mode == SRC_SIGNED
                  ? (fold ? ntt_fwd_kernel<LOGN, true, SRC_SIGNED> : ntt_fwd_kernel<LOGN, false, SRC_SIGNED>)
                  : (fold ? ntt_fwd_kernel<LOGN, true, SRC_INPLACE> : ntt_fwd_kernel<LOGN, false, SRC_INPLACE>);
  if (lds_bytes<LOGN>() > 48 * 1024)  // per device and cheap: set on every launch
    MMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<LOGN>()));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes<LOGN>(), s, d, ctx->d_mods,
                     ctx->d_tables, limbs, mod_base, src);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

static int dispatch(const mmh_ctx *ctx, int mode, u64 *d, const NttSource &src, size_t n_polys, int limbs,
                    int mod_base, hipStream_t s, const ReshareSink *fin = nullptr) {
  const size_t blocks = n_polys * (size_t)limbs;
  if (blocks == 0) return MMH_OK;
  MMH_REQUIRE(blocks <= 0x7fffffffu, MMH_ERR_OUT_BOUND);
  bool fold = true;  // every modulus of this launch must have the 2^b - c shape for the fast reducer
  for (int l = 0; l < limbs; ++l) fold = fold && ctx->h_mods[mod_base + l].fold_c != 0;
  switch (ctx->logN) {
    case 12: return launch<12>(mode, fold, d, ctx, blocks, limbs, mod_base, src, s, fin);
    case 13: return launch<13>(mode, fold, d, ctx, blocks, limbs, mod_base, src, s, fin);
    default: MMH_FAIL(MMH_ERR_CONFIG, "only N = 4096 and N = 8192 are built");
  }
}

// out[k][l] = NTT_l(lift_l(src[k])) for the first `limbs` moduli, src [n_polys][N] mod t
int ntt_lift_dispatch(const mmh_ctx *ctx, const u64 *d_src, u64 *d_out, size_t n_polys, int limbs,
                      hipStream_t s) {
  NttSource src{};
  src.p = d_src, src.t = ctx->t, src.thr = ctx->bfv.thr;
  return dispatch(ctx, SRC_PLAIN, d_out, src, n_polys, limbs, 0, s);
}

// out[k][l] = NTT_l(src[k] mod q_l) for the first `limbs` moduli, src [n_polys][N] signed
int ntt_signed_dispatch(const mmh_ctx *ctx, const int64_t *d_src, u64 *d_out, size_t n_polys, int limbs,
                        hipStream_t s) {
  NttSource src{};
  src.p = reinterpret_cast<const u64 *>(d_src);
  return dispatch(ctx, SRC_SIGNED, d_out, src, n_polys, limbs, 0, s);
}

// out[m*n_cslices + slice][l] = NTT_l(lift_l(EncodeFilter(filters[m], slice))), filters [M][C][fh][fw]
int ntt_filter_dispatch(const mmh_ctx *ctx, const u64 *d_filters, u64 *d_out, size_t n_filters, int C, int fh,
                        int fw, int wc, int n_cslices, long origin, long plane, long row, int limbs,
                        hipStream_t s) {
  // every tap must land inside the polynomial: largest index is origin, smallest origin - span
  const long span = (long)(wc - 1) * plane + (long)(fh - 1) * row + (fw - 1);
  MMH_REQUIRE(origin < (long)ctx->N && origin - span >= 0 && wc >= 1 && n_cslices >= 1, MMH_ERR_OUT_BOUND);
  MMH_REQUIRE((long)(n_cslices - 1) * wc < C && (long)n_cslices * wc >= C, MMH_ERR_OUT_BOUND);
  NttSource src{};
  src.p = d_filters, src.t = ctx->t, src.thr = ctx->bfv.thr;
  src.t_bhi = ctx->bfv.t.barrett_hi, src.t_blo = ctx->bfv.t.barrett_lo;
  src.C = C, src.taps = fh * fw, src.wc = wc, src.n_cslices = n_cslices;
  src.fw = fw, src.origin = (int)origin, src.plane = (int)plane, src.row = (int)row;
  return dispatch(ctx, SRC_FILTER, d_out, src, n_filters * (size_t)n_cslices, limbs, 0, s);
}

// out[k][j][l] = INTT_l(u_ntt[k][l] (.) key[j][l]) for j = 0, 1 and the first `limbs` moduli
int ntt_inv_product_dispatch(const mmh_ctx *ctx, const u64 *d_u_ntt, const u64 *d_key, u64 *d_out, size_t n,
                             int limbs, hipStream_t s) {
  NttSource src{};
  src.p = d_u_ntt, src.p2 = d_key;
  return dispatch(ctx, -2, d_out, src, n * 2, limbs, 0, s);
}

static ReshareSink make_sink(const mmh_ctx *ctx, const u64 *d_acc, const int64_t *d_e1, const u64 *d_mask, u64 *d_out,
                             const uint8_t *d_keep, size_t keep_period, const int *d_share_dst, size_t share_plane,
                             u64 *d_share_out) {
  const BfvConsts &b = ctx->bfv;
  const u64 q0 = ctx->q[0], g = b.gamma.q;
  ReshareSink f{d_acc, d_e1, d_mask, d_out, b.inv_qlast_mod_q0, b.mask0, b.mask1, d_keep, keep_period ? keep_period : 1,
                d_share_dst, share_plane, d_share_out, b, 0, 0, 0, 0, 0, 0, 0};
  f.inv_qlast_sh = shoup(b.inv_qlast_mod_q0, q0);
  f.qlast_small = ctx->L == 2 && ctx->q[1] < q0;
  // last level: one limb, Q = q0, so inv_punct = punct = 1
  f.fast = b.t_log2 >= 0 && g > q0 && (g >> 62) == 0 && b.lv[1].inv_punct_mod_q[0] == 1 && b.lv[1].punct_mod_gamma[0] == 1 &&
           (b.t.q == 1 || b.lv[1].punct_mod_t[0] == 1);
  f.sh_c = b.lv[1].prod_t_gamma_mod_q[0], f.sh_c_sh = shoup(f.sh_c, q0);
  f.sh_ng = b.lv[1].neg_inv_q_mod_gamma, f.sh_ng_sh = g ? shoup(f.sh_ng, g) : 0;
  return f;
}
// The re-sharing tail of n output ciphertexts from u in NTT form: components (p, j) = INTT(u_ntt[p] (.) key[j])
// at the last level go through ReshareSink into out / share_out; nothing else is written.
int ntt_inv_product_reshare(const mmh_ctx *ctx, const u64 *d_u_ntt, const u64 *d_key, size_t n, const u64 *d_acc,
                            const int64_t *d_e1, const u64 *d_mask, u64 *d_out, const uint8_t *d_keep,
                            size_t keep_period, const int *d_share_dst, size_t share_plane, u64 *d_share_out,
                            hipStream_t s) {
  NttSource src{};
  src.p = d_u_ntt, src.p2 = d_key;
  const ReshareSink fin = make_sink(ctx, d_acc, d_e1, d_mask, d_out, d_keep, keep_period, d_share_dst, share_plane, d_share_out);
  return dispatch(ctx, -3, d_out, src, n * 2, 1, 0, s, &fin);
}

// The same tail from an encryption of zero in memory, d_zero [n][2][N] (coefficient form at the last level).
int reshare_online_dispatch(const mmh_ctx *ctx, const u64 *d_zero, size_t n, const u64 *d_acc, const int64_t *d_e1,
                            const u64 *d_mask, u64 *d_out, const uint8_t *d_keep, size_t keep_period,
                            const int *d_share_dst, size_t share_plane, u64 *d_share_out, hipStream_t s) {
  if (n == 0) return MMH_OK;
  const size_t total = (n * 2) << ctx->logN;
  MMH_REQUIRE((total + 255) / 256 <= 0x7fffffffu, MMH_ERR_OUT_BOUND);
  const ReshareSink fin = make_sink(ctx, d_acc, d_e1, d_mask, d_out, d_keep, keep_period, d_share_dst, share_plane, d_share_out);
  hipLaunchKernelGGL(reshare_elem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_zero, ctx->d_mods, fin,
                     ctx->logN, total);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

int ntt_dispatch(const mmh_ctx *ctx, bool inverse, u64 *d_polys, size_t n_polys, int limbs,
                 int mod_base, hipStream_t s) {
  return dispatch(ctx, inverse ? -1 : SRC_INPLACE, d_polys, NttSource{}, n_polys, limbs, mod_base, s);
}
}  // namespace mmh

using namespace mmh;

extern "C" int mmh_ntt_forward(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limbs,
                               void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(limbs >= 1 && limbs <= ctx->L, MMH_ERR_DIM_MISMATCH);
  return ntt_dispatch(ctx, false, d_polys, n_polys, (int)limbs, 0, (hipStream_t)stream);
}
extern "C" int mmh_ntt_inverse(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limbs,
                               void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(limbs >= 1 && limbs <= ctx->L, MMH_ERR_DIM_MISMATCH);
  return ntt_dispatch(ctx, true, d_polys, n_polys, (int)limbs, 0, (hipStream_t)stream);
}
extern "C" int mmh_ntt_forward_t(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  return ntt_dispatch(ctx, false, d_polys, n_polys, 1, (int)ctx->L, (hipStream_t)stream);
}
extern "C" int mmh_ntt_inverse_t(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  return ntt_dispatch(ctx, true, d_polys, n_polys, 1, (int)ctx->L, (hipStream_t)stream);
}
// one limb at a time, as the reference's own loops call seal::util::ntt_negacyclic_harvey[_lazy] /
// inverse_ntt_negacyclic_harvey with ntt_tables[i] (hom_conv2d_ss.cc:306,315,333): every polynomial of the batch is
// transformed modulo q[limb_index]
extern "C" int mmh_ntt_forward_limb(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limb_index,
                                    void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(limb_index < ctx->L, MMH_ERR_DIM_MISMATCH);
  return ntt_dispatch(ctx, false, d_polys, n_polys, 1, (int)limb_index, (hipStream_t)stream);
}
extern "C" int mmh_ntt_inverse_limb(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limb_index,
                                    void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(limb_index < ctx->L, MMH_ERR_DIM_MISMATCH);
  return ntt_dispatch(ctx, true, d_polys, n_polys, 1, (int)limb_index, (hipStream_t)stream);
}
