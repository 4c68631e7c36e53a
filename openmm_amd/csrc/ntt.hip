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
#include "ntt_device.hpp"

namespace mmh {
// Workgroups start in launch order.  The units of a launch are laid out polynomial by polynomial ([poly][limb][N]); the
// grid walks them limb by limb instead -- all of limb 0, then all of limb 1 -- so that butterflies of one kind (64-bit
// integer, or double precision under a modulus below 2^50) have the chip to themselves for most of the launch.  Mixed
// workgroups cost the {60,49}-bit batch 3.7 % (560 -> 540 us per 32,768 transforms in steady state; N = 8192: 349 ->
// 336 us per 8,192), although either kind alone runs faster than the mix (60.8 and 72.2 M NTT/s).
__device__ __forceinline__ unsigned limb_major_unit(int limbs) {
  const unsigned b = blockIdx.x, np = gridDim.x / (unsigned)limbs;
  if (limbs <= 1 || np * (unsigned)limbs != gridDim.x) return b;
  return (b % np) * (unsigned)limbs + b / np;
}
template <int LOGN, bool FOLD, int SRC = SRC_INPLACE>
__global__ __launch_bounds__(256) void ntt_fwd_kernel(u64 *__restrict__ data,
                                                      const DevMod *__restrict__ mods,
                                                      const ulonglong2 *__restrict__ tables,
                                                      int limbs, int mod_base,
                                                      NttSource src) {
  extern __shared__ __align__(16) u64 lds[];
  ntt_fwd_unit<LOGN, FOLD, SRC>(limb_major_unit(limbs), data, mods, tables, limbs, mod_base, src, lds);
}

// The two operands of HomBNSS::bn's multiply_plain to NTT form in one launch (hom_bn_ss.cc:263-267): units
// [0, n_ct_units) are the client's ciphertexts with the server's share added on load (SRC_CT_PLAIN, written to `sum`),
// the rest lift the scale plaintexts (SRC_PLAIN, written to `pt_ntt`).
template <int LOGN, bool FOLD>
__global__ __launch_bounds__(256) void bn_front_kernel(u64 *__restrict__ sum, u64 *__restrict__ pt_ntt,
                                                       const DevMod *__restrict__ mods,
                                                       const ulonglong2 *__restrict__ tables, int limbs,
                                                       NttSource ct_src, NttSource pt_src, unsigned n_ct_units) {
  extern __shared__ __align__(16) u64 lds[];
  if (blockIdx.x < n_ct_units)
    ntt_fwd_unit<LOGN, FOLD, SRC_CT_PLAIN>(blockIdx.x, sum, mods, tables, limbs, 0, ct_src, lds);
  else
    ntt_fwd_unit<LOGN, FOLD, SRC_PLAIN>(blockIdx.x - n_ct_units, pt_ntt, mods, tables, limbs, 0, pt_src, lds);
}

// ---------------------------------------------------------------- N = 8192 with 512 threads
// The 256-thread kernels hold 32 coefficients per lane at N = 8192 and their 68 KiB image lets two workgroups share a
// CU: two waves per SIMD, too few to cover the latency of the passes.  These hold 16 per lane in 512 threads (four
// waves per SIMD with the same two workgroups): passes of 4 + 4 + 4 stages on registers with the image in LDS between
// them, and the 13th stage, whose pairs are adjacent coefficients, folded into the 16-byte-per-lane store (forward) or
// load (inverse).  In place, plain source / sink only; the other forms keep the 256-thread kernels.
//   pass A  k*512 + t            (stride 512: stages 0-3, wave-uniform twiddles)
//   pass B  jhi*512 + k*32 + jlo (jhi = t >> 5, jlo = t & 31: stages 4-7)
//   pass C  jhi*32 + k*2 + jlo   (jhi = t >> 1, jlo = t & 1: stages 8-11)
//   pass D  pairs (2P, 2P + 1), P = k*512 + t (stage 12)
template <bool FOLD>
__global__ __launch_bounds__(512, 4) void ntt_fwd512_kernel(u64 *__restrict__ data, const DevMod *__restrict__ mods,
                                                            const ulonglong2 *__restrict__ tables, int limbs,
                                                            int mod_base) {
  constexpr int N = 8192;
  extern __shared__ __align__(16) u64 lds[];
  const int t = threadIdx.x;
  const unsigned unit = limb_major_unit(limbs);
  const DevMod &M = mods[mod_base + unit % limbs];
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset;
  u64 *__restrict__ a = data + (size_t)unit * N;
  ulonglong2 *__restrict__ out = reinterpret_cast<ulonglong2 *>(a);
  if (M.f64) {  // uniform over the workgroup
    double *ldsd = reinterpret_cast<double *>(lds);
    const F64Mod m{M.qd, M.qinv};
    const double *__restrict__ twd = reinterpret_cast<const double *>(tw);
    double wB[15], wC[15];
    {
      const double bias = 0x1p52 + 2.0 * m.q;  // inputs in [0, 4q) become |v| <= 2q
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = f64_from_u64(a[k * 512 + t], bias);
      double wA[15];
      load_twiddles_f64<4, 0>(tw, 0, wA);
      load_twiddles_f64<4, 4>(tw, t >> 5, wB);
      fwd_radix_f64<4, false>(x, wA, m);
#pragma unroll
      for (int k = 0; k < 16; ++k) ldsd[pad(k * 512 + t)] = x[k];
    }
    __syncthreads();
    {
      const int base = (t >> 5) * 512 + (t & 31);
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = ldsd[pad(base + k * 32)];
      load_twiddles_f64<4, 8>(tw, t >> 1, wC);
      fwd_radix_f64<4, true>(x, wB, m);
#pragma unroll
      for (int k = 0; k < 16; ++k) ldsd[pad(base + k * 32)] = x[k];
    }
    __syncthreads();
    {
      const int base = (t >> 1) * 32 + (t & 1);
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = ldsd[pad(base + k * 2)];
      fwd_radix_f64<4, true>(x, wC, m);
#pragma unroll
      for (int k = 0; k < 16; ++k) ldsd[pad(base + k * 2)] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int P = k * 512 + t, j = pad(2 * P);
      const double X = f64_reduce(ldsd[j], m), T = f64_mulmod(twd[4096 + P], ldsd[j + 1], m);
      out[P] = make_ulonglong2(f64_canonical(X + T, m), f64_canonical(X - T, m));
    }
    return;
  }
  const Reducer<FOLD> rd(M);
  const u64 nq = M.nq;
  ulonglong2 wB[15], wC[15];
  {
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = a[k * 512 + t];
    ulonglong2 wA[15];
    load_twiddles<4, 0>(tw, 0, wA);          // wave-uniform: scalar loads
    load_twiddles<4, 4>(tw, t >> 5, wB);     // for pass B, in flight during pass A
    fwd_radix<4, 0, FOLD, true>(x, wA, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(k * 512 + t)] = x[k];
  }
  __syncthreads();
  {
    const int base = (t >> 5) * 512 + (t & 31);
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(base + k * 32)];
    load_twiddles<4, 8>(tw, t >> 1, wC);     // for pass C
    fwd_radix<4, 4, FOLD>(x, wB, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(base + k * 32)] = x[k];
  }
  __syncthreads();
  {
    const int base = (t >> 1) * 32 + (t & 1);
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(base + k * 2)];
    fwd_radix<4, 8, FOLD>(x, wC, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(base + k * 2)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int P = k * 512 + t, j = pad(2 * P);
    const ulonglong2 w = tw[4096 + P];
    const u64 X = rd.fwd_x(lds[j], 12);
    const u64 A = shoup_mad(X, lds[j + 1], w.x, w.y, nq);
    out[P] = make_ulonglong2(rd.fwd_final(A), rd.fwd_final(2 * X + rd.q4 - A));
  }
}

template <bool FOLD>
__global__ __launch_bounds__(512, 4) void ntt_inv512_kernel(u64 *__restrict__ data, const DevMod *__restrict__ mods,
                                                            const ulonglong2 *__restrict__ tables, int limbs,
                                                            int mod_base) {
  constexpr int N = 8192;
  extern __shared__ __align__(16) u64 lds[];
  const int t = threadIdx.x;
  const unsigned unit = limb_major_unit(limbs);
  const DevMod &M = mods[mod_base + unit % limbs];
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset + N;
  u64 *__restrict__ a = data + (size_t)unit * N;
  const ulonglong2 *__restrict__ in = reinterpret_cast<const ulonglong2 *>(a);
  if (M.f64) {
    double *ldsd = reinterpret_cast<double *>(lds);
    const F64Mod m{M.qd, M.qinv};
    const double *__restrict__ twd = reinterpret_cast<const double *>(tw);
    const double bias = 0x1p52 + m.q;  // inputs in [0, 2q) become |v| <= q
    double wC[15], wB[15];
    load_twiddles_f64<4, 8>(tw, t >> 1, wC);
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // stage 12 first (e = 0: sums are not reduced)
      const int P = k * 512 + t, j = pad(2 * P);
      const ulonglong2 v = in[P];
      const double X = f64_from_u64(v.x, bias), Y = f64_from_u64(v.y, bias);
      ldsd[j] = X + Y;
      ldsd[j + 1] = f64_mulmod(twd[4096 + P], X - Y, m);
    }
    __syncthreads();
    {
      const int base = (t >> 1) * 32 + (t & 1);
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = ldsd[pad(base + k * 2)];
      load_twiddles_f64<4, 4>(tw, t >> 5, wB);
      inv_radix_f64<4, 1>(x, wC, m);
#pragma unroll
      for (int k = 0; k < 16; ++k) ldsd[pad(base + k * 2)] = x[k];
    }
    __syncthreads();
    {
      const int base = (t >> 5) * 512 + (t & 31);
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = ldsd[pad(base + k * 32)];
      inv_radix_f64<4, 5>(x, wB, m);
#pragma unroll
      for (int k = 0; k < 16; ++k) ldsd[pad(base + k * 32)] = x[k];
    }
    __syncthreads();
    {
      double x[16], wA[15];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = ldsd[pad(k * 512 + t)];
      load_twiddles_f64<4, 0>(tw, 0, wA);
      inv_radix_f64<4, 9>(x, wA, m);
      const double ninv = (double)M.ninv;  // exact: below q < 2^49
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k * 512 + t] = f64_canonical(f64_mulmod(ninv, x[k], m), m);
    }
    return;
  }
  const Reducer<FOLD> rd(M);
  const u64 nq = M.nq;
  ulonglong2 wC[15], wB[15];
  load_twiddles<4, 8>(tw, t >> 1, wC);
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // stage 12 first (e = 0)
    const int P = k * 512 + t, j = pad(2 * P);
    const ulonglong2 v = in[P], w = tw[4096 + P];
    lds[j] = rd.inv_sum(v.x + v.y, 0);
    lds[j + 1] = shoup_mad(0, v.x + rd.inv_bias(0) - v.y, w.x, w.y, nq);
  }
  __syncthreads();
  {
    const int base = (t >> 1) * 32 + (t & 1);
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(base + k * 2)];
    load_twiddles<4, 4>(tw, t >> 5, wB);
    inv_radix<4, 8, 1, FOLD>(x, wC, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(base + k * 2)] = x[k];
  }
  __syncthreads();
  {
    const int base = (t >> 5) * 512 + (t & 31);
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(base + k * 32)];
    inv_radix<4, 4, 5, FOLD>(x, wB, nq, rd);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(base + k * 32)] = x[k];
  }
  __syncthreads();
  {
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(k * 512 + t)];
    ulonglong2 wA[15];
    load_twiddles<4, 0>(tw, 0, wA);
    inv_radix<4, 0, 9, FOLD, true>(x, wA, nq, rd);
    const u64 ninv = M.ninv, ninv_sh = M.ninv_sh;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k * 512 + t] = csub64(csub64(mul_shoup_lazy4(x[k], ninv, ninv_sh, nq), rd.q2), rd.q);
  }
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
  // acc is [n][2][N]: the accumulator after divide_and_round_q_last, canonical mod q0 (the matrix-core product's paired
  // items, conv.hip, write it so), instead of [n][2][2][N] at the first level
  int acc_switched;
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
  // mod_switch(acc) + zz + add, everything modulo q0 kept lazy until one canonical reduction at the end (the same
  // residue as the step-by-step divide_and_round_q_last / add / sub sequence, hence the same canonical value):
  //   last = [a1 + q1/2]_q1 (canonical: it is read as an integer), x = a0 - (last - half0) + q0 in (0, 3 q0),
  //   y = (zz + add) + x * q1^-1 lazily in [0, 5 q0) through the multiply-add chain, then one reduction.
  // zz and add are canonical mod q0.
  __device__ __forceinline__ u64 switched_plus(u64 a0, u64 a1, u64 zz, u64 add) const {
    const u64 last = csub(a1 + half, q1);
    const u64 lastq = f.qlast_small ? last : reduce64(last, q0);   // q1 < q0: already reduced
    const u64 x = a0 + (q0.q + half0) - lastq;
    const u64 y = shoup_mad_uniform(zz + add, x, f.inv_qlast, f.inv_qlast_sh, q0.nq);   // < 2 q0 + 3 q0
    if (q0.fold_c) return csub(fold_mod(y, q0.fold_c, q0.fold_shift, q0.fold_mask), q0.q);
    return csub(csub(csub(y, 4 * q0.q), 2 * q0.q), q0.q);
  }
  // the same from the switched accumulator ms = ((a0 - last + half0) q1^-1) mod q0: the same residue, one reduction
  __device__ __forceinline__ u64 switched_plus(u64 ms, u64 zz, u64 add) const {
    return csub(csub(ms + zz + add, 2 * q0.q), q0.q);   // three canonical terms (add <= q0)
  }
  __device__ __forceinline__ u64 c1_sw(u64 ms, u64 zz, int64_t e) const {
    return switched_plus(ms, zz, signed_to_mod(e, q0.q)) & f.mask1;
  }
  __device__ __forceinline__ u64 c0_sw(u64 ms, u64 zz, u64 m, bool kept) const {
    const u64 r = switched_plus(ms, zz, q0.q - m) & f.mask0;
    return kept ? r : 0;
  }
  __device__ __forceinline__ u64 c1(u64 a0, u64 a1, u64 zz, int64_t e) const {
    return switched_plus(a0, a1, zz, signed_to_mod(e, q0.q)) & f.mask1;
  }
  __device__ __forceinline__ u64 c0(u64 a0, u64 a1, u64 zz, u64 m, bool kept) const {
    const u64 r = switched_plus(a0, a1, zz, q0.q - m) & f.mask0;   // m canonical: q0 - m in (0, q0]
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
    const bool sw = f.acc_switched != 0;   // uniform over the launch
    const u64 *__restrict__ acc0 = f.acc + (sw ? (p * 2 + j) << LOGN : ((p * 2 + j) * 2) << LOGN);
    const u64 *__restrict__ acc1 = acc0 + (size_t(1) << LOGN);   // (not read when switched)
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
          a0[k] = acc0[i], a1[k] = sw ? 0 : acc1[i], e[k] = e1 ? e1[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < H; ++k)
          dst[(h + k) * 256 + t] = sw ? rc.c1_sw(a0[k], z[(h + k) * ZS], e[k]) : rc.c1(a0[k], a1[k], z[(h + k) * ZS], e[k]);
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
        a0[k] = kept[k] ? acc0[i] : 0, a1[k] = kept[k] && !sw ? acc1[i] : 0;
      }
#pragma unroll
      for (int k = 0; k < H; ++k) {
        if (d[k] >= 0) share[d[k]] = rc.share(m[k]);
        dst[(h + k) * 256 + t] = sw ? rc.c0_sw(a0[k], z[(h + k) * ZS], m[k], kept[k])
                                    : rc.c0(a0[k], a1[k], z[(h + k) * ZS], m[k], kept[k]);
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
  unsigned unit = limb_major_unit(limbs);
  if (SINK == SINK_RESHARE) {
    // Both components of an output ciphertext read the same u: workgroups go to the eight XCDs in turn, so units b
    // and b + 8 share an L2 -- give them to the two components of one ciphertext (groups of 16 units; a last partial
    // group keeps the plain order).
    const unsigned b = blockIdx.x, full = gridDim.x & ~15u;
    if (b < full) unit = (((b >> 4) << 3) + (b & 7)) * 2 + ((b >> 3) & 1);
  }
  const DevMod &M = mods[mod_base + unit % limbs];
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset + N;
  u64 *__restrict__ a = data + (size_t)unit * N;

  if (PROD) {
    const size_t kj = unit / limbs, l = unit % limbs;
    const ulonglong2 *__restrict__ in =
        reinterpret_cast<const ulonglong2 *>(src.p + (((kj >> 1) * limbs + l) << LOGN));
    const size_t kidx = ((src.p2_each ? kj : (kj & 1)) * limbs + l) << LOGN;
    const ulonglong2 *__restrict__ key = reinterpret_cast<const ulonglong2 *>(src.p2 + kidx);
    if (src.p3) {   // sum of two products (uniform over the launch)
      const ulonglong2 *__restrict__ in2 =
          reinterpret_cast<const ulonglong2 *>(src.p3 + (((kj >> 1) * limbs + l) << LOGN));
      const ulonglong2 *__restrict__ key2 = reinterpret_cast<const ulonglong2 *>(src.p4 + kidx);
      const bool fold = FOLD && mul_mod_fold_ok(M);
#pragma unroll
      for (int k = 0; k < N / 512; ++k) {
        const int j = 2 * (k * 256 + t);
        const ulonglong2 v = in[k * 256 + t], w = key[k * 256 + t], v2 = in2[k * 256 + t], w2 = key2[k * 256 + t];
        lds[pad(j)] = add_mod(fold ? mul_mod_fold(v.x, w.x, M) : mul_mod(v.x, w.x, M),
                              fold ? mul_mod_fold(v2.x, w2.x, M) : mul_mod(v2.x, w2.x, M), M.q);
        lds[pad(j) + 1] = add_mod(fold ? mul_mod_fold(v.y, w.y, M) : mul_mod(v.y, w.y, M),
                                  fold ? mul_mod_fold(v2.y, w2.y, M) : mul_mod(v2.y, w2.y, M), M.q);
      }
    } else if (FOLD && mul_mod_fold_ok(M)) {   // uniform over the workgroup
#pragma unroll
      for (int k = 0; k < N / 512; ++k) {
        const int j = 2 * (k * 256 + t);
        const ulonglong2 v = in[k * 256 + t], w = key[k * 256 + t];
        lds[pad(j)] = mul_mod_fold(v.x, w.x, M);
        lds[pad(j) + 1] = mul_mod_fold(v.y, w.y, M);
      }
    } else {
#pragma unroll
      for (int k = 0; k < N / 512; ++k) {
        const int j = 2 * (k * 256 + t);
        const ulonglong2 v = in[k * 256 + t], w = key[k * 256 + t];
        lds[pad(j)] = mul_mod(v.x, w.x, M);
        lds[pad(j) + 1] = mul_mod(v.y, w.y, M);
      }
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
  const Emitter<LOGN, SINK> emit(a, fin, mods, unit);  // (SINK_RESHARE runs with limbs == 1)
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
  const bool sw = f.acc_switched != 0;
  const size_t a = sw ? idx : (((p * 2 + j) * 2) << logN) + i;
  const u64 zz = zero[idx];
  if (j == 1) {
    const int64_t e = f.e1 ? f.e1[(p << logN) + i] : 0;
    f.out[idx] = sw ? rc.c1_sw(f.acc[a], zz, e) : rc.c1(f.acc[a], f.acc[a + (size_t(1) << logN)], zz, e);
    return;
  }
  const size_t kp = ((p % f.keep_period) << logN) + i;
  const bool kept = !f.keep || f.keep[kp] != 0;
  const int d = f.share_dst ? f.share_dst[kp] : -1;
  const u64 m = f.mask[(p << logN) + i];
  if (d >= 0) f.share_out[(p / f.keep_period) * f.share_plane + d] = rc.share(m);
  const u64 a0 = kept ? f.acc[a] : 0, a1 = kept && !sw ? f.acc[a + (size_t(1) << logN)] : 0;
  f.out[idx] = sw ? rc.c0_sw(a0, zz, m, kept) : rc.c0(a0, a1, zz, m, kept);
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
  // N = 8192 in place: the 512-thread kernels where they measure faster -- the inverse transform (445 against 470 us
  // per 2048 x {60,60,60,49}) and forward launches whose moduli all take the double-precision butterflies (126-130
  // against 132-138 us per 4096).  The forward integer butterflies are bound by the vector ALU at either occupancy
  // (1.08 T butterflies/s at N = 8192 as at N = 4096) and lose 12 % to the extra LDS pass.  MMH_NTT_512=0 / 1 forces
  // the 256- / 512-thread kernels (A/B measurements, tests of both).
  bool all_f64 = true;
  for (int l = 0; l < limbs; ++l) all_f64 = all_f64 && ctx->h_mods[mod_base + l].f64;
  if (ctx->logN == 13 && (mode == SRC_INPLACE || mode == -1)) {
    const char *e = getenv("MMH_NTT_512");
    const bool use512 = e ? e[0] != '0' : (mode == -1 || all_f64);
    if (use512) {
      auto go = [&](auto kern) -> int {
        MMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes<13>()));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds_bytes<13>(), s, d, ctx->d_mods, ctx->d_tables,
                           limbs, mod_base);
        MMH_HIP(hipGetLastError());
        return MMH_OK;
      };
      if (mode == -1) return fold ? go(ntt_inv512_kernel<true>) : go(ntt_inv512_kernel<false>);
      return fold ? go(ntt_fwd512_kernel<true>) : go(ntt_fwd512_kernel<false>);
    }
  }
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

// sum[k][j][l] = NTT_l(cts[k][j][l] + [j = 0] round(Q share[k] / t)), pt_ntt[k][l] = NTT_l(lift_l(scales[k])) at the
// first level, one launch (bn_front_kernel)
template <int LOGN>
static int bn_front_launch(const mmh_ctx *ctx, bool fold, u64 *d_sum, u64 *d_pt_ntt, const NttSource &cs,
                           const NttSource &ps, size_t n, hipStream_t s) {
  auto kern = fold ? bn_front_kernel<LOGN, true> : bn_front_kernel<LOGN, false>;
  const int limbs = (int)ctx->L;
  if (lds_bytes<LOGN>() > 48 * 1024)
    MMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes<LOGN>()));
  hipLaunchKernelGGL(kern, dim3((unsigned)(n * 3 * limbs)), dim3(256), lds_bytes<LOGN>(), s, d_sum, d_pt_ntt, ctx->d_mods,
                     ctx->d_tables, limbs, cs, ps, (unsigned)(n * 2 * limbs));
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
int ntt_bn_front_dispatch(const mmh_ctx *ctx, const u64 *d_cts, const u64 *d_share_pts, const u64 *d_scale_pts, size_t n,
                          u64 *d_sum, u64 *d_pt_ntt, hipStream_t s) {
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(ctx->d_bfv && n * 3 * ctx->L <= 0x7fffffffu, MMH_ERR_OUT_BOUND);
  NttSource cs{}, ps{};
  cs.p = d_cts, cs.p2 = d_share_pts, cs.bfv = ctx->d_bfv;
  ps.p = d_scale_pts, ps.t = ctx->t, ps.thr = ctx->bfv.thr;
  bool fold = true;
  for (size_t l = 0; l < ctx->L; ++l) fold = fold && ctx->h_mods[l].fold_c != 0;
  switch (ctx->logN) {
    case 12: return bn_front_launch<12>(ctx, fold, d_sum, d_pt_ntt, cs, ps, n, s);
    case 13: return bn_front_launch<13>(ctx, fold, d_sum, d_pt_ntt, cs, ps, n, s);
    default: MMH_FAIL(MMH_ERR_CONFIG, "only N = 4096 and N = 8192 are built");
  }
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

// out[k][j][l] = INTT_l(pt_ntt[k][l] (.) ct_ntt[k][j][l] (+ pt2_ntt[k][l] (.) ct2_ntt[k][j][l])) for the first `limbs`
// moduli: multiply_plain (and the sum of two) with the dyadic product formed while the inverse transform loads
// (HomBNSS::bn, hom_bn_ss.cc:265-267; ::bole :304-306)
int ntt_inv_product_each_dispatch(const mmh_ctx *ctx, const u64 *d_pt_ntt, const u64 *d_ct_ntt, const u64 *d_pt2_ntt,
                                  const u64 *d_ct2_ntt, u64 *d_out, size_t n, int limbs, hipStream_t s) {
  NttSource src{};
  src.p = d_pt_ntt, src.p2 = d_ct_ntt, src.p3 = d_pt2_ntt, src.p4 = d_ct2_ntt, src.p2_each = 1;
  return dispatch(ctx, -2, d_out, src, n * 2, limbs, 0, s);
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
                             u64 *d_share_out, bool acc_switched) {
  const BfvConsts &b = ctx->bfv;
  const u64 q0 = ctx->q[0], g = b.gamma.q;
  ReshareSink f{d_acc, d_e1, d_mask, d_out, b.inv_qlast_mod_q0, b.mask0, b.mask1, d_keep, keep_period ? keep_period : 1,
                d_share_dst, share_plane, d_share_out, b, 0, 0, 0, 0, 0, 0, 0, acc_switched ? 1 : 0};
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
                            hipStream_t s, bool acc_switched) {
  NttSource src{};
  src.p = d_u_ntt, src.p2 = d_key;
  const ReshareSink fin = make_sink(ctx, d_acc, d_e1, d_mask, d_out, d_keep, keep_period, d_share_dst, share_plane, d_share_out,
                                    acc_switched);
  return dispatch(ctx, -3, d_out, src, n * 2, 1, 0, s, &fin);
}

// The same tail from an encryption of zero in memory, d_zero [n][2][N] (coefficient form at the last level).
int reshare_online_dispatch(const mmh_ctx *ctx, const u64 *d_zero, size_t n, const u64 *d_acc, const int64_t *d_e1,
                            const u64 *d_mask, u64 *d_out, const uint8_t *d_keep, size_t keep_period,
                            const int *d_share_dst, size_t share_plane, u64 *d_share_out, hipStream_t s,
                            bool acc_switched) {
  if (n == 0) return MMH_OK;
  const size_t total = (n * 2) << ctx->logN;
  MMH_REQUIRE((total + 255) / 256 <= 0x7fffffffu, MMH_ERR_OUT_BOUND);
  const ReshareSink fin = make_sink(ctx, d_acc, d_e1, d_mask, d_out, d_keep, keep_period, d_share_dst, share_plane, d_share_out,
                                    acc_switched);
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
