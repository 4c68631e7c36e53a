// Shared declarations of libmm_hip: context layout, error plumbing, device-side
// 64-bit modular arithmetic (Shoup / Barrett on VALU; no MFMA on this path).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mm_hip.h"

namespace mmh {
using u64 = uint64_t;
using u32 = uint32_t;

constexpr int kMaxLimbs = 4;

// One modulus as the kernels see it.  Twiddles are (w, floor(w*2^64/q)) pairs indexed
// like SEAL's bit-reversed root-power tables: fwd[m + i] = psi^{bitrev(m+i)}.
struct DevMod {
  u64 q;
  u64 nq;                      // 2^64 - q
  u64 barrett_hi, barrett_lo;  // floor(2^128 / q)
  u64 ninv, ninv_sh;           // N^-1 mod q and its Shoup quotient
  const ulonglong2 *fwd;       // length N
  const ulonglong2 *inv;       // length N, inverses of fwd
  u64 tw_offset;               // fwd = tables + tw_offset, inv = fwd + N (one allocation per context)
  // q = 2^b - c with c < 2^31 (every prime SEAL's generator emits): v mod q ~ (v mod 2^b) + (v >> b)*c
  u32 fold_c;                  // c, or 0 when q does not have that shape
  u32 fold_shift;              // b - 32
  u32 fold_mask;               // 2^(b-32) - 1
  // q < 2^49: the NTT runs in double precision (exact integer arithmetic through FMA, ntt.hip) and
  // its table halves hold N doubles (double)w each, compactly, instead of N (w, Shoup quotient) pairs
  u32 f64;
  double qd, qinv;             // (double)q and 1/q rounded to nearest
};

// A modulus without NTT tables (plain modulus t, BEHZ gamma): Barrett only.
struct ModC {
  u64 q;
  u64 barrett_hi, barrett_lo;
};

// BFV / BEHZ constants of one level (SEAL ContextData + RNSTool), passed to kernels by value.
struct LevelConsts {
  int limbs;
  u64 delta_mod_q[2];         // floor(Q/t) mod q_l
  u64 q_mod_t;                // Q mod t
  u64 prod_t_gamma_mod_q[2];  // t*gamma mod q_l
  u64 inv_punct_mod_q[2];     // (Q/q_l)^-1 mod q_l
  u64 punct_mod_t[2];         // Q/q_l mod t
  u64 punct_mod_gamma[2];     // Q/q_l mod gamma
  u64 neg_inv_q_mod_t, neg_inv_q_mod_gamma;
};
struct BfvConsts {
  ModC t, gamma;
  u64 thr;              // (t+1)/2: plain_upper_half_threshold
  int t_log2;           // log2(t) when t is a power of two, else -1
  u64 t_inv64;          // t^-1 mod 2^64 for odd t (exact division)
  u64 inv_gamma_mod_t;
  u64 inv_qlast_mod_q0; // q1^-1 mod q0 (mod-switch to the last level)
  u64 mask0, mask1;     // truncate_for_decryption
  LevelConsts lv[2];
};

void set_error(const std::string &msg);
int ctx_device(const mmh_ctx *ctx);
// call-scoped device scratch from a cache of blocks reused in stream order (context.hip); same
// calling convention as hipMallocAsync / hipFreeAsync
hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t s);
hipError_t scratch_free(void *p, hipStream_t s);

#define MMH_FAIL(code, msg)                                                    \
  do {                                                                         \
    ::mmh::set_error(std::string(__func__) + ": " + (msg));                    \
    return (code);                                                             \
  } while (0)

#define MMH_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) MMH_FAIL(MMH_ERR_INTERNAL, std::string(#expr " -> ") + hipGetErrorString(e_)); \
  } while (0)

#define MMH_REQUIRE(cond, code)                                                \
  do {                                                                         \
    if (!(cond)) MMH_FAIL(code, "requirement failed: " #cond);                 \
  } while (0)

// Every entry point that takes a context or a plan runs on that context's device: the guard makes it current for
// the length of the call and puts the caller's device back afterwards (torch and other libraries in the process keep
// their own notion of the current device).  hipGetDevice is a thread-local read; hipSetDevice is only issued when
// the devices differ.  The stream argument must then belong to the context's device: the caller's business.
struct DeviceScope {
  int prev = -1, rc = MMH_OK;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (device < 0) return;  // null context: the entry point reports it
    (void)hipGetLastError();  // an error another library left behind is not this call's: launch checks start clean
    if (hipGetDevice(&prev) != hipSuccess) {
      rc = MMH_ERR_INTERNAL;
      return;
    }
    if (prev != device) {
      if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();  // do not leave the failure behind for the next launch check to find
        rc = MMH_ERR_CONFIG;
      } else {
        switched = true;
      }
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};
#define MMH_DEVICE_SCOPE(ctxp)                                                              \
  const mmh_ctx *dev_scope_ctx_ = (ctxp);                                                   \
  ::mmh::DeviceScope dev_scope_(dev_scope_ctx_ ? ::mmh::ctx_device(dev_scope_ctx_) : -1);   \
  if (dev_scope_.rc) MMH_FAIL(dev_scope_.rc, "cannot make the context's device current")

// ---------------------------------------------------------------- device math
__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

// x*w mod q in [0, 2q) for any 64-bit x, given wsh = floor(w*2^64/q)
__device__ __forceinline__ u64 mul_shoup_lazy(u64 x, u64 w, u64 wsh, u64 q) {
  return x * w - mulhi64(x, wsh) * q;
}
__device__ __forceinline__ u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }

// v_mad_u64_u32 is the cheapest wide multiply on gfx950 (measured 4 cycles per wave64 against
// 8 for v_mul_lo_u32 / v_mul_hi_u32, profiles/r01_a_valu_probe.txt), and chains of it need no
// register shuffling because the 64-bit result feeds the next addend directly.  hipcc narrows a
// C-level (u64)a*b+c whose high half is unused back to v_mul_lo_u32, hence the asm form.
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) {
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ u64 mul32(u32 a, u32 b) {
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }

// low 32 bits of a0*b0 + a1*b1 + a2*b2 + a3*b3 as one dependent chain of four mads (one asm
// statement so that hipcc pads it once, not after every instruction)
__device__ __forceinline__ u32 dot4_lo(u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3) {
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\t"
      "v_mad_u64_u32 %0, %1, %4, %5, %0\n\t"
      "v_mad_u64_u32 %0, %1, %6, %7, %0\n\t"
      "v_mad_u64_u32 %0, %1, %8, %9, %0"
      : "=&v"(d), "=&s"(carry)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
  return (u32)d;
}
__device__ __forceinline__ u64 make64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// base + x*w mod q with the product lazily in [0, 3q): Shoup multiplication whose accumulator
// starts at `base`, so that the butterfly's addition costs nothing.  nq = 2^64 - q.
// The quotient floor(x*wsh / 2^64) drops only the low partial product x0*s0 (short by at most 1)
// and keeps the carry of the two middle products on its high half.  12 VALU instructions.
// TW is "v" for per-lane twiddles and "s" when the twiddle is wave-uniform (pass A), which saves
// the SGPR->VGPR copies; every instruction reads at most one SGPR.
#define MMH_SHOUP_MAD_BODY(TW)                                                                        \
  const u32 x0 = lo32(x), x1 = hi32(x);                                                               \
  u64 m, h, cy, cyd, t, acc;                                                                          \
  u32 h1, hi;                                                                                         \
  /* three independent chains interleaved so that no instruction waits on its predecessor:       */  \
  /*   quotient  m = x1*s0 + x0*s1 (carry -> cy), h = x1*s1 + (m >> 32)                          */  \
  /*   low word  t = base + x0*w0                                                                */  \
  /*   cross     acc = x0*w1 + x1*w0   (only its low 32 bits are used)                           */  \
  asm("v_mad_u64_u32 %0, %5, %7, %10, 0\n\t"                                                         \
      "v_mad_u64_u32 %2, %5, %6, %8, %12\n\t"                                                        \
      "v_mad_u64_u32 %0, %4, %6, %11, %0\n\t"                                                        \
      "v_mad_u64_u32 %3, %5, %6, %9, 0\n\t"                                                          \
      "v_lshrrev_b64 %0, 32, %0\n\t"                                                                 \
      "v_mad_u64_u32 %3, %5, %7, %8, %3\n\t"                                                         \
      "v_mad_u64_u32 %1, %5, %7, %11, %0"                                                              \
      : "=&v"(m), "=&v"(h), "=&v"(t), "=&v"(acc), "=&s"(cy), "=&s"(cyd)                               \
      : "v"(x0), "v"(x1), TW(lo32(w)), TW(hi32(w)), TW(lo32(wsh)), TW(hi32(wsh)), "v"(base));         \
  /* h1 = hi(h) + carry; t += h0*nq0; acc += h0*nq1 + h1*nq0                                     */  \
  asm("v_addc_co_u32_e64 %0, %3, 0, %4, %6\n\t"                                                      \
      "v_mad_u64_u32 %1, %3, %5, %7, %1\n\t"                                                         \
      "v_mad_u64_u32 %2, %3, %5, %8, %2\n\t"                                                         \
      "v_mad_u64_u32 %2, %3, %0, %7, %2"                                                               \
      : "=&v"(h1), "+v"(t), "+v"(acc), "=&s"(cyd)                                                     \
      : "v"(hi32(h)), "v"(lo32(h)), "s"(cy), "s"(lo32(nq)), "s"(hi32(nq)));                           \
  /* opaque add: hipcc otherwise widens it to a 64-bit add of {0, acc} (three instructions) */        \
  asm("v_add_u32 %0, %1, %2" : "=v"(hi) : "v"(hi32(t)), "v"(lo32(acc)));                              \
  return make64(lo32(t), hi);

__device__ __forceinline__ u64 shoup_mad(u64 base, u64 x, u64 w, u64 wsh, u64 nq) { MMH_SHOUP_MAD_BODY("v") }
__device__ __forceinline__ u64 shoup_mad_uniform(u64 base, u64 x, u64 w, u64 wsh, u64 nq) { MMH_SHOUP_MAD_BODY("s") }
__device__ __forceinline__ u64 mul_shoup_lazy4(u64 x, u64 w, u64 wsh, u64 nq) { return shoup_mad(0, x, w, wsh, nq); }

// Partial reduction for q = 2^b - c: any 64-bit v becomes (v mod 2^b) + (v >> b)*c = v (mod q),
// below 2q whenever c*2^(64-b) < q - 2^b + q (checked when the context is built).  3 VALU.
__device__ __forceinline__ u64 fold_mod(u64 v, u32 c, u32 shift, u32 mask) {
  u32 k, hm;  // opaque steps: hipcc re-associates the C form into five instructions
  asm("v_lshrrev_b32 %0, %1, %2" : "=v"(k) : "s"(shift), "v"(hi32(v)));
  asm("v_and_b32 %0, %1, %2" : "=v"(hm) : "s"(mask), "v"(hi32(v)));
  u64 r, cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cy) : "v"(k), "s"(c), "v"(make64(lo32(v), hm)));
  return r;
}
__device__ __forceinline__ u64 mul_shoup(u64 x, u64 w, u64 wsh, u64 q) {
  return csub(mul_shoup_lazy(x, w, wsh, q), q);
}
__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Barrett reduction of a 128-bit value (hi:lo) with ratio = floor(2^128/q); result in [0,q).
// Same quotient estimate as SEAL util::barrett_reduce_128.
__device__ __forceinline__ u64 barrett128(u64 hi, u64 lo, u64 q, u64 r_hi, u64 r_lo) {
  // floor((hi:lo) * ratio / 2^128), dropping the lo*r_lo low word like SEAL does
  u64 carry = mulhi64(lo, r_lo);
  u64 t_lo = lo * r_hi, t_hi = mulhi64(lo, r_hi);
  u64 s = t_lo + carry;
  u64 tmp3 = t_hi + (s < t_lo);
  u64 u_lo = hi * r_lo, u_hi = mulhi64(hi, r_lo);
  u64 s2 = s + u_lo;
  u64 c2 = u_hi + (s2 < s);
  u64 quo = hi * r_hi + tmp3 + c2;
  u64 r = lo - quo * q;
  return csub(r, q);
}
__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, const DevMod &m) {
  return barrett128(mulhi64(a, b), a * b, m.q, m.barrett_hi, m.barrett_lo);
}
// One step of x mod (2^b - c) on a value held in 32-bit words w3:w2:w1:w0: (x mod 2^b) + (x >> b) * c with
// b = 32 + s, 8 <= s <= 30, w3 < 2^s.  Funnel shifts and two v_mad_u64_u32: a 64-bit shift by a register amount
// costs several times as much.  The result is below 2^96 (w3 becomes 0).
__device__ __forceinline__ void fold_words(u32 &w0, u32 &w1, u32 &w2, u32 w3, u32 c, u32 s, u32 smask) {
  const u32 h0 = __builtin_amdgcn_alignbit(w2, w1, s), h1 = __builtin_amdgcn_alignbit(w3, w2, s);
  const u64 t0 = mad32(h0, c, make64(w0, w1 & smask));   // below 2^64: the low part < 2^62, the product < 2^63
  const u64 t1 = mul32(h1, c);
  const u64 m1 = (u64)hi32(t0) + lo32(t1);
  w0 = lo32(t0), w1 = lo32(m1), w2 = hi32(t1) + hi32(m1);
}

// a * b mod q, canonical, for q = 2^b - c with 40 <= b <= 62 (fold_shift 8..30) and a, b < q: the 128-bit product on
// 32-bit words (four v_mad_u64_u32), two word folds, one 64-bit fold: ~25 VALU instructions against ~65 for the generic
// Barrett form above (five 64-bit high products).  Same canonical value.
__device__ __forceinline__ u64 mul_mod_fold(u64 a, u64 b, const DevMod &m) {
  const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
  const u64 t0 = mul32(a0, b0);
  const u64 t1 = mad32(a0, b1, hi32(t0));
  const u64 t2 = mad32(a1, b0, lo32(t1));
  const u64 t3 = mad32(a1, b1, (u64)hi32(t1) + hi32(t2));
  u32 w0 = lo32(t0), w1 = lo32(t2), w2 = lo32(t3);
  fold_words(w0, w1, w2, hi32(t3), m.fold_c, m.fold_shift, m.fold_mask);
  fold_words(w0, w1, w2, 0, m.fold_c, m.fold_shift, m.fold_mask);
  return csub(fold_mod(make64(w0, w1), m.fold_c, m.fold_shift, m.fold_mask), m.q);
}
__device__ __forceinline__ bool mul_mod_fold_ok(const DevMod &m) { return m.fold_c != 0 && m.fold_shift >= 8 && m.fold_shift <= 30; }
__device__ __forceinline__ u64 reduce64(u64 a, const DevMod &m) {
  u64 r = a - mulhi64(a, m.barrett_hi) * m.q;
  return csub(r, m.q);
}
}  // namespace mmh

// The opaque context of the C ABI.
struct mmh_ctx {
  int device = 0;
  int n_cu = 256;   // compute units of the device (grid size of the persistent product kernel)
  size_t N = 0;
  int logN = 0;
  size_t L = 0;
  mmh::u64 q[mmh::kMaxLimbs] = {0};
  mmh::u64 psi[mmh::kMaxLimbs] = {0};
  mmh::u64 t = 0;
  bool t_pow2 = false;
  bool batching = false;
  mmh::u64 t_psi = 0;
  mmh::u64 gamma = 0;
  // device side: mods[0..L) = coefficient moduli, mods[L] = plain modulus when batching
  mmh::DevMod *d_mods = nullptr;
  ulonglong2 *d_tables = nullptr;  // all twiddle tables, [(fwd,inv) per modulus][N]
  mmh::DevMod h_mods[mmh::kMaxLimbs + 1];
  mmh::BfvConsts bfv;              // valid when t >= 2
  uint32_t *d_slot_map = nullptr;  // BatchEncoder index map (batching only)
  mmh::BfvConsts *d_bfv = nullptr; // bfv again, in device memory (kernels that take it by pointer: ntt_device.hpp SRC_CT_PLAIN)
  std::vector<void *> allocs;
};
namespace mmh {
inline int ctx_device(const mmh_ctx *ctx) { return ctx->device; }
}  // namespace mmh
