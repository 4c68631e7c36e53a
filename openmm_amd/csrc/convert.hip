// Modulus conversion between Z_{2^l} and Z_p: the LOCAL share arithmetic around the single
// COT call of SCI's Truncation::{ring_to_field, field_to_ring, field_to_ring_with_truncate,
// ring_to_ring, ring_to_ring128} (SCI/src/BuildingBlocks/truncation.cpp:319-695) and the BOLE
// cross term (SCI/src/cheetah/cheetah-api.cpp:648-649).  Pure streaming kernels (HBM-bound,
// ~25 B per element); the COT between them is the caller's (host, or cot.hip's device post-processing).
// party: 1 = ALICE (COT sender), 2 = BOB.
#include "common.hpp"
#include "convert_device.hpp"

namespace mmh {
// The streaming kernels below take V = 2 elements per lane (16-byte loads and stores, 2 for the bit arrays) when every
// array they touch is 16-byte aligned, V = 1 otherwise (aligned_pairs below); the last lane of an odd n does one.
template <int V>
__device__ __forceinline__ void load_elems(const u64 *__restrict__ p, size_t i, bool pair, u64 (&r)[2]) {
  if (V == 2 && pair) {
    const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(p + i);
    r[0] = t.x, r[1] = t.y;
  } else {
    r[0] = p[i], r[1] = 0;
  }
}
template <int V>
__device__ __forceinline__ void store_elems(u64 *__restrict__ p, size_t i, bool pair, const u64 (&r)[2]) {
  if (V == 2 && pair) *reinterpret_cast<ulonglong2 *>(p + i) = make_ulonglong2(r[0], r[1]);
  else p[i] = r[0];
}
template <int V>
__device__ __forceinline__ void load_bits(const uint8_t *__restrict__ p, size_t i, bool pair, uint8_t (&r)[2]) {
  if (V == 2 && pair) {
    const uchar2 t = *reinterpret_cast<const uchar2 *>(p + i);
    r[0] = t.x, r[1] = t.y;
  } else {
    r[0] = p[i], r[1] = 0;
  }
}
template <int V>
__device__ __forceinline__ void store_bits(uint8_t *__restrict__ p, size_t i, bool pair, const uint8_t (&r)[2]) {
  if (V == 2 && pair) *reinterpret_cast<uchar2 *>(p + i) = make_uchar2(r[0], r[1]);
  else p[i] = r[0];
}

// mode 0: ring (+2^{bw-2}, msb = bit bw-1); mode 1: field (+p/4 mod p, msb = tmp >= p/2+1)
template <int MODE, int V>
__global__ __launch_bounds__(256) void pre_kernel(int party, const u64 *__restrict__ in, size_t n, int bw, u64 p,
                                                  u64 corr_value, int corr_is_not_msb, u64 *__restrict__ tmp,
                                                  uint8_t *__restrict__ msb, u64 *__restrict__ corr) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= n) return;
  const bool pair = i + 1 < n;
  u64 v[2], c[2];
  uint8_t mb[2];
  load_elems<V>(in, i, pair, v);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    bool m;
    if (MODE == 0) {
      if (party == kAlice) v[e] = (v[e] + (u64(1) << (bw - 2))) & bw_mask(bw);
      m = (v[e] >> (bw - 1)) & 1;
    } else {
      v[e] = f2r_pre_elem(party, v[e], p, m);
    }
    mb[e] = m;
    c[e] = corr_is_not_msb ? (u64)(!m) : (m ? 0 : corr_value);
  }
  store_elems<V>(tmp, i, pair, v);
  store_bits<V>(msb, i, pair, mb);
  if (party == kAlice && corr) store_elems<V>(corr, i, pair, c);
}

template <int V>
__global__ __launch_bounds__(256) void r2f_post_kernel(int party, const u64 *__restrict__ tmp,
                                                       const uint8_t *__restrict__ msb, const u64 *__restrict__ y,
                                                       size_t n, u64 p, u64 cv, u64 big_mod_p,
                                                       u64 *__restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= n) return;
  const bool pair = i + 1 < n;
  u64 t[2], yy[2], o[2];
  uint8_t mb[2];
  load_elems<V>(tmp, i, pair, t);
  load_elems<V>(y, i, pair, yy);
  load_bits<V>(msb, i, pair, mb);
#pragma unroll
  for (int e = 0; e < V; ++e) o[e] = r2f_post_elem(party, t[e], mb[e], yy[e], p, cv, big_mod_p);
  store_elems<V>(out, i, pair, o);
}

// truncating = 0: truncation.cpp:609-611,:626-634; truncating = 1: :667-669,:684-692
template <int V>
__global__ __launch_bounds__(256) void f2r_post_kernel(int party, const u64 *__restrict__ tmp,
                                                       const uint8_t *__restrict__ msb, const u64 *__restrict__ y,
                                                       size_t n, u64 p, int bw, int shift, int truncating,
                                                       u64 *__restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= n) return;
  const bool pair = i + 1 < n;
  const u64 mask = bw_mask(bw), big = p / 4, cv = truncating ? p >> shift : p;
  u64 t[2], yv[2], o[2];
  uint8_t mb[2];
  load_elems<V>(tmp, i, pair, t);
  load_elems<V>(y, i, pair, yv);
  load_bits<V>(msb, i, pair, mb);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    u64 yy = yv[e];
    if (party == kAlice) yy = mb[e] ? ((cv - yy) & mask) : ((0 - yy) & mask);
    if (!truncating) o[e] = party == kAlice ? t[e] - yy - big : t[e] - yy;
    else o[e] = party == kAlice ? ((t[e] - big) >> shift) - yy : (t[e] >> shift) - yy + 1;
    o[e] &= mask;
  }
  store_elems<V>(out, i, pair, o);
}

// truncation.cpp:353-355,:371-379 (64-bit) and :414-416,:434-442 (128-bit, WIDE)
template <bool WIDE>
__global__ __launch_bounds__(256) void r2r_post_kernel(int party, const u64 *__restrict__ tmp,
                                                       const uint8_t *__restrict__ msb, const u64 *__restrict__ y,
                                                       size_t n, int bw_a, int bw_b, u64 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 big = u64(1) << (bw_a - 2);
  u64 yy = y[i];
  if (party == kAlice) yy = ((u64)msb[i] - yy) & bw_mask(bw_b - bw_a);
  if (!WIDE) {
    u64 o = tmp[i] - (yy << bw_a);
    if (party == kAlice) o -= big;
    out[i] = o & bw_mask(bw_b);
  } else {
    // 128-bit: (tmp - (yy << bw_a) - big) mod 2^bw_b, bw_a in [2,64], bw_b in (bw_a, 128]
    const u64 s_lo = bw_a >= 64 ? 0 : yy << bw_a;
    const u64 s_hi = bw_a >= 64 ? yy : (bw_a == 0 ? 0 : yy >> (64 - bw_a));
    u64 lo = tmp[i] - s_lo;
    u64 hi = 0 - s_hi - (tmp[i] < s_lo);
    if (party == kAlice) {
      hi -= lo < big;
      lo -= big;
    }
    if (bw_b <= 64) {
      lo &= bw_mask(bw_b);
      hi = 0;
    } else {
      hi &= bw_mask(bw_b - 64);
    }
    out[2 * i] = lo;
    out[2 * i + 1] = hi;
  }
}

// out = (out + (x mod p) * y mod p) mod p with the reference's wrapping u64 addition
// Truncation::truncate / truncate_msb / truncate_msb0 of the USE_CHEETAH build (truncation-cheetah.cpp:2-218):
// the share arithmetic before the wrap sub-protocol (heuristic offset on BOB :23-29, signed offset on ALICE
// :42-47, operand of mill->compare :51-56) ...
template <int V>
__global__ __launch_bounds__(256) void trunc_pre_kernel(int party, const u64 *__restrict__ in, size_t n, int bw,
                                                        int signed_arith, int heuristic, u64 *__restrict__ adj,
                                                        u64 *__restrict__ cmp) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= n) return;
  const bool pair = i + 1 < n;
  const u64 mask = bw_mask(bw);
  u64 v[2], c[2];
  load_elems<V>(in, i, pair, v);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    if (heuristic && party != kAlice) v[e] = (v[e] + (u64(1) << (bw - 3))) & mask;
    if (signed_arith && party == kAlice) v[e] = (v[e] + (u64(1) << (bw - 1))) & mask;
    const u64 w = v[e] & mask;
    c[e] = party == kAlice ? w : (mask - w) & mask;
  }
  store_elems<V>(adj, i, pair, v);
  if (cmp) store_elems<V>(cmp, i, pair, c);
}
// ... and after B2A (:64-75 = :133-144 = :201-212; heuristic :32-34); aw = arithmetic share of the wrap bit
template <int V>
__global__ __launch_bounds__(256) void trunc_post_kernel(int party, const u64 *__restrict__ adj,
                                                         const u64 *__restrict__ aw, size_t n, int shift, int bw,
                                                         int signed_arith, int heuristic, u64 *__restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= n) return;
  const bool pair = i + 1 < n;
  const u64 mask = bw_mask(bw);
  u64 a[2], w[2], o[2];
  load_elems<V>(adj, i, pair, a);
  load_elems<V>(aw, i, pair, w);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    o[e] = (((a[e] >> shift) & bw_mask(bw - shift)) - (u64(1) << (bw - shift)) * w[e]) & mask;
    if (signed_arith && party == kAlice) o[e] = (o[e] - (u64(1) << (bw - shift - 1))) & mask;
    if (heuristic && party != kAlice) o[e] = (o[e] - (u64(1) << (bw - 3 - shift))) & mask;
  }
  store_elems<V>(out, i, pair, o);
}
// Truncation::truncate_and_reduce (truncation.cpp:289-317): aw == nullptr gives the operand of
// wrap_computation (:300-304), else the result in the smaller ring (:311-313)
__global__ __launch_bounds__(256) void trunc_reduce_kernel(const u64 *__restrict__ in, const u64 *__restrict__ aw,
                                                           size_t n, int shift, int bw, u64 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = aw ? ((in[i] >> shift) + aw[i]) & bw_mask(bw - shift) : in[i] & bw_mask(shift);
}

// ReLURingProtocol::relu of the USE_CHEETAH build (SCI/src/NonLinear/relu-ring.h:216-252, :308-313): msb of the
// share and the operand of millionaire->compare before it, the DReLU share and the multiplexer's selector after
__global__ __launch_bounds__(256) void relu_pre_kernel(int party, const u64 *__restrict__ share, size_t n, int l,
                                                       uint8_t *__restrict__ msb, u64 *__restrict__ cmp) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 cut = (u64(1) << (l - 1)) - 1, v = share[i];
  msb[i] = (uint8_t)(v >> (l - 1));
  cmp[i] = party == kAlice ? (v & cut) : cut - (v & cut);
}
__global__ __launch_bounds__(256) void relu_post_kernel(int party, const uint8_t *__restrict__ msb,
                                                        const uint8_t *__restrict__ wrap, size_t n,
                                                        uint8_t *__restrict__ drelu, uint8_t *__restrict__ sel) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t d = (uint8_t)(((int)msb[i] + (int)wrap[i]) % 2);
  drelu[i] = d;
  sel[i] = party == kAlice ? d ^ 1 : d;
}
// AuxProtocols::multiplexer (aux-protocols.cpp:62-93): data_r == nullptr gives the COT correlation (:76-78),
// else the output (:86-88)
__global__ __launch_bounds__(256) void mux_kernel(const uint8_t *__restrict__ sel, const u64 *__restrict__ x,
                                                  const u64 *__restrict__ data_r, const u64 *__restrict__ data_s,
                                                  size_t n, int bw_y, u64 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 s = sel[i];
  out[i] = (data_r ? x[i] * s + data_r[i] - data_s[i] : x[i] * (1 - 2 * s)) & bw_mask(bw_y);
}
// AuxProtocols::B2A (aux-protocols.cpp:95-126): y == nullptr gives ALICE's correlation -2x (:107-109), else the
// outputs (:112-114, :119-121; the bit itself for bw_y = 1, :97-102)
__global__ __launch_bounds__(256) void b2a_kernel(int party, const uint8_t *__restrict__ x,
                                                  const u64 *__restrict__ y, size_t n, int bw_y, int post,
                                                  u64 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 b = x[i], mask = bw_mask(bw_y);
  if (!post) out[i] = (u64(0) - 2 * b) & mask;
  else if (bw_y == 1) out[i] = b & 1;
  else out[i] = (party == kAlice ? b - y[i] : b + y[i]) & mask;
}

__global__ __launch_bounds__(256) void cross_term_kernel(u64 *__restrict__ out, const u64 *__restrict__ x,
                                                         const u64 *__restrict__ y, size_t n, u64 p, u64 r_hi,
                                                         u64 r_lo) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 a = csub(x[i] - mulhi64(x[i], r_hi) * p, p), b = csub(y[i] - mulhi64(y[i], r_hi) * p, p);
  const u64 m = barrett128(mulhi64(a, b), a * b, p, r_hi, r_lo);
  const u64 s = out[i] + m;
  out[i] = csub(s - mulhi64(s, r_hi) * p, p);
}

static inline unsigned nblocks(size_t n) { return (unsigned)((n + 255) / 256); }
static int check_common(int party, size_t n, const void *a, const void *b, const void *c) {
  MMH_REQUIRE(party == 1 || party == 2, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(n == 0 || (a && b && c), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(n <= 0x7fffffffull * 256, MMH_ERR_OUT_BOUND);
  return MMH_OK;
}
// FusedBN filter scaling (networks/main_resnet50.cpp:1736-1753): scaled = round(f/2^k * s/2^k * 2^k) in double
// arithmetic, exactly as written (the division by 2^k and the last product are exact; the middle product rounds to
// nearest, as v_mul_f64 does; no addition follows a product, so nothing can contract into an FMA), then sci::neg_mod.
// One thread per filter entry; OIHW = 1 writes [co][ci][fh][fw] (what the conv entry points read), else the
// reference's [fh][fw][ci][co].  *any_nonzero becomes 1 when some scaled entry is not zero (:1754).
#pragma clang fp contract(off)
__device__ __forceinline__ long long sign_value(u64 x, int bitlength) {  // getSignValue, main_resnet50.cpp:21-28
  const long long upper = 1ll << bitlength;
  const u64 mask = (u64)upper - 1, half = (u64)upper >> 1;
  x &= mask;
  return (long long)(x >= half ? x - (u64)upper : x);
}
__global__ __launch_bounds__(256) void fused_bn_filters_kernel(const u64 *__restrict__ filters,
                                                               const u64 *__restrict__ scales, size_t total, int fh,
                                                               int fw, int ci, int co, int bitlength, double scale,
                                                               long long prime_mod, int oihw, u64 *__restrict__ out,
                                                               u32 *__restrict__ any_nonzero) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  int nz = 0;
  if (idx < total) {
    const int m = (int)(idx % co);
    const double bn_scale = (double)sign_value(scales[m], bitlength) / scale;
    const double fval = (double)sign_value(filters[idx], bitlength) / scale;
    const long long f64 = (long long)round(fval * bn_scale * scale);
    const u64 v = (u64)(((f64 % prime_mod) + prime_mod) % prime_mod);
    size_t o = idx;
    if (oihw) {
      const size_t r = idx / co;  // (h*fw + w)*ci + c
      const int c = (int)(r % ci);
      const size_t hw = r / ci;
      o = ((size_t)m * ci + c) * ((size_t)fh * fw) + hw;
    }
    out[o] = v;
    nz = v != 0;
  }
  if (__syncthreads_or(nz) && threadIdx.x == 0) atomicOr(any_nonzero, 1u);
}
// :1771-1786: out[p][m] += bias[m] << k (ScaleUp in the SCI_OT build, then SecretAdd = wrapping add)
__global__ __launch_bounds__(256) void fused_bn_bias_kernel(u64 *__restrict__ out, const u64 *__restrict__ bias,
                                                            size_t total, int co, int k_scale) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) out[idx] += bias[idx % co] << k_scale;
}
// Layout changes of the layer adapters (SCI/src/library_fixed_uniform_cheetah.cpp:236-262, :284-299): the networks
// keep activations as [H][W][C] and filters as [FH][FW][CI][CO]; HomConv2DSS wants [C][H][W] and [CO][CI][FH][FW],
// entries reduced into the ring (getRingElt: & (2^l - 1)).  One thread per OUTPUT element (coalesced writes; the
// strided reads of a few MiB are served by the L2).
__global__ __launch_bounds__(256) void nhwc_to_chw_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                          size_t hw, size_t c, u64 mask) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= hw * c) return;
  const size_t ch = idx / hw, p = idx % hw;
  out[idx] = in[p * c + ch] & mask;
}
__global__ __launch_bounds__(256) void chw_to_nhwc_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                          size_t hw, size_t c) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= hw * c) return;
  const size_t p = idx / c, ch = idx % c;
  out[idx] = in[ch * hw + p];
}
__global__ __launch_bounds__(256) void hwio_to_oihw_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                           size_t hw, size_t ci, size_t co, u64 mask) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= hw * ci * co) return;
  const size_t p = idx % hw, r = idx / hw, c = r % ci, m = r / ci;
  out[idx] = in[(p * ci + c) * co + m] & mask;
}
}  // namespace mmh

using namespace mmh;
#define CHECK_RC(x)      \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)
// every array 16-byte aligned, the bit array 2-byte (null ones do not count): two elements per lane
static bool aligned_pairs(std::initializer_list<const void *> ptrs, const void *bit_array = nullptr) {
  uintptr_t bits = 0;
  for (const void *p : ptrs) bits |= (uintptr_t)p;
  return (bits & 15) == 0 && ((uintptr_t)bit_array & 1) == 0;
}
#define LAUNCH_PAIRS(kern2, kern1, n, s, ptrs, ...)                                                                \
  do {                                                                                                             \
    if ((n) > 0) {                                                                                                 \
      if (aligned_pairs ptrs)                                                                                      \
        hipLaunchKernelGGL((kern2), dim3(nblocks(((n) + 1) / 2)), dim3(256), 0, (hipStream_t)(s), __VA_ARGS__);    \
      else                                                                                                         \
        hipLaunchKernelGGL((kern1), dim3(nblocks(n)), dim3(256), 0, (hipStream_t)(s), __VA_ARGS__);                \
      MMH_HIP(hipGetLastError());                                                                                  \
    }                                                                                                              \
  } while (0)
#define LAUNCH1D(kern, n, s, ...)                                                          \
  do {                                                                                     \
    if ((n) > 0) {                                                                         \
      hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)(s), __VA_ARGS__); \
      MMH_HIP(hipGetLastError());                                                          \
    }                                                                                      \
  } while (0)

extern "C" int mmh_r2f_pre(int party, const uint64_t *d_in, size_t n, int bw, uint64_t p, uint64_t *d_tmp,
                           uint8_t *d_msb, uint64_t *d_corr, void *stream) {
  CHECK_RC(check_common(party, n, d_in, d_tmp, d_msb));
  MMH_REQUIRE(bw >= 2 && bw <= 64 && p >= 2, MMH_ERR_INVALID_ARG);
  const u64 mask = bw >= 64 ? ~u64(0) : (u64(1) << bw) - 1;
  LAUNCH_PAIRS((pre_kernel<0, 2>), (pre_kernel<0, 1>), n, stream, ({d_in, d_tmp, d_corr}, d_msb), party, d_in, n, bw, p,
               mask % p + 1, 0, d_tmp, d_msb, d_corr);
  return MMH_OK;
}
extern "C" int mmh_r2f_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y,
                            size_t n, int bw, uint64_t p, uint64_t *d_out, void *stream) {
  CHECK_RC(check_common(party, n, d_tmp, d_msb, d_y));
  MMH_REQUIRE(n == 0 || d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 2 && bw <= 64 && p >= 2, MMH_ERR_INVALID_ARG);
  const u64 mask = bw >= 64 ? ~u64(0) : (u64(1) << bw) - 1;
  LAUNCH_PAIRS(r2f_post_kernel<2>, r2f_post_kernel<1>, n, stream, ({d_tmp, d_y, d_out}, d_msb), party, d_tmp, d_msb,
               d_y, n, p, mask % p + 1, (u64(1) << (bw - 2)) % p, d_out);
  return MMH_OK;
}
extern "C" int mmh_f2r_pre(int party, const uint64_t *d_in, size_t n, uint64_t p, int shift, uint64_t *d_tmp,
                           uint8_t *d_msb, uint64_t *d_corr, void *stream) {
  CHECK_RC(check_common(party, n, d_in, d_tmp, d_msb));
  MMH_REQUIRE(p >= 4 && shift >= 0 && shift < 64, MMH_ERR_INVALID_ARG);
  LAUNCH_PAIRS((pre_kernel<1, 2>), (pre_kernel<1, 1>), n, stream, ({d_in, d_tmp, d_corr}, d_msb), party, d_in, n, 0, p,
               p >> shift, 0, d_tmp, d_msb, d_corr);
  return MMH_OK;
}
extern "C" int mmh_f2r_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y,
                            size_t n, uint64_t p, int bw, uint64_t *d_out, void *stream) {
  CHECK_RC(check_common(party, n, d_tmp, d_msb, d_y));
  MMH_REQUIRE(n == 0 || d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 1 && bw <= 64 && p >= 4, MMH_ERR_INVALID_ARG);
  LAUNCH_PAIRS(f2r_post_kernel<2>, f2r_post_kernel<1>, n, stream, ({d_tmp, d_y, d_out}, d_msb), party, d_tmp, d_msb,
               d_y, n, p, bw, 0, 0, d_out);
  return MMH_OK;
}
extern "C" int mmh_f2r_trunc_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y,
                                  size_t n, uint64_t p, int bw, int shift, uint64_t *d_out, void *stream) {
  CHECK_RC(check_common(party, n, d_tmp, d_msb, d_y));
  MMH_REQUIRE(n == 0 || d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 1 && bw <= 64 && p >= 4 && shift >= 0 && shift < 64, MMH_ERR_INVALID_ARG);
  LAUNCH_PAIRS(f2r_post_kernel<2>, f2r_post_kernel<1>, n, stream, ({d_tmp, d_y, d_out}, d_msb), party, d_tmp, d_msb,
               d_y, n, p, bw, shift, 1, d_out);
  return MMH_OK;
}
extern "C" int mmh_r2r_pre(int party, const uint64_t *d_in, size_t n, int bw_a, uint64_t *d_tmp, uint8_t *d_msb,
                           uint64_t *d_corr, void *stream) {
  CHECK_RC(check_common(party, n, d_in, d_tmp, d_msb));
  MMH_REQUIRE(bw_a >= 2 && bw_a <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH_PAIRS((pre_kernel<0, 2>), (pre_kernel<0, 1>), n, stream, ({d_in, d_tmp, d_corr}, d_msb), party, d_in, n, bw_a,
               2, 0, 1, d_tmp, d_msb, d_corr);
  return MMH_OK;
}
extern "C" int mmh_r2r_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y,
                            size_t n, int bw_a, int bw_b, uint64_t *d_out, void *stream) {
  CHECK_RC(check_common(party, n, d_tmp, d_msb, d_y));
  MMH_REQUIRE(n == 0 || d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw_a >= 2 && bw_a < bw_b && bw_b <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(r2r_post_kernel<false>, n, stream, party, d_tmp, d_msb, d_y, n, bw_a, bw_b, d_out);
  return MMH_OK;
}
extern "C" int mmh_r2r128_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y,
                               size_t n, int bw_a, int bw_b, uint64_t *d_out, void *stream) {
  CHECK_RC(check_common(party, n, d_tmp, d_msb, d_y));
  MMH_REQUIRE(n == 0 || d_out, MMH_ERR_NULL_POINTER);
  // truncation.cpp:392: the extension is at most 64 bits wide
  MMH_REQUIRE(bw_a >= 2 && bw_a <= 64 && bw_a < bw_b && bw_b <= 128 && bw_b - bw_a <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(r2r_post_kernel<true>, n, stream, party, d_tmp, d_msb, d_y, n, bw_a, bw_b, d_out);
  return MMH_OK;
}
extern "C" int mmh_trunc_pre(int party, const uint64_t *d_in, size_t n, int bw, int signed_arith, int heuristic,
                             uint64_t *d_adj, uint64_t *d_cmp, void *stream) {
  MMH_REQUIRE(party == 1 || party == 2, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(n == 0 || (d_in && d_adj), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= (heuristic ? 4 : 2) && bw <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH_PAIRS(trunc_pre_kernel<2>, trunc_pre_kernel<1>, n, stream, ({d_in, d_adj, d_cmp}), party, d_in, n, bw,
               signed_arith, heuristic, d_adj, d_cmp);
  return MMH_OK;
}
extern "C" int mmh_trunc_post(int party, const uint64_t *d_adj, const uint64_t *d_aw, size_t n, int shift, int bw,
                              int signed_arith, int heuristic, uint64_t *d_out, void *stream) {
  MMH_REQUIRE(party == 1 || party == 2, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(n == 0 || (d_adj && d_aw && d_out), MMH_ERR_NULL_POINTER);
  // the reference's asserts (truncation-cheetah.cpp:12-13); the heuristic's offset 2^(bw-3-shift) must exist
  MMH_REQUIRE(bw >= 2 && bw <= 64 && shift >= 1 && bw - shift > 0 && (!heuristic || shift <= bw - 3),
              MMH_ERR_INVALID_ARG);
  LAUNCH_PAIRS(trunc_post_kernel<2>, trunc_post_kernel<1>, n, stream, ({d_adj, d_aw, d_out}), party, d_adj, d_aw, n,
               shift, bw, signed_arith, heuristic, d_out);
  return MMH_OK;
}
extern "C" int mmh_trunc_reduce_pre(const uint64_t *d_in, size_t n, int shift, uint64_t *d_lower, void *stream) {
  MMH_REQUIRE(n == 0 || (d_in && d_lower), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(shift >= 1 && shift <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(trunc_reduce_kernel, n, stream, d_in, (const u64 *)nullptr, n, shift, 64, d_lower);
  return MMH_OK;
}
extern "C" int mmh_trunc_reduce_post(const uint64_t *d_in, const uint64_t *d_aw, size_t n, int shift, int bw,
                                     uint64_t *d_out, void *stream) {
  MMH_REQUIRE(n == 0 || (d_in && d_aw && d_out), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 2 && bw <= 64 && shift >= 1 && bw - shift > 0, MMH_ERR_INVALID_ARG);
  LAUNCH1D(trunc_reduce_kernel, n, stream, d_in, d_aw, n, shift, bw, d_out);
  return MMH_OK;
}

extern "C" int mmh_relu_pre(int party, const uint64_t *d_share, size_t n, int l, uint8_t *d_msb, uint64_t *d_cmp,
                            void *stream) {
  MMH_REQUIRE(party == 1 || party == 2, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(n == 0 || (d_share && d_msb && d_cmp), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(l >= 2 && l <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(relu_pre_kernel, n, stream, party, d_share, n, l, d_msb, d_cmp);
  return MMH_OK;
}
extern "C" int mmh_relu_post(int party, const uint8_t *d_msb, const uint8_t *d_wrap, size_t n, uint8_t *d_drelu,
                             uint8_t *d_sel, void *stream) {
  MMH_REQUIRE(party == 1 || party == 2, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(n == 0 || (d_msb && d_wrap && d_drelu && d_sel), MMH_ERR_NULL_POINTER);
  LAUNCH1D(relu_post_kernel, n, stream, party, d_msb, d_wrap, n, d_drelu, d_sel);
  return MMH_OK;
}
extern "C" int mmh_mux_pre(const uint8_t *d_sel, const uint64_t *d_x, size_t n, int bw_y, uint64_t *d_corr,
                           void *stream) {
  MMH_REQUIRE(n == 0 || (d_sel && d_x && d_corr), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw_y >= 1 && bw_y <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(mux_kernel, n, stream, d_sel, d_x, (const u64 *)nullptr, (const u64 *)nullptr, n, bw_y, d_corr);
  return MMH_OK;
}
extern "C" int mmh_mux_post(const uint8_t *d_sel, const uint64_t *d_x, const uint64_t *d_data_r,
                            const uint64_t *d_data_s, size_t n, int bw_y, uint64_t *d_y, void *stream) {
  MMH_REQUIRE(n == 0 || (d_sel && d_x && d_data_r && d_data_s && d_y), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw_y >= 1 && bw_y <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(mux_kernel, n, stream, d_sel, d_x, d_data_r, d_data_s, n, bw_y, d_y);
  return MMH_OK;
}
extern "C" int mmh_b2a_pre(const uint8_t *d_x, size_t n, int bw_y, uint64_t *d_corr, void *stream) {
  MMH_REQUIRE(n == 0 || (d_x && d_corr), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw_y >= 1 && bw_y <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(b2a_kernel, n, stream, 1, d_x, (const u64 *)nullptr, n, bw_y, 0, d_corr);
  return MMH_OK;
}
extern "C" int mmh_b2a_post(int party, const uint8_t *d_x, const uint64_t *d_y, size_t n, int bw_y, uint64_t *d_out,
                            void *stream) {
  MMH_REQUIRE(party == 1 || party == 2, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(n == 0 || (d_x && d_out && (d_y || bw_y == 1)), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw_y >= 1 && bw_y <= 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(b2a_kernel, n, stream, party, d_x, d_y, n, bw_y, 1, d_out);
  return MMH_OK;
}

extern "C" int mmh_bole_cross_term(uint64_t *d_out, const uint64_t *d_x, const uint64_t *d_y, size_t n,
                                   uint64_t p, void *stream) {
  MMH_REQUIRE(n == 0 || (d_out && d_x && d_y), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(p >= 2 && p < (u64(1) << 62), MMH_ERR_INVALID_ARG);
  const unsigned __int128 ratio = p & (p - 1) ? ~(unsigned __int128)0 / p : (~(unsigned __int128)0 / p) + 1;
  LAUNCH1D(cross_term_kernel, n, stream, d_out, d_x, d_y, n, p, (u64)(ratio >> 64), (u64)ratio);
  return MMH_OK;
}

extern "C" int mmh_fused_bn_filters(const uint64_t *d_filters, const uint64_t *d_scales, int fh, int fw, int ci, int co,
                                    int bitlength, int k_scale, uint64_t prime_mod, int oihw, uint64_t *d_out,
                                    uint32_t *d_any_nonzero, void *stream) {
  MMH_REQUIRE(d_filters && d_scales && d_out && d_any_nonzero, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(fh > 0 && fw > 0 && ci > 0 && co > 0, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(bitlength >= 2 && bitlength <= 62 && k_scale >= 0 && k_scale < bitlength, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(prime_mod >= 2 && prime_mod < (uint64_t(1) << 62), MMH_ERR_INVALID_ARG);  // neg_mod works in int64_t
  MMH_REQUIRE(d_out != d_filters || !oihw, MMH_ERR_INVALID_ARG);                       // the transposing form is out of place
  const size_t total = (size_t)fh * fw * ci * co;
  MMH_HIP(hipMemsetAsync(d_any_nonzero, 0, 4, (hipStream_t)stream));
  LAUNCH1D(fused_bn_filters_kernel, total, stream, d_filters, d_scales, total, fh, fw, ci, co, bitlength,
           (double)(uint64_t(1) << k_scale), (long long)prime_mod, oihw, d_out, d_any_nonzero);
  return MMH_OK;
}

extern "C" int mmh_fused_bn_bias(uint64_t *d_out_nhwc, const uint64_t *d_bias, size_t pixels, int co, int k_scale,
                                 void *stream) {
  MMH_REQUIRE(d_out_nhwc && d_bias, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(co > 0 && k_scale >= 0 && k_scale < 64, MMH_ERR_INVALID_ARG);
  LAUNCH1D(fused_bn_bias_kernel, pixels * (size_t)co, stream, d_out_nhwc, d_bias, pixels * (size_t)co, co, k_scale);
  return MMH_OK;
}

extern "C" int mmh_repack_nhwc_to_chw(const uint64_t *d_in, size_t h, size_t w, size_t c, uint64_t ring_mask,
                                      uint64_t *d_out, void *stream) {
  MMH_REQUIRE(d_in && d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(d_in != d_out, MMH_ERR_INVALID_ARG);
  LAUNCH1D(nhwc_to_chw_kernel, h * w * c, stream, d_in, d_out, h * w, c, ring_mask);
  return MMH_OK;
}
extern "C" int mmh_repack_chw_to_nhwc(const uint64_t *d_in, size_t c, size_t h, size_t w, uint64_t *d_out,
                                      void *stream) {
  MMH_REQUIRE(d_in && d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(d_in != d_out, MMH_ERR_INVALID_ARG);
  LAUNCH1D(chw_to_nhwc_kernel, h * w * c, stream, d_in, d_out, h * w, c);
  return MMH_OK;
}
extern "C" int mmh_repack_hwio_to_oihw(const uint64_t *d_in, size_t fh, size_t fw, size_t ci, size_t co,
                                       uint64_t ring_mask, uint64_t *d_out, void *stream) {
  MMH_REQUIRE(d_in && d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(d_in != d_out, MMH_ERR_INVALID_ARG);
  LAUNCH1D(hwio_to_oihw_kernel, fh * fw * ci * co, stream, d_in, d_out, fh * fw, ci, co, ring_mask);
  return MMH_OK;
}
