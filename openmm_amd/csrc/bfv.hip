// BFV arithmetic on batches: plain scaling, plaintext multiply-accumulate, mod-switch,
// encryption with injected randomness, decryption (BEHZ scale-and-round), truncation,
// BatchEncoder.  All kernels are element-wise over coefficients (HBM-bound streaming) and
// use the batched NTT of ntt.hip for the transforms.  Reference call sites: include/mm_hip.h.
#include "common.hpp"
#include "bfv_device.hpp"

#include <mutex>

namespace mmh {
int ntt_dispatch(const mmh_ctx *ctx, bool inverse, u64 *d_polys, size_t n_polys, int limbs,
                 int mod_base, hipStream_t s);
int ntt_signed_dispatch(const mmh_ctx *ctx, const int64_t *d_src, u64 *d_out, size_t n_polys, int limbs,
                        hipStream_t s);
int ntt_inv_product_reshare(const mmh_ctx *ctx, const u64 *d_u_ntt, const u64 *d_key, size_t n, const u64 *d_acc,
                            const int64_t *d_e1, const u64 *d_mask, u64 *d_out, const uint8_t *d_keep,
                            size_t keep_period, const int *d_share_dst, size_t share_plane, u64 *d_share_out,
                            hipStream_t s, bool acc_switched);
int ntt_inv_product_dispatch(const mmh_ctx *ctx, const u64 *d_u_ntt, const u64 *d_key, u64 *d_out, size_t n,
                             int limbs, hipStream_t s);
int ntt_lift_dispatch(const mmh_ctx *ctx, const u64 *d_src, u64 *d_out, size_t n_polys, int limbs,
                      hipStream_t s);
int reshare_online_dispatch(const mmh_ctx *ctx, const u64 *d_zero, size_t n, const u64 *d_acc, const int64_t *d_e1,
                            const u64 *d_mask, u64 *d_out, const uint8_t *d_keep, size_t keep_period,
                            const int *d_share_dst, size_t share_plane, u64 *d_share_out, hipStream_t s,
                            bool acc_switched);

// call-scoped scratch buffer (stream-ordered reuse, see scratch_alloc)
struct StreamBuf {
  void *p = nullptr;
  hipStream_t s;
  explicit StreamBuf(hipStream_t st) : s(st) {}
  hipError_t alloc(size_t bytes) { return scratch_alloc(&p, bytes ? bytes : 1, s); }
  ~StreamBuf() {
    if (p) (void)scratch_free(p, s);
  }
  template <class T>
  T *as() { return reinterpret_cast<T *>(p); }
};

// out = in (+|-) scaled plaintext, out-of-place: every coefficient of both polynomials is written, so
// no separate copy of the ciphertexts is needed
__global__ __launch_bounds__(256) void add_plain_copy_kernel(const u64 *__restrict__ in, const u64 *__restrict__ pts,
                                                             u64 *__restrict__ out, size_t total, int logN, int sub,
                                                             BfvConsts b, int lv, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const LevelConsts &c = b.lv[lv];
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const u64 m = pts[idx];
  const u64 fix = m ? scale_fix(m, b, c) : 0;
  for (int l = 0; l < c.limbs; ++l) {
    const DevMod &q = mods[l];
    const size_t o0 = (((k * 2) * c.limbs + l) << logN) + i, o1 = (((k * 2 + 1) * c.limbs + l) << logN) + i;
    u64 v = in[o0];
    if (m) {
      const u64 sc = scaled_plain(m, fix, c.delta_mod_q[l], q);
      v = sub ? sub_mod(v, sc, q.q) : add_mod(v, sc, q.q);
    }
    out[o0] = v;
    out[o1] = in[o1];
  }
}

__global__ __launch_bounds__(256) void add_plain_kernel(u64 *__restrict__ cts, const u64 *__restrict__ pts,
                                                        size_t total, int logN, int sub, BfvConsts b, int lv,
                                                        const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const LevelConsts &c = b.lv[lv];
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const u64 m = pts[idx];
  if (m == 0) return;
  const u64 fix = scale_fix(m, b, c);
  for (int l = 0; l < c.limbs; ++l) {
    const DevMod &q = mods[l];
    const u64 s = scaled_plain(m, fix, c.delta_mod_q[l], q);
    u64 *d = cts + (((k * 2) * c.limbs + l) << logN) + i;
    *d = sub ? sub_mod(*d, s, q.q) : add_mod(*d, s, q.q);
  }
}

// out[o1][o2] = sum_k ct[...] (.) pt[...], TM outputs along o1 per thread so that each ct
// coefficient is loaded once per TM plaintext coefficients.  128-bit lazy accumulation,
// reduced every 8 terms.
template <int TM>
__global__ __launch_bounds__(256) void dyadic_mac_kernel(const u64 *__restrict__ cts, const u64 *__restrict__ pts,
                                                         u64 *__restrict__ out, int logN, int limbs, unsigned n_o1,
                                                         unsigned n_o2, unsigned K, size_t ct_sk, size_t ct_s2,
                                                         size_t pt_s1, size_t pt_sk, size_t pt_s2,
                                                         const DevMod *__restrict__ mods) {
  const unsigned blocks_per_limb = (1u << logN) / 256;
  const int l = blockIdx.x / blocks_per_limb;
  const size_t i = (size_t)(blockIdx.x % blocks_per_limb) * 256 + threadIdx.x;
  const unsigned o2 = blockIdx.y, o1_base = blockIdx.z * TM;
  const DevMod q = mods[l];
  const size_t poly = size_t(1) << logN;
  u64 lo[TM][2], hi[TM][2];
#pragma unroll
  for (int j = 0; j < TM; ++j) lo[j][0] = lo[j][1] = hi[j][0] = hi[j][1] = 0;
  // operands of term k+1 are fetched while term k is multiplied: the loop is a pure stream over
  // the plaintext operand and needs the next loads in flight to cover the HBM latency
  const u64 *ct_ptr = cts + ((o2 * ct_s2 * 2) * limbs + l) * poly + i;
  const size_t ct_step = ct_sk * 2 * limbs * poly, c1_off = (size_t)limbs * poly;
  const u64 *pt_ptr[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const unsigned o1 = o1_base + j < n_o1 ? o1_base + j : n_o1 - 1;  // tail lanes repeat the last filter
    pt_ptr[j] = pts + ((o1 * pt_s1 + o2 * pt_s2) * limbs + l) * poly + i;
  }
  const size_t pt_step = pt_sk * limbs * poly;
  u64 c0n = ct_ptr[0], c1n = ct_ptr[c1_off], wn[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) wn[j] = pt_ptr[j][0];
  for (unsigned k = 0; k < K; ++k) {
    const u64 c0 = c0n, c1 = c1n;
    u64 w[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) w[j] = wn[j];
    if (k + 1 < K) {
      ct_ptr += ct_step;
      c0n = ct_ptr[0];
      c1n = ct_ptr[c1_off];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        pt_ptr[j] += pt_step;
        wn[j] = pt_ptr[j][0];
      }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      u64 p = c0 * w[j];
      lo[j][0] += p;
      hi[j][0] += mulhi64(c0, w[j]) + (lo[j][0] < p);
      p = c1 * w[j];
      lo[j][1] += p;
      hi[j][1] += mulhi64(c1, w[j]) + (lo[j][1] < p);
    }
    // Barrett's 64-bit quotient needs the sum below q*2^64: 8 products of < q^2 with q < 2^61
    if ((k & 7) == 7) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          lo[j][c] = reduce128_wide(hi[j][c], lo[j][c], q);
          hi[j][c] = 0;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < TM; ++j)
    if (o1_base + j < n_o1) {
      const size_t o = (size_t)(o1_base + j) * n_o2 + o2;
      out[((o * 2 + 0) * limbs + l) * poly + i] = reduce128_wide(hi[j][0], lo[j][0], q);
      out[((o * 2 + 1) * limbs + l) * poly + i] = reduce128_wide(hi[j][1], lo[j][1], q);
    }
}

// c = a (+|-) b element-wise over [n_polys][limbs][N]; b may be signed noise broadcast over limbs
__global__ __launch_bounds__(256) void poly_add_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, size_t total,
                                                       int logN, int limbs, int sub, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const u64 q = mods[(idx >> logN) % limbs].q;
  a[idx] = sub ? sub_mod(a[idx], b[idx], q) : add_mod(a[idx], b[idx], q);
}

// ct.c0 -= r : cts [n][2][limbs][N], r [n][limbs][N]
__global__ __launch_bounds__(256) void sub_c0_kernel(u64 *__restrict__ cts, const u64 *__restrict__ r, size_t total,
                                                     int logN, int limbs, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t i = idx & ((size_t(1) << logN) - 1), pl = idx >> logN;  // pl = k*limbs + l
  const size_t k = pl / limbs, l = pl % limbs;
  u64 *d = cts + (((k * 2) * limbs + l) << logN) + i;
  *d = sub_mod(*d, r[idx], mods[l].q);
}

// flood_ciphertext, exact build (hom_conv2d_ss.cc:358-380): c0[l] += (random & bitmask) mod q_l
__global__ __launch_bounds__(256) void flood_add_kernel(u64 *__restrict__ cts, const u64 *__restrict__ random,
                                                        size_t total, int logN, int limbs, u64 bitmask,
                                                        const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][N]
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const u64 r = random[idx] & bitmask;
  for (int l = 0; l < limbs; ++l) {
    u64 *d = cts + (((k * 2) * limbs + l) << logN) + i;
    *d = add_mod(*d, reduce64(r, mods[l]), mods[l].q);
  }
}

// v mod t element-wise
__global__ __launch_bounds__(256) void reduce_t_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t total,
                                                       ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) out[idx] = reduce128(0, in[idx], t);
}

__global__ __launch_bounds__(256) void mod_switch_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                         size_t total, int logN, u64 inv_qlast,
                                                         const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n*2][N]
  if (idx >= total) return;
  const size_t i = idx & ((size_t(1) << logN) - 1), p = idx >> logN;
  const DevMod &q0 = mods[0];
  const u64 q1 = mods[1].q, half = q1 >> 1;
  u64 last = in[((p * 2 + 1) << logN) + i] + half;
  last = csub(last, q1);
  const u64 tmp = sub_mod(reduce64(last, q0), reduce64(half, q0), q0.q);
  const u64 x = sub_mod(in[((p * 2) << logN) + i], tmp, q0.q);
  out[idx] = mul_mod(x, inv_qlast, q0);
}

// signed small polynomial [n][N] -> RNS [n][limbs][N]
__global__ __launch_bounds__(256) void signed_to_rns_kernel(const int64_t *__restrict__ v, u64 *__restrict__ out,
                                                            size_t total, int logN, int limbs,
                                                            const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const int64_t x = v[idx];
  for (int l = 0; l < limbs; ++l) out[((k * limbs + l) << logN) + i] = signed_to_mod(x, mods[l].q);
}

// out[k][j][l] = u_ntt[k][l] (.) pk[j][l]
__global__ __launch_bounds__(256) void mul_pk_kernel(const u64 *__restrict__ u_ntt, const u64 *__restrict__ pk,
                                                     u64 *__restrict__ out, size_t total, int logN, int limbs,
                                                     const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][limbs][N]
  if (idx >= total) return;
  const size_t i = idx & ((size_t(1) << logN) - 1), pl = idx >> logN;
  const size_t k = pl / limbs, l = pl % limbs;
  const DevMod &q = mods[l];
  const u64 u = u_ntt[idx];
  for (int j = 0; j < 2; ++j)
    out[(((k * 2 + j) * limbs + l) << logN) + i] = mul_mod(u, pk[((j * limbs + l) << logN) + i], q);
}

// ct[k][j][l][i] += e_j[k][i]
__global__ __launch_bounds__(256) void add_noise_kernel(u64 *__restrict__ cts, const int64_t *__restrict__ e0,
                                                        const int64_t *__restrict__ e1, size_t total, int logN,
                                                        int limbs, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][N]
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  for (int j = 0; j < 2; ++j) {
    const int64_t *e = j ? e1 : e0;
    if (!e) continue;
    const int64_t x = e[idx];
    for (int l = 0; l < limbs; ++l) {
      u64 *d = cts + (((k * 2 + j) * limbs + l) << logN) + i;
      *d = add_mod(*d, signed_to_mod(x, mods[l].q), mods[l].q);
    }
  }
}

// tmp[k][l] (.)= key[l]   (key broadcast over k)
__global__ __launch_bounds__(256) void mul_key_kernel(u64 *__restrict__ tmp, const u64 *__restrict__ key,
                                                      size_t total, int logN, int limbs,
                                                      const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t i = idx & ((size_t(1) << logN) - 1), l = (idx >> logN) % limbs;
  tmp[idx] = mul_mod(tmp[idx], key[(l << logN) + i], mods[l]);
}

// gather polynomial `which` of every ciphertext: dst[k][l] = cts[k][which][l]
__global__ __launch_bounds__(256) void gather_poly_kernel(const u64 *__restrict__ cts, u64 *__restrict__ dst,
                                                          size_t total, int logN, int limbs, int which) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t i = idx & ((size_t(1) << logN) - 1), pl = idx >> logN;
  const size_t k = pl / limbs, l = pl % limbs;
  dst[idx] = cts[(((k * 2 + which) * limbs + l) << logN) + i];
}

// symmetric encryption tail: c0 = -(as + e) + scaled(m), c1 = a
__global__ __launch_bounds__(256) void sym_finish_kernel(const u64 *__restrict__ as, const u64 *__restrict__ a,
                                                         const int64_t *__restrict__ e, const u64 *__restrict__ pts,
                                                         u64 *__restrict__ out, size_t total, int logN, BfvConsts b,
                                                         int lv, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][N]
  if (idx >= total) return;
  const LevelConsts &c = b.lv[lv];
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const int64_t x = e[idx];
  const u64 m = pts ? pts[idx] : 0;
  const u64 fix = m ? scale_fix(m, b, c) : 0;
  for (int l = 0; l < c.limbs; ++l) {
    const DevMod &q = mods[l];
    const size_t src = ((k * c.limbs + l) << logN) + i;
    u64 v = add_mod(as[src], signed_to_mod(x, q.q), q.q);
    v = v ? q.q - v : 0;
    if (m) v = add_mod(v, scaled_plain(m, fix, c.delta_mod_q[l], q), q.q);
    out[(((k * 2) * c.limbs + l) << logN) + i] = v;
    out[(((k * 2 + 1) * c.limbs + l) << logN) + i] = a[src];
  }
}

// pk = (-(a s + e), a), all NTT form; e_ntt [L][N]
__global__ __launch_bounds__(256) void pk_kernel(const u64 *__restrict__ sk, const u64 *__restrict__ a,
                                                 const u64 *__restrict__ e_ntt, u64 *__restrict__ pk, size_t total,
                                                 int logN, int limbs, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [L][N]
  if (idx >= total) return;
  const size_t l = idx >> logN;
  const DevMod &q = mods[l];
  const u64 v = add_mod(mul_mod(a[idx], sk[idx], q), e_ntt[idx], q.q);
  pk[idx] = v ? q.q - v : 0;
  pk[idx + ((size_t)limbs << logN)] = a[idx];
}

// decrypt tail: (tmp[k][l] + c0[k][l]) -> scale and round -> pts[k]
__global__ __launch_bounds__(256) void decrypt_finish_kernel(const u64 *__restrict__ tmp, const u64 *__restrict__ cts,
                                                             u64 *__restrict__ pts, size_t total, int logN,
                                                             BfvConsts b, int lv, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][N]
  if (idx >= total) return;
  const LevelConsts &c = b.lv[lv];
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  u64 x[2];
  for (int l = 0; l < c.limbs; ++l)
    x[l] = add_mod(tmp[((k * c.limbs + l) << logN) + i], cts[(((k * 2) * c.limbs + l) << logN) + i], mods[l].q);
  pts[idx] = behz_round(x, b, c, mods);
}

__global__ __launch_bounds__(256) void scale_round_kernel(const u64 *__restrict__ polys, const u32 *__restrict__ targets,
                                                          u64 *__restrict__ out, size_t n, size_t n_targets,
                                                          int logN, BfvConsts b, int lv,
                                                          const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][n_targets]
  if (idx >= n * n_targets) return;
  const LevelConsts &c = b.lv[lv];
  const size_t k = idx / n_targets, j = idx % n_targets;
  const size_t i = targets ? targets[j] : j;
  u64 x[2];
  for (int l = 0; l < c.limbs; ++l) x[l] = polys[((k * c.limbs + l) << logN) + i];
  out[idx] = behz_round(x, b, c, mods);
}

__global__ __launch_bounds__(256) void truncate_kernel(u64 *__restrict__ cts, size_t total, int logN, u64 mask0,
                                                       u64 mask1, const uint8_t *__restrict__ keep,
                                                       size_t keep_period) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n][N]
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const bool kept = !keep || keep[((k % keep_period) << logN) + i];
  u64 *c0 = cts + ((k * 2) << logN) + i, *c1 = c0 + (size_t(1) << logN);
  *c0 = kept ? (*c0 & mask0) : 0;  // remove_unused_coeffs walks ct.data(0) only: c1 keeps every coefficient
  *c1 &= mask1;
}

__global__ __launch_bounds__(256) void slot_scatter_kernel(const u64 *__restrict__ vals, u64 *__restrict__ pts,
                                                           size_t total, int logN, const u32 *__restrict__ map,
                                                           ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  pts[(k << logN) + map[i]] = reduce128(0, vals[idx], t);
}
__global__ __launch_bounds__(256) void slot_gather_kernel(const u64 *__restrict__ tmp, u64 *__restrict__ vals,
                                                          size_t total, int logN, const u32 *__restrict__ map) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  vals[idx] = tmp[(k << logN) + map[i]];
}

static inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

static int check_bfv(const mmh_ctx *ctx, int lv) {
  MMH_REQUIRE(ctx, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->t >= 2, MMH_ERR_CONFIG);
  MMH_REQUIRE(lv == 0 || lv == 1, MMH_ERR_INVALID_ARG);
  return MMH_OK;
}
static inline int limbs_of(const mmh_ctx *ctx, int lv) { return lv == 0 ? (int)ctx->L : 1; }
// seal::util polynomial arithmetic modulo one word-sized modulus, element-wise (polyarithsmallmod.h as the
// reference's loops use it): op 0 a + b, 1 a - b, 2 a * b (dyadic), 3 a * scalar, 4 a mod m.  Operands of the
// first four are reduced (below m); op 4 takes any 64-bit value.
__global__ __launch_bounds__(256) void poly_op_kernel(int op, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                      u64 scalar, size_t n, ModC m, u64 *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 x = a[i];
  u64 r;
  switch (op) {
    case 0: r = add_mod(x, b[i], m.q); break;
    case 1: r = sub_mod(x, b[i], m.q); break;
    case 2: r = barrett128(mulhi64(x, b[i]), x * b[i], m.q, m.barrett_hi, m.barrett_lo); break;
    case 3: r = barrett128(mulhi64(x, scalar), x * scalar, m.q, m.barrett_hi, m.barrett_lo); break;
    default: r = barrett128(0, x, m.q, m.barrett_hi, m.barrett_lo); break;
  }
  out[i] = r;
}
}  // namespace mmh

using namespace mmh;
#define CHECK_RC(x)        \
  do {                     \
    int rc_ = (x);         \
    if (rc_) return rc_;   \
  } while (0)
#define LAUNCH(kern, total, s, ...)                                                  \
  do {                                                                               \
    if ((total) > 0) {                                                               \
      hipLaunchKernelGGL(kern, dim3(blocks_for(total)), dim3(256), 0, s, __VA_ARGS__); \
      MMH_HIP(hipGetLastError());                                                    \
    }                                                                                \
  } while (0)

extern "C" int mmh_add_plain(const mmh_ctx *ctx, int lv, const uint64_t *d_cts, const uint64_t *d_pts,
                             size_t n, int sub, uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_cts && d_pts && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N;
  const int limbs = limbs_of(ctx, lv);
  (void)limbs;
  if (d_out != d_cts) {
    LAUNCH(add_plain_copy_kernel, n * N, s, d_cts, d_pts, d_out, n * N, ctx->logN, sub, ctx->bfv, lv, ctx->d_mods);
    return MMH_OK;
  }
  LAUNCH(add_plain_kernel, n * N, s, d_out, d_pts, n * N, ctx->logN, sub, ctx->bfv, lv, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_plain_lift_ntt(const mmh_ctx *ctx, int lv, const uint64_t *d_pts, size_t n,
                                  uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_pts && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const int limbs = limbs_of(ctx, lv);
  return ntt_lift_dispatch(ctx, d_pts, d_out, n, limbs, s);  // lift fused into the transform's load
}

extern "C" int mmh_dyadic_mac(const mmh_ctx *ctx, int lv, const uint64_t *d_cts_ntt, size_t n_cts,
                              const uint64_t *d_pts_ntt, size_t n_pts, size_t n_o1, size_t n_o2, size_t K,
                              size_t ct_sk, size_t ct_s2, size_t pt_s1, size_t pt_sk, size_t pt_s2,
                              uint64_t *d_out_ntt, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n_o1 == 0 || n_o2 == 0) return MMH_OK;
  MMH_REQUIRE(d_cts_ntt && d_pts_ntt && d_out_ntt, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(K >= 1, MMH_ERR_INVALID_ARG);
  // every index the kernel forms must lie inside the operand arrays
  MMH_REQUIRE((K - 1) * ct_sk + (n_o2 - 1) * ct_s2 < n_cts, MMH_ERR_OUT_BOUND);
  MMH_REQUIRE((n_o1 - 1) * pt_s1 + (K - 1) * pt_sk + (n_o2 - 1) * pt_s2 < n_pts, MMH_ERR_OUT_BOUND);
  MMH_REQUIRE(n_o2 <= 65535 && (n_o1 + 3) / 4 <= 65535, MMH_ERR_OUT_BOUND);
  const int limbs = limbs_of(ctx, lv);
  dim3 grid((unsigned)(ctx->N / 256 * limbs), (unsigned)n_o2, (unsigned)((n_o1 + 3) / 4));
  hipLaunchKernelGGL(dyadic_mac_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, d_cts_ntt, d_pts_ntt,
                     d_out_ntt, ctx->logN, limbs, (unsigned)n_o1, (unsigned)n_o2, (unsigned)K, ct_sk, ct_s2,
                     pt_s1, pt_sk, pt_s2, ctx->d_mods);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_mul_plain_acc(const mmh_ctx *ctx, int lv, const uint64_t *d_cts, size_t n_cts,
                                 const uint64_t *d_pts, size_t n_pts, size_t n_o1, size_t n_o2, size_t K,
                                 size_t ct_sk, size_t ct_s2, size_t pt_s1, size_t pt_sk, size_t pt_s2,
                                 uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n_o1 == 0 || n_o2 == 0) return MMH_OK;
  MMH_REQUIRE(d_cts && d_pts && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const int limbs = limbs_of(ctx, lv);
  const size_t poly = ctx->N * 8;
  StreamBuf ct_ntt(s), pt_ntt(s);
  MMH_HIP(ct_ntt.alloc(n_cts * 2 * limbs * poly));
  MMH_HIP(pt_ntt.alloc(n_pts * limbs * poly));
  MMH_HIP(hipMemcpyAsync(ct_ntt.p, d_cts, n_cts * 2 * limbs * poly, hipMemcpyDeviceToDevice, s));
  CHECK_RC(ntt_dispatch(ctx, false, ct_ntt.as<u64>(), n_cts * 2, limbs, 0, s));
  CHECK_RC(mmh_plain_lift_ntt(ctx, lv, d_pts, n_pts, pt_ntt.as<u64>(), stream));
  CHECK_RC(mmh_dyadic_mac(ctx, lv, ct_ntt.as<u64>(), n_cts, pt_ntt.as<u64>(), n_pts, n_o1, n_o2, K, ct_sk,
                          ct_s2, pt_s1, pt_sk, pt_s2, d_out, stream));
  return ntt_dispatch(ctx, true, d_out, n_o1 * n_o2 * 2, limbs, 0, s);
}

extern "C" int mmh_ct_add(const mmh_ctx *ctx, int lv, uint64_t *d_a, const uint64_t *d_b, size_t n,
                          void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_a && d_b, MMH_ERR_NULL_POINTER);
  const int limbs = limbs_of(ctx, lv);
  const size_t total = n * 2 * limbs * ctx->N;
  LAUNCH(poly_add_kernel, total, (hipStream_t)stream, d_a, d_b, total, ctx->logN, limbs, 0, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_mod_switch_to_last(const mmh_ctx *ctx, const uint64_t *d_in, size_t n, uint64_t *d_out,
                                      void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 0));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_in && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  if (ctx->L == 1) {
    MMH_HIP(hipMemcpyAsync(d_out, d_in, n * 2 * ctx->N * 8, hipMemcpyDeviceToDevice, s));
    return MMH_OK;
  }
  const size_t total = n * 2 * ctx->N;
  LAUNCH(mod_switch_kernel, total, s, d_in, d_out, total, ctx->logN, ctx->bfv.inv_qlast_mod_q0, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_encrypt_zero_asym(const mmh_ctx *ctx, int lv, const uint64_t *d_pk_ntt, const int64_t *d_u,
                                     const int64_t *d_e0, const int64_t *d_e1, size_t n, uint64_t *d_out,
                                     void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_pk_ntt && d_u && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const int limbs = limbs_of(ctx, lv);
  const size_t N = ctx->N;
  StreamBuf u(s);
  MMH_HIP(u.alloc(n * limbs * N * 8));
  LAUNCH(signed_to_rns_kernel, n * N, s, d_u, u.as<u64>(), n * N, ctx->logN, limbs, ctx->d_mods);
  CHECK_RC(ntt_dispatch(ctx, false, u.as<u64>(), n, limbs, 0, s));
  LAUNCH(mul_pk_kernel, n * limbs * N, s, u.as<u64>(), d_pk_ntt, d_out, n * limbs * N, ctx->logN, limbs,
         ctx->d_mods);
  CHECK_RC(ntt_dispatch(ctx, true, d_out, n * 2, limbs, 0, s));
  if (d_e0 || d_e1) LAUNCH(add_noise_kernel, n * N, s, d_out, d_e0, d_e1, n * N, ctx->logN, limbs, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_encrypt_symmetric(const mmh_ctx *ctx, int lv, const uint64_t *d_sk_ntt, const uint64_t *d_a,
                                     const int64_t *d_e, const uint64_t *d_pts, size_t n, uint64_t *d_out,
                                     void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_sk_ntt && d_a && d_e && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const int limbs = limbs_of(ctx, lv);
  const size_t N = ctx->N, total = n * limbs * N;
  StreamBuf as(s);
  MMH_HIP(as.alloc(total * 8));
  MMH_HIP(hipMemcpyAsync(as.p, d_a, total * 8, hipMemcpyDeviceToDevice, s));
  CHECK_RC(ntt_dispatch(ctx, false, as.as<u64>(), n, limbs, 0, s));
  LAUNCH(mul_key_kernel, total, s, as.as<u64>(), d_sk_ntt, total, ctx->logN, limbs, ctx->d_mods);
  CHECK_RC(ntt_dispatch(ctx, true, as.as<u64>(), n, limbs, 0, s));
  LAUNCH(sym_finish_kernel, n * N, s, as.as<u64>(), d_a, d_e, d_pts, d_out, n * N, ctx->logN, ctx->bfv, lv,
         ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_make_public_key(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_a_ntt,
                                   const int64_t *d_e, uint64_t *d_pk_ntt, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 0));
  MMH_REQUIRE(d_sk_ntt && d_a_ntt && d_e && d_pk_ntt, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const int limbs = (int)ctx->L;
  const size_t N = ctx->N;
  StreamBuf e(s);
  MMH_HIP(e.alloc(limbs * N * 8));
  LAUNCH(signed_to_rns_kernel, N, s, d_e, e.as<u64>(), N, ctx->logN, limbs, ctx->d_mods);
  CHECK_RC(ntt_dispatch(ctx, false, e.as<u64>(), 1, limbs, 0, s));
  LAUNCH(pk_kernel, limbs * N, s, d_sk_ntt, d_a_ntt, e.as<u64>(), d_pk_ntt, limbs * N, ctx->logN, limbs,
         ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_decrypt(const mmh_ctx *ctx, int lv, const uint64_t *d_sk_ntt, const uint64_t *d_cts,
                           size_t n, uint64_t *d_pts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_sk_ntt && d_cts && d_pts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  const int limbs = limbs_of(ctx, lv);
  const size_t N = ctx->N, total = n * limbs * N;
  StreamBuf tmp(s);
  MMH_HIP(tmp.alloc(total * 8));
  LAUNCH(gather_poly_kernel, total, s, d_cts, tmp.as<u64>(), total, ctx->logN, limbs, 1);
  CHECK_RC(ntt_dispatch(ctx, false, tmp.as<u64>(), n, limbs, 0, s));
  LAUNCH(mul_key_kernel, total, s, tmp.as<u64>(), d_sk_ntt, total, ctx->logN, limbs, ctx->d_mods);
  CHECK_RC(ntt_dispatch(ctx, true, tmp.as<u64>(), n, limbs, 0, s));
  LAUNCH(decrypt_finish_kernel, n * N, s, tmp.as<u64>(), d_cts, d_pts, n * N, ctx->logN, ctx->bfv, lv,
         ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_scale_and_round(const mmh_ctx *ctx, int lv, const uint64_t *d_polys, size_t n,
                                   const uint32_t *d_targets, size_t n_targets, uint64_t *d_out,
                                   void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (!d_targets) n_targets = ctx->N;
  if (n == 0 || n_targets == 0) return MMH_OK;
  MMH_REQUIRE(d_polys && d_out, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(n_targets <= ctx->N, MMH_ERR_OUT_BOUND);
  LAUNCH(scale_round_kernel, n * n_targets, (hipStream_t)stream, d_polys, d_targets, d_out, n, n_targets,
         ctx->logN, ctx->bfv, lv, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_sub_rns_poly(const mmh_ctx *ctx, int lv, uint64_t *d_cts, const uint64_t *d_r, size_t n,
                                void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_cts && d_r, MMH_ERR_NULL_POINTER);
  const int limbs = limbs_of(ctx, lv);
  const size_t total = n * limbs * ctx->N;
  LAUNCH(sub_c0_kernel, total, (hipStream_t)stream, d_cts, d_r, total, ctx->logN, limbs, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_flood_add(const mmh_ctx *ctx, uint64_t *d_cts, const uint64_t *d_random, size_t n, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 0));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_cts && d_random, MMH_ERR_NULL_POINTER);
  // noise_len = total_coeff_modulus_bit_count - bit_count(t) - 1, capped at 64 (:361-368); the mask
  // is (1 << (bitlen % 64)) - 1 as written at :154, which is 0 for bitlen = 64
  unsigned __int128 Q = 1;
  MMH_REQUIRE(ctx->L <= 2, MMH_ERR_CONFIG);
  for (size_t l = 0; l < ctx->L; ++l) Q *= ctx->h_mods[l].q;
  int q_bits = 0, t_bits = 0;
  while (Q) ++q_bits, Q >>= 1;
  for (u64 t = ctx->t; t; t >>= 1) ++t_bits;
  const int bitlen = std::min(64, q_bits - t_bits - 1);
  MMH_REQUIRE(bitlen >= 0, MMH_ERR_CONFIG);
  const u64 bitmask = (1ULL << (bitlen % 64)) - 1;
  LAUNCH(flood_add_kernel, n * ctx->N, (hipStream_t)stream, d_cts, d_random, n * ctx->N, ctx->logN, (int)ctx->L,
         bitmask, ctx->d_mods);
  return MMH_OK;
}

extern "C" int mmh_reduce_mod_t(const mmh_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t count,
                                void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 0));
  if (count == 0) return MMH_OK;
  MMH_REQUIRE(d_in && d_out, MMH_ERR_NULL_POINTER);
  LAUNCH(reduce_t_kernel, count, (hipStream_t)stream, d_in, d_out, count, ctx->bfv.t);
  return MMH_OK;
}

extern "C" int mmh_truncate_masks(const mmh_ctx *ctx, uint64_t *mask0, uint64_t *mask1) {
  CHECK_RC(check_bfv(ctx, 1));
  MMH_REQUIRE(mask0 && mask1, MMH_ERR_NULL_POINTER);
  *mask0 = ctx->bfv.mask0;
  *mask1 = ctx->bfv.mask1;
  return MMH_OK;
}

extern "C" int mmh_truncate_and_zero(const mmh_ctx *ctx, uint64_t *d_cts, size_t n, const uint8_t *d_keep,
                                     size_t keep_period, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 1));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_cts, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!d_keep || keep_period >= 1, MMH_ERR_INVALID_ARG);
  LAUNCH(truncate_kernel, n * ctx->N, (hipStream_t)stream, d_cts, n * ctx->N, ctx->logN, ctx->bfv.mask0,
         ctx->bfv.mask1, d_keep, keep_period ? keep_period : 1);
  return MMH_OK;
}

// addRandomMask + truncate_for_decryption + removeUnusedCoeffs of the shipped USE_APPROX_RESHARE build
// (hom_conv2d_ss.cc:761-788, :698-703, :800-851; hom_fc_ss.cc:473-561) on n accumulated ciphertexts:
// d_acc [n][2][2][N] at the first level, in NTT or coefficient form (consumed), d_out [n][2][N] at the last.
// Four launches: inverse NTT of acc, forward NTT of u straight from the signed samples, inverse NTT of
// its products with the key (formed while loading), and one pass that mod-switches, adds the encryption of zero and e1, subtracts the mask,
// truncates, clears the unused coefficients and (d_share_dst [keep_period][N]: target index or -1) scatters
// the mask's scale-and-round into the share tensor d_share_out [n / keep_period][share_plane].  Same bits as the step-by-step entry points.
namespace mmh {
int reshare_tail_approx_u(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_pk_ntt_last,
                          const int64_t *d_u, const u64 *d_u_ntt, const int64_t *d_e1, const u64 *d_mask,
                          const uint8_t *d_keep, size_t keep_period, u64 *d_out, hipStream_t s, const int *d_share_dst,
                          size_t share_plane, u64 *d_share_out, bool acc_switched = false);
int reshare_tail_approx(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_pk_ntt_last,
                        const int64_t *d_u,
                        const int64_t *d_e1, const u64 *d_mask, const uint8_t *d_keep, size_t keep_period,
                        u64 *d_out, hipStream_t s, const int *d_share_dst, size_t share_plane, u64 *d_share_out) {
  return reshare_tail_approx_u(ctx, d_acc, acc_ntt_form, n, d_pk_ntt_last, d_u, nullptr, d_e1, d_mask, d_keep,
                               keep_period, d_out, s, d_share_dst, share_plane, d_share_out);
}
size_t reshare_split_max() {
  const char *split_env = getenv("MMH_TAIL_SPLIT_MAX");
  return split_env ? (size_t)atol(split_env) : (size_t)128;
}
// d_u_ntt: u already transformed (the front launch of a direct-route layer does it beside its own work), else null
// acc_switched: d_acc holds the accumulator AFTER divide_and_round_q_last, [n][2][N] canonical mod q0 in coefficient
// form (the matrix-core product's paired items write it so), instead of [n][2][2][N] at the first level
int reshare_tail_approx_u(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_pk_ntt_last,
                          const int64_t *d_u, const u64 *d_u_ntt, const int64_t *d_e1, const u64 *d_mask,
                          const uint8_t *d_keep, size_t keep_period, u64 *d_out, hipStream_t s, const int *d_share_dst,
                          size_t share_plane, u64 *d_share_out, bool acc_switched) {
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  MMH_REQUIRE(!(acc_switched && acc_ntt_form), MMH_ERR_INTERNAL);
  const size_t N = ctx->N;
  StreamBuf u(s);
  if (!d_u_ntt) MMH_HIP(u.alloc(n * N * 8));
  if (acc_ntt_form) CHECK_RC(ntt_dispatch(ctx, true, d_acc, n * 2, 2, 0, s));
  if (!d_u_ntt) {
    CHECK_RC(ntt_signed_dispatch(ctx, d_u, u.as<u64>(), n, 1, s));
    d_u_ntt = u.as<u64>();
  }
  // Few output ciphertexts: the fused kernel has one workgroup per component and walks its coefficients in four
  // batches, each a memory round trip, so below the size where it fills the chip the encryption of zero is written
  // out and the tail runs one thread per coefficient instead; same bits.  Measured per layer call (GPU time, fused
  // against split): 64 ciphertexts 66.7 / 59.8 us, 256: 83.5 / 87.3, 512: 111.9 / 128.9, 1024: 203 / 243.
  // MMH_TAIL_SPLIT_MAX overrides the size (A/B measurements, tests of both forms).
  if (n <= reshare_split_max()) {
    StreamBuf z(s);
    MMH_HIP(z.alloc(n * 2 * N * 8));
    CHECK_RC(ntt_inv_product_dispatch(ctx, d_u_ntt, d_pk_ntt_last, z.as<u64>(), n, 1, s));
    return reshare_online_dispatch(ctx, z.as<u64>(), n, d_acc, d_e1, d_mask, d_out, d_keep, keep_period, d_share_dst,
                                   share_plane, d_share_out, s, acc_switched);
  }
  // u * pk_j back to coefficient form and, coefficient by coefficient, into the tail (ReshareSink, ntt.hip): the
  // encryption of zero is never written
  CHECK_RC(ntt_inv_product_reshare(ctx, d_u_ntt, d_pk_ntt_last, n, d_acc, d_e1, d_mask, d_out, d_keep,
                                   keep_period, d_share_dst, share_plane, d_share_out, s, acc_switched));
  return MMH_OK;
}
}  // namespace mmh

// The tail split at the point where the inputs enter it.  The pk-encryption of zero under the tail
// (asymmetric_encrypt_zero, hom_conv2d_ss.cc:275-340: u * pk, two of the tail's three transforms per ciphertext)
// depends on the key and on fresh randomness only, not on the layer's input: a deployment can make it while it waits
// for the client (an offline phase), and the online tail is then one pass.  The reference computes it inside
// conv2DSS on every call, so the headline figures keep it inside; this pair is reported beside them.
namespace mmh {
int reshare_tail_online(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_zero,
                        const int64_t *d_e1, const u64 *d_mask, const uint8_t *d_keep, size_t keep_period, u64 *d_out,
                        hipStream_t s, const int *d_share_dst, size_t share_plane, u64 *d_share_out,
                        bool acc_switched = false) {
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  MMH_REQUIRE(!(acc_switched && acc_ntt_form), MMH_ERR_INTERNAL);
  if (acc_ntt_form) CHECK_RC(ntt_dispatch(ctx, true, d_acc, n * 2, 2, 0, s));
  return reshare_online_dispatch(ctx, d_zero, n, d_acc, d_e1, d_mask, d_out, d_keep, keep_period, d_share_dst,
                                 share_plane, d_share_out, s, acc_switched);
}
}  // namespace mmh
extern "C" int mmh_reshare_zero_prepare(const mmh_ctx *ctx, const uint64_t *d_pk_ntt_last, const int64_t *d_u, size_t n,
                                        uint64_t *d_zero, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 1));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_pk_ntt_last && d_u && d_zero, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  StreamBuf u(s);
  MMH_HIP(u.alloc(n * ctx->N * 8));
  CHECK_RC(ntt_signed_dispatch(ctx, d_u, u.as<u64>(), n, 1, s));
  return ntt_inv_product_dispatch(ctx, u.as<u64>(), d_pk_ntt_last, d_zero, n, 1, s);
}

// Fresh symmetric ciphertexts travel as c0 plus the seed their c1 was expanded from (seal::Serializable<Ciphertext>
// of encrypt_symmetric, cheetah-api.cpp:11-29 on :401-406): the two halves of [n][2][limbs][N] apart and back together.
extern "C" int mmh_ct_split_c0(const mmh_ctx *ctx, int lv, const uint64_t *d_cts, size_t n, uint64_t *d_c0,
                               void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_cts && d_c0, MMH_ERR_NULL_POINTER);
  const size_t half = (size_t)ctx->bfv.lv[lv].limbs * ctx->N * 8;
  MMH_HIP(hipMemcpy2DAsync(d_c0, half, d_cts, 2 * half, half, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMH_OK;
}
extern "C" int mmh_ct_join(const mmh_ctx *ctx, int lv, const uint64_t *d_c0, const uint64_t *d_c1, size_t n,
                           uint64_t *d_cts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, lv));
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_cts && d_c0 && d_c1, MMH_ERR_NULL_POINTER);
  const size_t half = (size_t)ctx->bfv.lv[lv].limbs * ctx->N * 8;
  hipStream_t s = (hipStream_t)stream;
  MMH_HIP(hipMemcpy2DAsync(d_cts, 2 * half, d_c0, half, half, n, hipMemcpyDeviceToDevice, s));
  MMH_HIP(hipMemcpy2DAsync(reinterpret_cast<char *>(d_cts) + half, 2 * half, d_c1, half, half, n,
                           hipMemcpyDeviceToDevice, s));
  return MMH_OK;
}

extern "C" int mmh_batch_encode(const mmh_ctx *ctx, const uint64_t *d_vals, size_t n, uint64_t *d_pts,
                                void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 0));
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_vals && d_pts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  LAUNCH(slot_scatter_kernel, n * ctx->N, s, d_vals, d_pts, n * ctx->N, ctx->logN, ctx->d_slot_map, ctx->bfv.t);
  return ntt_dispatch(ctx, true, d_pts, n, 1, (int)ctx->L, s);
}

extern "C" int mmh_batch_decode(const mmh_ctx *ctx, const uint64_t *d_pts, size_t n, uint64_t *d_vals,
                                void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  CHECK_RC(check_bfv(ctx, 0));
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_vals && d_pts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  StreamBuf tmp(s);
  MMH_HIP(tmp.alloc(n * ctx->N * 8));
  MMH_HIP(hipMemcpyAsync(tmp.p, d_pts, n * ctx->N * 8, hipMemcpyDeviceToDevice, s));
  CHECK_RC(ntt_dispatch(ctx, false, tmp.as<u64>(), n, 1, (int)ctx->L, s));
  LAUNCH(slot_gather_kernel, n * ctx->N, s, tmp.as<u64>(), d_vals, n * ctx->N, ctx->logN, ctx->d_slot_map);
  return MMH_OK;
}

// ------------------------------------------------------------------- allocation-free decode
// mmh_batch_decode for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc).  The copy of the plaintexts that the forward transform works on lives in a file-scope device
// array instead of the scratch cache: 2^20 words = 8 MiB per GPU (256 plaintexts at N = 4096, the 2^20-element
// vector of tests/bole_test.cpp:568), held until the process ends.  Every GPU has its own copy of the array at
// its own address; the address is looked up once per device, with that device current, and host code uses
// that pointer only.  All calls on one GPU share the array: calls that overlap on one GPU (two streams, two
// threads) race.  Nothing in the array has to start at zero: a call overwrites the words it reads.
namespace mmh {
constexpr size_t kDecodeScratchWords = size_t(1) << 20;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned long long scratch[kDecodeScratchWords];
static u64 *g_decode_scratch[kMaxDevices];
static std::mutex g_decode_scratch_mu;

static int decode_scratch_ptr(int device, u64 **out) {
  MMH_REQUIRE(device >= 0 && device < kMaxDevices, MMH_ERR_CONFIG);
  std::lock_guard<std::mutex> lock(g_decode_scratch_mu);
  if (!g_decode_scratch[device]) {
    int current = -1;
    MMH_HIP(hipGetDevice(&current));
    MMH_REQUIRE(current == device, MMH_ERR_CONFIG);
    u64 *ptr = nullptr;
    MMH_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    MMH_REQUIRE(ptr != nullptr, MMH_ERR_INTERNAL);
    g_decode_scratch[device] = ptr;
  }
  *out = g_decode_scratch[device];
  return MMH_OK;
}
}  // namespace mmh

extern "C" int mmh_batch_decode_noalloc_max(const mmh_ctx *ctx, size_t *max_n) {
  MMH_REQUIRE(ctx && max_n, MMH_ERR_NULL_POINTER);
  *max_n = kDecodeScratchWords / ctx->N;
  return MMH_OK;
}

extern "C" int mmh_batch_decode_noalloc(const mmh_ctx *ctx, const uint64_t *d_pts, size_t n, uint64_t *d_vals,
                                        void *stream) {
  CHECK_RC(check_bfv(ctx, 0));
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  if (n > kDecodeScratchWords / ctx->N)
    MMH_FAIL(MMH_ERR_OUT_BOUND, "n = " + std::to_string(n) + " plaintexts exceed the static scratch of " +
                                    std::to_string(kDecodeScratchWords / ctx->N) + " (2^20 words per GPU)");
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_vals && d_pts, MMH_ERR_NULL_POINTER);
  int current = -1;
  MMH_HIP(hipGetDevice(&current));
  MMH_REQUIRE(current == ctx->device, MMH_ERR_CONFIG);  // the context's tables live on ctx->device
  (void)hipGetLastError();  // a stale error another library left behind is not this call's (DeviceScope does the same)
  u64 *tmp = nullptr;
  CHECK_RC(decode_scratch_ptr(current, &tmp));
  hipStream_t s = (hipStream_t)stream;
  MMH_HIP(hipMemcpyAsync(tmp, d_pts, n * ctx->N * 8, hipMemcpyDeviceToDevice, s));
  CHECK_RC(ntt_dispatch(ctx, false, tmp, n, 1, (int)ctx->L, s));
  LAUNCH(slot_gather_kernel, n * ctx->N, s, tmp, d_vals, n * ctx->N, ctx->logN, ctx->d_slot_map);
  return MMH_OK;
}

extern "C" int mmh_poly_op(int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t scalar, size_t n,
                           uint64_t modulus, uint64_t *d_out, void *stream) {
  MMH_REQUIRE(op >= 0 && op <= 4, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(modulus >= 2 && modulus < (uint64_t(1) << 62), MMH_ERR_INVALID_ARG);  // SEAL moduli are at most 61 bits
  if (n == 0) return MMH_OK;
  MMH_REQUIRE(d_a && d_out && (d_b || op >= 3), MMH_ERR_NULL_POINTER);
  const unsigned __int128 ratio =
      (modulus & (modulus - 1)) ? ~(unsigned __int128)0 / modulus : (~(unsigned __int128)0 / modulus) + 1;
  const ModC m{modulus, (u64)(ratio >> 64), (u64)ratio};
  hipStream_t s = (hipStream_t)stream;
  LAUNCH(poly_op_kernel, n, s, op, d_a, d_b, scalar, n, m, d_out);
  return MMH_OK;
}
