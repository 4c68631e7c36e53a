/*
 * mm_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * See mm_oracle.h for the scope statement and the "parity unpinned" note.
 * Every function cites the reference call site (relative to /root/reference)
 * it restates, or the upstream SEAL 3.7.2 / HEXL 1.2.2 routine it re-derives.
 */
#include "mm_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ scalar */

uint64_t mmo_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t mmo_powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = mmo_mulmod(r, a, q);
    a = mmo_mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

/* extended Euclid; SEAL util::try_invert_uint_mod */
uint64_t mmo_invmod(uint64_t a, uint64_t q) {
  __int128 t0 = 0, t1 = 1;
  u128 r0 = q, r1 = a % q;
  while (r1) {
    u128 k = r0 / r1;
    u128 r2 = r0 - k * r1;
    __int128 t2 = t0 - (__int128)k * t1;
    r0 = r1; r1 = r2; t0 = t1; t1 = t2;
  }
  if (r0 != 1) return 0;
  if (t0 < 0) t0 += q;
  return (uint64_t)t0;
}

/* deterministic Miller-Rabin for 64-bit (SEAL util::is_prime is probabilistic MR; same set) */
int mmo_is_prime(uint64_t n) {
  static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return 0;
  for (size_t i = 0; i < 12; i++) {
    if (n == bases[i]) return 1;
    if (n % bases[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  int r = 0;
  while (!(d & 1)) { d >>= 1; r++; }
  for (size_t i = 0; i < 12; i++) {
    uint64_t x = mmo_powmod(bases[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int j = 1; j < r; j++) {
      x = mmo_mulmod(x, x, n);
      if (x == n - 1) { comp = 0; break; }
    }
    if (comp) return 0;
  }
  return 1;
}

/* SEAL util::get_primes(factor, bit_size, count): scan downward from
 * 2^bit_size - factor + 1 in steps of `factor`, stop at 2^(bit_size-1). */
int mmo_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out) {
  uint64_t value = ((((uint64_t)1) << bit_size) - 1) / factor * factor + 1;
  uint64_t lower = ((uint64_t)1) << (bit_size - 1);
  size_t got = 0;
  while (got < count && value > lower) {
    if (mmo_is_prime(value)) out[got++] = value;
    value -= factor;
  }
  return got == count ? 0 : -1;
}

/* SEAL CoeffModulus::Create(N, bit_sizes) (cheetah-api.cpp:98-99): per distinct bit
 * size take the k largest primes = 1 mod 2N; emit per position the LAST (smallest)
 * remaining one of that size. */
int mmo_coeff_modulus_create(size_t N, const int *bits, size_t n, uint64_t *out) {
  uint64_t tbl[64][MMO_MAX_L];
  size_t cnt[64] = {0}, left[64];
  for (size_t i = 0; i < n; i++) {
    if (bits[i] < 2 || bits[i] > 61 || n > MMO_MAX_L) return -1;
    cnt[bits[i]]++;
  }
  for (int b = 0; b < 64; b++) {
    left[b] = cnt[b];
    if (cnt[b] && mmo_get_primes(2 * (uint64_t)N, b, cnt[b], tbl[b])) return -1;
  }
  for (size_t i = 0; i < n; i++) out[i] = tbl[bits[i]][--left[bits[i]]];
  return 0;
}

/* SEAL util::try_minimal_primitive_root(degree = 2N, q) */
uint64_t mmo_min_primitive_root(uint64_t two_n, uint64_t q) {
  uint64_t root = 0;
  for (uint64_t g = 2; g < q; g++) {
    uint64_t r = mmo_powmod(g, (q - 1) / two_n, q);
    if (mmo_powmod(r, two_n / 2, q) == q - 1) { root = r; break; }
  }
  uint64_t gen_sq = mmo_mulmod(root, root, q), cur = root, best = root;
  for (uint64_t i = 0; i < two_n / 2; i++) {
    if (cur < best) best = cur;
    cur = mmo_mulmod(cur, gen_sq, q);
  }
  return best;
}

uint64_t mmo_shoup(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) / q); }

/* --------------------------------------------------------------------- NTT */

static inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
static int ilog2(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }

void mmo_ntt_tables(size_t N, uint64_t q, uint64_t psi, uint64_t *psi_rev, uint64_t *ipsi_rev) {
  int lg = ilog2(N);
  uint64_t ipsi = mmo_invmod(psi, q), p = 1, ip = 1;
  for (size_t k = 0; k < N; k++) {
    uint32_t r = bitrev((uint32_t)k, lg);
    psi_rev[r] = p;
    ipsi_rev[r] = ip;
    p = mmo_mulmod(p, psi, q);
    ip = mmo_mulmod(ip, ipsi, q);
  }
}

/* SEAL util::ntt_negacyclic_harvey == HEXL NTT::ComputeForward (function level):
 * Cooley-Tukey, natural in, bit-reversed out, output in [0,q).  SURVEY Appendix A. */
void mmo_ntt_forward(uint64_t *a, size_t N, uint64_t q, const uint64_t *psi_rev) {
  for (size_t m = 1, gap = N >> 1; m < N; m <<= 1, gap >>= 1)
    for (size_t i = 0; i < m; i++) {
      uint64_t W = psi_rev[m + i];
      for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
        uint64_t X = a[j], Y = mmo_mulmod(a[j + gap], W, q);
        a[j] = X + Y >= q ? X + Y - q : X + Y;
        a[j + gap] = X >= Y ? X - Y : X + q - Y;
      }
    }
}

/* SEAL util::inverse_ntt_negacyclic_harvey: Gentleman-Sande, bit-reversed in,
 * natural out, times N^-1, output in [0,q). */
void mmo_ntt_inverse(uint64_t *a, size_t N, uint64_t q, const uint64_t *ipsi_rev) {
  for (size_t m = N >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1)
    for (size_t i = 0; i < m; i++) {
      uint64_t W = ipsi_rev[m + i];
      for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
        uint64_t X = a[j], Y = a[j + gap];
        a[j] = X + Y >= q ? X + Y - q : X + Y;
        a[j + gap] = mmo_mulmod(X >= Y ? X - Y : X + q - Y, W, q);
      }
    }
  uint64_t ninv = mmo_invmod((uint64_t)N % q, q);
  for (size_t j = 0; j < N; j++) a[j] = mmo_mulmod(a[j], ninv, q);
}

static inline uint64_t shoup_mul_lazy(uint64_t x, uint64_t w, uint64_t wsh, uint64_t q) {
  uint64_t h = (uint64_t)(((u128)x * wsh) >> 64);
  return x * w - h * q; /* in [0, 2q) */
}

/* Harvey lazy butterflies, values kept in [0,4q) (needs q < 2^62) */
void mmo_ntt_forward_fast(uint64_t *a, size_t N, uint64_t q, const uint64_t *w, const uint64_t *wsh) {
  const uint64_t q2 = 2 * q;
  for (size_t m = 1, gap = N >> 1; m < N; m <<= 1, gap >>= 1)
    for (size_t i = 0; i < m; i++) {
      uint64_t W = w[m + i], Ws = wsh[m + i];
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        uint64_t X = x[j]; X -= (X >= q2) ? q2 : 0;
        uint64_t Y = shoup_mul_lazy(y[j], W, Ws, q);
        x[j] = X + Y;
        y[j] = X + q2 - Y;
      }
    }
  for (size_t j = 0; j < N; j++) {
    uint64_t v = a[j];
    v -= (v >= q2) ? q2 : 0;
    v -= (v >= q) ? q : 0;
    a[j] = v;
  }
}

void mmo_ntt_inverse_fast(uint64_t *a, size_t N, uint64_t q, const uint64_t *w, const uint64_t *wsh) {
  const uint64_t q2 = 2 * q;
  for (size_t m = N >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1)
    for (size_t i = 0; i < m; i++) {
      uint64_t W = w[m + i], Ws = wsh[m + i];
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        uint64_t X = x[j], Y = y[j];
        uint64_t s = X + Y; s -= (s >= q2) ? q2 : 0;
        x[j] = s;
        y[j] = shoup_mul_lazy(X + q2 - Y, W, Ws, q);
      }
    }
  uint64_t ninv = mmo_invmod((uint64_t)N % q, q), ninv_sh = mmo_shoup(ninv, q);
  for (size_t j = 0; j < N; j++) {
    uint64_t v = shoup_mul_lazy(a[j], ninv, ninv_sh, q);
    v -= (v >= q) ? q : 0;
    a[j] = v;
  }
}

void mmo_negacyclic_schoolbook(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t N, uint64_t q) {
  for (size_t k = 0; k < N; k++) c[k] = 0;
  for (size_t i = 0; i < N; i++) {
    if (!a[i]) continue;
    for (size_t j = 0; j < N; j++) {
      uint64_t p = mmo_mulmod(a[i], b[j], q);
      size_t k = i + j;
      if (k < N) { c[k] += p; if (c[k] >= q) c[k] -= q; }
      else { k -= N; c[k] = c[k] >= p ? c[k] - p : c[k] + q - p; }
    }
  }
}

/* --------------------------------------------------------------- pointwise */
/* util::dyadic_product_coeffmod (hom_conv2d_ss.cc:308) */
void mmo_dyadic_mul(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n, uint64_t q) {
  for (size_t i = 0; i < n; i++) c[i] = mmo_mulmod(a[i], b[i], q);
}
/* util::add_poly_coeffmod (hom_conv2d_ss.cc:335,375) */
void mmo_add_mod(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n, uint64_t q) {
  for (size_t i = 0; i < n; i++) { uint64_t s = a[i] + b[i]; c[i] = s >= q ? s - q : s; }
}
/* util::sub_poly_coeffmod (hom_conv2d_ss.cc:99) */
void mmo_sub_mod(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n, uint64_t q) {
  for (size_t i = 0; i < n; i++) c[i] = a[i] >= b[i] ? a[i] - b[i] : a[i] + q - b[i];
}

/* ----------------------------------------------------------------- context */

size_t mmo_level_limbs(const mmo_ctx *c, int lv) { return lv == 0 ? c->L : 1; }

static u128 level_Q(const mmo_ctx *c, int lv) {
  u128 Q = 1;
  for (size_t i = 0; i < mmo_level_limbs(c, lv); i++) Q *= c->q[i];
  return Q;
}

mmo_ctx *mmo_ctx_create_q(size_t N, const uint64_t *q, size_t L, uint64_t t) {
  if (L < 1 || L > MMO_MAX_L) return NULL;
  mmo_ctx *c = (mmo_ctx *)calloc(1, sizeof(mmo_ctx));
  c->N = N; c->logN = ilog2(N); c->L = L; c->t = t;
  c->t_is_pow2 = t && !(t & (t - 1));
  for (size_t l = 0; l < L; l++) {
    c->q[l] = q[l];
    c->psi[l] = mmo_min_primitive_root(2 * N, q[l]);
    c->psi_rev[l] = (uint64_t *)malloc(N * 8); c->ipsi_rev[l] = (uint64_t *)malloc(N * 8);
    c->psi_rev_sh[l] = (uint64_t *)malloc(N * 8); c->ipsi_rev_sh[l] = (uint64_t *)malloc(N * 8);
    mmo_ntt_tables(N, q[l], c->psi[l], c->psi_rev[l], c->ipsi_rev[l]);
    for (size_t k = 0; k < N; k++) {
      c->psi_rev_sh[l][k] = mmo_shoup(c->psi_rev[l][k], q[l]);
      c->ipsi_rev_sh[l][k] = mmo_shoup(c->ipsi_rev[l][k], q[l]);
    }
  }
  if (t < 2) return c;             /* NTT-only context (N=8192 micro-benchmark, 4 limbs) */
  if (L > 2) return c;             /* BFV constants need Q < 2^128 */
  c->upper_half_threshold = (t + 1) >> 1;
  /* gamma: SEAL RNSTool::initialize -> get_primes(2N, 61, ...)[1] */
  uint64_t g[2];
  mmo_get_primes(2 * N, 61, 2, g);
  c->gamma = g[1];
  c->inv_gamma_mod_t = mmo_invmod(c->gamma % t, t);
  for (int lv = 0; lv < 2; lv++) {
    size_t Ll = mmo_level_limbs(c, lv);
    u128 Q = level_Q(c, lv);
    c->q_mod_t[lv] = (uint64_t)(Q % t);
    u128 delta = Q / t;
    for (size_t i = 0; i < Ll; i++) {
      uint64_t qi = c->q[i];
      c->delta_mod_q[lv][i] = (uint64_t)(delta % qi);
      u128 punct = Q / qi;
      c->inv_punct_mod_q[lv][i] = mmo_invmod((uint64_t)(punct % qi), qi);
      c->punct_mod_t[lv][i] = (uint64_t)(punct % t);
      c->punct_mod_gamma[lv][i] = (uint64_t)(punct % c->gamma);
      c->prod_t_gamma_mod_q[lv][i] = mmo_mulmod(t % qi, c->gamma % qi, qi);
    }
    uint64_t Qt = (uint64_t)(Q % t), Qg = (uint64_t)(Q % c->gamma);
    uint64_t it = mmo_invmod(Qt, t), ig = mmo_invmod(Qg, c->gamma);
    c->neg_inv_q_mod_t[lv] = it ? (t - it) % t : 0;
    c->neg_inv_q_mod_gamma[lv] = c->gamma - ig;
  }
  for (size_t i = 0; i + 1 < L; i++) c->inv_qlast_mod_q[i] = mmo_invmod(c->q[L - 1] % c->q[i], c->q[i]);
  /* BatchEncoder: t prime and t = 1 mod 2N (ContextData qualifiers().using_batching) */
  if (mmo_is_prime(t) && (t % (2 * N)) == 1) {
    c->batching = 1;
    c->t_psi = mmo_min_primitive_root(2 * N, t);
    c->t_psi_rev = (uint64_t *)malloc(N * 8); c->t_ipsi_rev = (uint64_t *)malloc(N * 8);
    mmo_ntt_tables(N, t, c->t_psi, c->t_psi_rev, c->t_ipsi_rev);
    c->slot_map = (uint32_t *)malloc(N * 4);
    /* BatchEncoder::populate_matrix_reps_index_map: generator 3 */
    size_t m = 2 * N, row = N >> 1;
    uint64_t pos = 1;
    for (size_t i = 0; i < row; i++) {
      uint64_t i1 = (pos - 1) >> 1, i2 = (m - pos - 1) >> 1;
      c->slot_map[i] = bitrev((uint32_t)i1, c->logN);
      c->slot_map[row + i] = bitrev((uint32_t)i2, c->logN);
      pos = (pos * 3) & (m - 1);
    }
  }
  return c;
}

mmo_ctx *mmo_ctx_create(size_t N, const int *bits, size_t L, uint64_t t) {
  uint64_t q[MMO_MAX_L];
  if (L > MMO_MAX_L || mmo_coeff_modulus_create(N, bits, L, q)) return NULL;
  return mmo_ctx_create_q(N, q, L, t);
}

void mmo_ctx_destroy(mmo_ctx *c) {
  if (!c) return;
  for (size_t l = 0; l < c->L; l++) {
    free(c->psi_rev[l]); free(c->ipsi_rev[l]); free(c->psi_rev_sh[l]); free(c->ipsi_rev_sh[l]);
  }
  free(c->t_psi_rev); free(c->t_ipsi_rev); free(c->slot_map);
  free(c);
}

static void fwd(const mmo_ctx *c, size_t l, uint64_t *a) {
  mmo_ntt_forward_fast(a, c->N, c->q[l], c->psi_rev[l], c->psi_rev_sh[l]);
}
static void fwd_any(const mmo_ctx *c, size_t l, uint64_t *a);  /* fwd / inv, or their AVX-512 restatement when */
static void inv_any(const mmo_ctx *c, size_t l, uint64_t *a);  /* mmo_set_vector_isa(1) is in force: same bits  */
static void inv(const mmo_ctx *c, size_t l, uint64_t *a) {
  mmo_ntt_inverse_fast(a, c->N, c->q[l], c->ipsi_rev[l], c->ipsi_rev_sh[l]);
}

/* SEAL Evaluator::multiply_plain_normal, using_fast_plain_lift branch:
 * coefficient >= plain_upper_half_threshold gets plain_upper_half_increment[l] = q_l - t */
void mmo_plain_lift(const mmo_ctx *c, int lv, const uint64_t *pt, size_t len, uint64_t *out) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  for (size_t l = 0; l < Ll; l++)
    for (size_t i = 0; i < N; i++) {
      uint64_t m = i < len ? pt[i] : 0;
      out[l * N + i] = m >= c->upper_half_threshold ? m + (c->q[l] - c->t) : m;
    }
}

void mmo_multiply_plain(const mmo_ctx *c, int lv, const uint64_t *ct, const uint64_t *pt, size_t len,
                        uint64_t *out) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  uint64_t *p = (uint64_t *)malloc(Ll * N * 8), *tmp = (uint64_t *)malloc(N * 8);
  mmo_plain_lift(c, lv, pt, len, p);
  for (size_t l = 0; l < Ll; l++) fwd_any(c, l, p + l * N);
  for (size_t k = 0; k < 2; k++)
    for (size_t l = 0; l < Ll; l++) {
      const uint64_t *src = ct + (k * Ll + l) * N;
      memcpy(tmp, src, N * 8);
      fwd_any(c, l, tmp);
      mmo_dyadic_mul(tmp, p + l * N, tmp, N, c->q[l]);
      inv_any(c, l, tmp);
      memcpy(out + (k * Ll + l) * N, tmp, N * 8);
    }
  free(p); free(tmp);
}

/* SEAL util::multiply_add_plain_with_scaling_variant (scalingvariant.cpp):
 * fix = floor((m*(Q mod t) + (t+1)/2) / t); scaled = m*(floor(Q/t) mod q_l) + fix mod q_l */
static void scale_plain(const mmo_ctx *c, int lv, uint64_t *ct, const uint64_t *pt, size_t len, int sub) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  for (size_t i = 0; i < len && i < N; i++) {
    u128 num = (u128)pt[i] * c->q_mod_t[lv] + c->upper_half_threshold;
    uint64_t fix = (uint64_t)(num / c->t);
    for (size_t l = 0; l < Ll; l++) {
      uint64_t q = c->q[l];
      uint64_t s = (uint64_t)(((u128)pt[i] * c->delta_mod_q[lv][l] + fix) % q);
      uint64_t *d = ct + l * N + i;
      if (!sub) { uint64_t v = *d + s; *d = v >= q ? v - q : v; }
      else *d = *d >= s ? *d - s : *d + q - s;
    }
  }
}
void mmo_add_plain(const mmo_ctx *c, int lv, uint64_t *ct, const uint64_t *pt, size_t len) {
  scale_plain(c, lv, ct, pt, len, 0);
}
void mmo_sub_plain(const mmo_ctx *c, int lv, uint64_t *ct, const uint64_t *pt, size_t len) {
  scale_plain(c, lv, ct, pt, len, 1);
}

void mmo_ct_add(const mmo_ctx *c, int lv, uint64_t *a, const uint64_t *b, size_t polys) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  for (size_t k = 0; k < polys; k++)
    for (size_t l = 0; l < Ll; l++)
      mmo_add_mod(a + (k * Ll + l) * N, b + (k * Ll + l) * N, a + (k * Ll + l) * N, N, c->q[l]);
}

/* SEAL RNSTool::divide_and_round_q_last_inplace (BFV mod_switch_to_next):
 * last += q_last/2; for each i<L-1: tmp = last mod q_i - (q_last/2 mod q_i);
 * x_i = (x_i - tmp) * q_last^-1 mod q_i.   SURVEY Appendix A. */
void mmo_mod_switch_to_last(const mmo_ctx *c, const uint64_t *in, uint64_t *out, size_t polys) {
  size_t L = c->L, N = c->N;
  if (L == 1) { memcpy(out, in, polys * N * 8); return; }
  uint64_t ql = c->q[L - 1], half = ql >> 1;
  for (size_t k = 0; k < polys; k++)
    for (size_t i = 0; i < N; i++) {
      uint64_t last = in[(k * L + L - 1) * N + i] + half;
      if (last >= ql) last -= ql;
      for (size_t l = 0; l + 1 < L; l++) {
        uint64_t q = c->q[l];
        uint64_t tmp = last % q, hq = half % q;
        tmp = tmp >= hq ? tmp - hq : tmp + q - hq;
        uint64_t x = in[(k * L + l) * N + i];
        x = x >= tmp ? x - tmp : x + q - tmp;
        out[(k * (L - 1) + l) * N + i] = mmo_mulmod(x, c->inv_qlast_mod_q[l], q);
      }
    }
}

/* SEAL RNSTool::decrypt_scale_and_round, patched to accept n <= N coefficients
 * (patch/SEAL.patch:125-175). */
void mmo_decrypt_scale_and_round(const mmo_ctx *c, int lv, const uint64_t *in, size_t n, uint64_t *out) {
  size_t Ll = mmo_level_limbs(c, lv);
  uint64_t t = c->t, g = c->gamma, g2 = g >> 1;
  for (size_t i = 0; i < n; i++) {
    u128 acc_t = 0, acc_g = 0;
    for (size_t l = 0; l < Ll; l++) {
      uint64_t q = c->q[l];
      uint64_t x = mmo_mulmod(in[l * n + i], c->prod_t_gamma_mod_q[lv][l], q);
      x = mmo_mulmod(x, c->inv_punct_mod_q[lv][l], q);
      acc_t = (acc_t + (u128)x * c->punct_mod_t[lv][l]) % t;
      acc_g = (acc_g + (u128)x * c->punct_mod_gamma[lv][l]) % g;
    }
    uint64_t vt = mmo_mulmod((uint64_t)acc_t, c->neg_inv_q_mod_t[lv], t);
    uint64_t vg = mmo_mulmod((uint64_t)acc_g, c->neg_inv_q_mod_gamma[lv], g);
    uint64_t r;
    if (vg > g2) { uint64_t d = (g - vg) % t; r = vt + d; if (r >= t) r -= t; }
    else { uint64_t d = vg % t; r = vt >= d ? vt - d : vt + t - d; }
    if (r) r = mmo_mulmod(r, c->inv_gamma_mod_t, t);
    out[i] = r;
  }
}

/* ------------------------------------------------------------- encryption */

static inline uint64_t signed_to_mod(int64_t v, uint64_t q) { return v >= 0 ? (uint64_t)v : q - (uint64_t)(-v); }

void mmo_encrypt_symmetric(const mmo_ctx *c, int lv, const uint64_t *sk_ntt, const uint64_t *a,
                           const int64_t *e, const uint64_t *pt, size_t len, uint64_t *ct) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  uint64_t *tmp = (uint64_t *)malloc(N * 8);
  for (size_t l = 0; l < Ll; l++) {
    uint64_t q = c->q[l];
    uint64_t *c0 = ct + l * N, *c1 = ct + (Ll + l) * N;
    memcpy(c1, a + l * N, N * 8);
    memcpy(tmp, a + l * N, N * 8);
    fwd(c, l, tmp);
    mmo_dyadic_mul(tmp, sk_ntt + l * N, tmp, N, q);
    inv(c, l, tmp);
    for (size_t i = 0; i < N; i++) {
      uint64_t v = tmp[i] + signed_to_mod(e[i], q);
      if (v >= q) v -= q;
      c0[i] = v ? q - v : 0;
    }
  }
  free(tmp);
  if (pt) mmo_add_plain(c, lv, ct, pt, len);
}

void mmo_make_public_key(const mmo_ctx *c, const uint64_t *sk_ntt, const uint64_t *a_ntt,
                         const int64_t *e, uint64_t *pk) {
  size_t L = c->L, N = c->N;
  uint64_t *tmp = (uint64_t *)malloc(N * 8);
  for (size_t l = 0; l < L; l++) {
    uint64_t q = c->q[l];
    for (size_t i = 0; i < N; i++) tmp[i] = signed_to_mod(e[i], q);
    fwd(c, l, tmp);
    for (size_t i = 0; i < N; i++) {
      uint64_t v = mmo_mulmod(a_ntt[l * N + i], sk_ntt[l * N + i], q) + tmp[i];
      if (v >= q) v -= q;
      pk[l * N + i] = v ? q - v : 0;
      pk[(L + l) * N + i] = a_ntt[l * N + i];
    }
  }
  free(tmp);
}

void mmo_encrypt_zero_asymmetric(const mmo_ctx *c, int lv, const uint64_t *pk, const int64_t *u,
                                 const int64_t *e0, const int64_t *e1, uint64_t *out) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  uint64_t *un = (uint64_t *)malloc(N * 8);
  for (size_t l = 0; l < Ll; l++) {
    uint64_t q = c->q[l];
    for (size_t i = 0; i < N; i++) un[i] = signed_to_mod(u[i], q);
    fwd(c, l, un);
    for (size_t j = 0; j < 2; j++) {
      uint64_t *d = out + (j * Ll + l) * N;
      mmo_dyadic_mul(un, pk + (j * Ll + l) * N, d, N, q);
      inv(c, l, d);
      const int64_t *e = j == 0 ? e0 : e1;
      if (e)
        for (size_t i = 0; i < N; i++) {
          uint64_t v = d[i] + signed_to_mod(e[i], q);
          d[i] = v >= q ? v - q : v;
        }
    }
  }
  free(un);
}

/* SEAL Decryptor::bfv_decrypt: dot_product_ct_sk_array then decrypt_scale_and_round */
void mmo_decrypt(const mmo_ctx *c, int lv, const uint64_t *sk_ntt, const uint64_t *ct, uint64_t *out) {
  size_t Ll = mmo_level_limbs(c, lv), N = c->N;
  uint64_t *tmp = (uint64_t *)malloc(Ll * N * 8);
  for (size_t l = 0; l < Ll; l++) {
    uint64_t q = c->q[l], *d = tmp + l * N;
    memcpy(d, ct + (Ll + l) * N, N * 8);
    fwd(c, l, d);
    mmo_dyadic_mul(d, sk_ntt + l * N, d, N, q);
    inv(c, l, d);
    mmo_add_mod(d, ct + l * N, d, N, q);
  }
  mmo_decrypt_scale_and_round(c, lv, tmp, N, out);
  free(tmp);
}

static int bit_count(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }
static uint64_t make_bits_mask(int n_low_zeros) { /* hom_conv2d_ss.cc:59-63 */
  if (n_low_zeros < 0) n_low_zeros = 0;
  if (n_low_zeros > 63) n_low_zeros = 63;
  return (~(uint64_t)0 >> n_low_zeros) << n_low_zeros;
}
/* hom_conv2d_ss.cc:418-436.  n_var_bits = (int)log2(12*N*sqrt(1/18)) evaluated in integers:
 * 12*N/sqrt(18) = 2.828427..*N, so floor(log2) = logN + 1 for every power-of-two N. */
void mmo_truncate_masks(const mmo_ctx *c, uint64_t *mask0, uint64_t *mask1) {
  int n_delta_bits = bit_count(c->q[0]) - bit_count(c->t);
  int one_more = c->t_is_pow2 ? 0 : 1;
  int n_var_bits = c->logN + 1;
  *mask0 = make_bits_mask(n_delta_bits - 1 - one_more);
  *mask1 = make_bits_mask(n_delta_bits - n_var_bits);
}

/* ------------------------------------------------------------ BatchEncoder */
/* SEAL BatchEncoder::encode(vector<uint64_t>): scatter by index map, zero fill,
 * inverse NTT mod t.  Inputs may be non-canonical (< 2^l, Appendix B): the reference
 * feeds them to inverse_ntt_negacyclic_harvey unchecked (hom_bn_ss.cc:176-180);
 * the function of those values mod t is what the transform defines, so reduce first. */
void mmo_batch_encode(const mmo_ctx *c, const uint64_t *vals, size_t n, uint64_t *pt) {
  size_t N = c->N;
  for (size_t i = 0; i < N; i++) pt[i] = 0;
  for (size_t i = 0; i < n && i < N; i++) pt[c->slot_map[i]] = vals[i] % c->t;
  mmo_ntt_inverse(pt, N, c->t, c->t_ipsi_rev);
}
void mmo_batch_decode(const mmo_ctx *c, const uint64_t *pt, uint64_t *vals) {
  size_t N = c->N;
  uint64_t *tmp = (uint64_t *)malloc(N * 8);
  memcpy(tmp, pt, N * 8);
  mmo_ntt_forward(tmp, N, c->t, c->t_psi_rev);
  for (size_t i = 0; i < N; i++) vals[i] = tmp[c->slot_map[i]];
  free(tmp);
}

/* --------------------------------------------------- modulus conversions */
#define ALICE 1
static inline uint64_t bw_mask(int bw) { return bw == 64 ? ~(uint64_t)0 : (((uint64_t)1) << bw) - 1; }
static inline uint64_t secure_add(uint64_t x, uint64_t y, uint64_t p) { /* truncation.cpp:533-536 */
  return x >= (p - y) ? x - (p - y) : x + y;
}
static inline uint64_t secure_sub(uint64_t x, uint64_t y, uint64_t p) { /* truncation.cpp:538-541 */
  return x >= y ? x - y : x + (p - y);
}

/* truncation.cpp:525-531 (+big), :561-564 (msb), :546-548 (corr), :566 (corr value) */
void mmo_r2f_pre(int party, const uint64_t *in, size_t n, int bw, uint64_t p,
                 uint64_t *tmp, uint8_t *msb, uint64_t *corr) {
  uint64_t mask = bw_mask(bw), msb_mask = ((uint64_t)1) << (bw - 1), big = ((uint64_t)1) << (bw - 2);
  uint64_t cv = mask % p + 1;
  for (size_t i = 0; i < n; i++) {
    tmp[i] = party == ALICE ? (in[i] + big) & mask : in[i];
    msb[i] = (tmp[i] & msb_mask) != 0;
    if (party == ALICE && corr) corr[i] = msb[i] ? 0 : cv;
  }
}
/* truncation.cpp:551-554 (sender fix-up), :569-578 */
void mmo_r2f_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                  int bw, uint64_t p, uint64_t *out) {
  uint64_t mask = bw_mask(bw), big = ((uint64_t)1) << (bw - 2), cv = mask % p + 1;
  for (size_t i = 0; i < n; i++) {
    uint64_t yy = y[i];
    if (party == ALICE) {
      uint64_t t = secure_sub(0, yy, p);
      yy = msb[i] ? secure_add(cv, t, p) : t;
    }
    uint64_t o = secure_sub(tmp[i], yy, p);
    if (party == ALICE) o = secure_sub(o, big % p, p);
    out[i] = o;
  }
}
/* truncation.cpp:591-598 / :649-656 (+p/4), :618-621 (msb), :603-606 / :661-664 (corr) */
void mmo_f2r_pre(int party, const uint64_t *in, size_t n, uint64_t p, int shift,
                 uint64_t *tmp, uint8_t *msb, uint64_t *corr) {
  uint64_t big = p / 4, half = p / 2 + 1, cv = p >> shift;
  for (size_t i = 0; i < n; i++) {
    if (party == ALICE) tmp[i] = in[i] >= p - big ? in[i] - p + big : in[i] + big;
    else tmp[i] = in[i];
    msb[i] = tmp[i] >= half;
    if (party == ALICE && corr) corr[i] = msb[i] ? 0 : cv;
  }
}
/* truncation.cpp:609-611, :626-634 */
void mmo_f2r_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                  uint64_t p, int bw, uint64_t *out) {
  uint64_t mask = bw_mask(bw), big = p / 4;
  for (size_t i = 0; i < n; i++) {
    uint64_t yy = y[i];
    if (party == ALICE) yy = msb[i] ? ((p - yy) & mask) : ((0 - yy) & mask);
    out[i] = party == ALICE ? (tmp[i] - yy - big) & mask : (tmp[i] - yy) & mask;
  }
}
/* truncation.cpp:667-669, :684-692 */
void mmo_f2r_trunc_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                        uint64_t p, int bw, int shift, uint64_t *out) {
  uint64_t mask = bw_mask(bw), big = p / 4, cv = p >> shift;
  for (size_t i = 0; i < n; i++) {
    uint64_t yy = y[i];
    if (party == ALICE) yy = msb[i] ? ((cv - yy) & mask) : ((0 - yy) & mask);
    out[i] = party == ALICE ? (((tmp[i] - big) >> shift) - yy) & mask
                            : ((tmp[i] >> shift) - yy + 1) & mask;
  }
}
/* truncation.cpp:334-342, :365-368, :349-351 */
void mmo_r2r_pre(int party, const uint64_t *in, size_t n, int bw_a,
                 uint64_t *tmp, uint8_t *msb, uint64_t *corr) {
  uint64_t mask = bw_mask(bw_a), msb_mask = ((uint64_t)1) << (bw_a - 1), big = ((uint64_t)1) << (bw_a - 2);
  for (size_t i = 0; i < n; i++) {
    tmp[i] = party == ALICE ? (in[i] + big) & mask : in[i];
    msb[i] = (tmp[i] & msb_mask) != 0;
    if (party == ALICE && corr) corr[i] = msb[i] ^ 1;
  }
}
/* truncation.cpp:353-355, :371-379 (and :327-330 when bw_a >= bw_b: tmp is then the raw input) */
void mmo_r2r_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                  int bw_a, int bw_b, uint64_t *out) {
  uint64_t bmask = bw_mask(bw_b), big = ((uint64_t)1) << (bw_a - 2), ymask = bw_mask(bw_b - bw_a);
  for (size_t i = 0; i < n; i++) {
    uint64_t yy = y[i];
    if (party == ALICE) yy = ((uint64_t)msb[i] - yy) & ymask;
    uint64_t o = tmp[i] - (yy << bw_a);
    if (party == ALICE) o -= big;
    out[i] = o & bmask;
  }
}
/* truncation.cpp:414-416, :434-442 */
void mmo_r2r128_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                     int bw_a, int bw_b, uint64_t *out) {
  u128 bmask = bw_b == 128 ? ~(u128)0 : (((u128)1) << bw_b) - 1;
  uint64_t big = ((uint64_t)1) << (bw_a - 2), ymask = bw_mask(bw_b - bw_a);
  for (size_t i = 0; i < n; i++) {
    uint64_t yy = y[i];
    if (party == ALICE) yy = ((uint64_t)msb[i] - yy) & ymask;
    u128 o = (u128)tmp[i] - (((u128)yy) << bw_a);
    if (party == ALICE) o -= (u128)big;
    o &= bmask;
    out[2 * i] = (uint64_t)o;
    out[2 * i + 1] = (uint64_t)(o >> 64);
  }
}

/* ---- FusedBN filter scaling, networks/main_resnet50.cpp:1728-1766 (server branch), with getSignValue :21-28 and
 * sci::neg_mod (SCI/src/utils/utils.hpp:229-231).  filters and out are [fh][fw][ci][co] (Arr4DIdxRowM), scales [co].
 * The only floating point on the path: double arithmetic and std::round exactly as written.  Returns 1 when every
 * scaled filter entry is zero (the reference then prints a warning and exits, :1754-1762), else 0. */
#include <math.h>
static int64_t get_sign_value(uint64_t x, int bitlength) {           /* :21-28 */
  int64_t upper = (int64_t)1 << bitlength;
  uint64_t mask = (uint64_t)upper - 1, half = (uint64_t)upper >> 1;
  x &= mask;
  return (int64_t)(x >= half ? x - (uint64_t)upper : x);
}
static int64_t neg_mod_i64(int64_t val, int64_t mod) { return ((val % mod) + mod) % mod; }
int mmo_fused_bn_filters(const uint64_t *filters, const uint64_t *bn_scales, long fh, long fw, long ci, long co,
                         int bitlength, int k_scale, uint64_t prime_mod, uint64_t *out) {
  const double scale = pow(2., k_scale);                             /* :1736 */
  int all_zero = 1;
  for (long h = 0; h < fh; h++)
    for (long w = 0; w < fw; w++)
      for (long c = 0; c < ci; c++)
        for (long m = 0; m < co; m++) {
          const double bn_scale = get_sign_value(bn_scales[m], bitlength) / scale;          /* :1741 */
          const size_t idx = (size_t)((h * fw + w) * ci + c) * (size_t)co + (size_t)m;
          const double fval = get_sign_value(filters[idx], bitlength) / scale;             /* :1747 */
          const int64_t f64 = (int64_t)round(fval * bn_scale * scale);                      /* :1748 */
          out[idx] = (uint64_t)neg_mod_i64(f64, (int64_t)prime_mod);                        /* :1749 */
          if (out[idx]) all_zero = 0;
        }
  return all_zero;
}
/* the bias the server adds after the convolution, :1771-1786: ScaleUp1 (arr << sf in the SCI_OT build,
 * library_fixed_uniform.cpp:1394-1402) then SecretAdd = plain wrapping add (:1861-1867), NHWC output */
void mmo_fused_bn_bias(uint64_t *out_nhwc, const uint64_t *bn_bias, size_t pixels, size_t co, int k_scale) {
  for (size_t p = 0; p < pixels; p++)
    for (size_t m = 0; m < co; m++) out_nhwc[p * co + m] += bn_bias[m] << k_scale;
}

/* ---- Truncation::truncate / truncate_msb / truncate_msb0 of the USE_CHEETAH build (truncation-cheetah.cpp) and
 * truncate_and_reduce (truncation.cpp:289-317): the local share arithmetic around their wrap / B2A sub-protocols.
 * pre: the msb0 heuristic's +2^(bw-3) on BOB (truncation-cheetah.cpp:23-29), the signed case's +2^(bw-1) on ALICE
 * (:42-47, :102-107, :183-188), and the operand of mill->compare (:51-56): adj & mask, on BOB mask - that. */
void mmo_trunc_pre(int party, const uint64_t *in, size_t n, int bw, int signed_arith, int heuristic,
                   uint64_t *adj, uint64_t *cmp) {
  uint64_t mask = bw_mask(bw);
  for (size_t i = 0; i < n; i++) {
    uint64_t v = in[i];
    if (heuristic && party != ALICE) v = (v + (((uint64_t)1) << (bw - 3))) & mask;
    if (signed_arith && party == ALICE) v = (v + (((uint64_t)1) << (bw - 1))) & mask;
    adj[i] = v;
    if (cmp) {
      uint64_t c = v & mask;
      cmp[i] = party == ALICE ? c : (mask - c) & mask;
    }
  }
}
/* post: truncation-cheetah.cpp:64-75 (= :133-144 = :201-212) with aw the B2A output mod 2^shift, then the
 * heuristic's -2^(bw-3-shift) on BOB (:32-34) */
void mmo_trunc_post(int party, const uint64_t *adj, const uint64_t *aw, size_t n, int shift, int bw,
                    int signed_arith, int heuristic, uint64_t *out) {
  uint64_t mask = bw_mask(bw), mask_upper = bw_mask(bw - shift);
  for (size_t i = 0; i < n; i++) {
    uint64_t o = (((adj[i] >> shift) & mask_upper) - (((uint64_t)1) << (bw - shift)) * aw[i]) & mask;
    if (signed_arith && party == ALICE) o = (o - (((uint64_t)1) << (bw - shift - 1))) & mask;
    if (heuristic && party != ALICE) o = (o - (((uint64_t)1) << (bw - 3 - shift))) & mask;
    out[i] = o;
  }
}
/* truncation.cpp:300-304: the operand of wrap_computation; :311-313: the result in the smaller ring */
void mmo_trunc_reduce_pre(const uint64_t *in, size_t n, int shift, uint64_t *lower) {
  uint64_t mask_shift = bw_mask(shift);
  for (size_t i = 0; i < n; i++) lower[i] = in[i] & mask_shift;
}
void mmo_trunc_reduce_post(const uint64_t *in, const uint64_t *aw, size_t n, int shift, int bw, uint64_t *out) {
  uint64_t mask_out = bw_mask(bw - shift);
  for (size_t i = 0; i < n; i++) out[i] = ((in[i] >> shift) + aw[i]) & mask_out;
}

/* ---- ReLURingProtocol::relu of the USE_CHEETAH build (SCI/src/NonLinear/relu-ring.h:216-252, :308-313):
 * msb of the share and the operand of millionaire->compare on l-1 bits (ALICE: low l-1 bits, BOB: 2^(l-1)-1 - them) */
void mmo_relu_pre(int party, const uint64_t *share, size_t n, int l, uint8_t *msb, uint64_t *cmp) {
  uint64_t cut = (((uint64_t)1) << (l - 1)) - 1;
  for (size_t i = 0; i < n; i++) {
    msb[i] = (uint8_t)(share[i] >> (l - 1));
    uint64_t low = share[i] & cut;
    cmp[i] = party == ALICE ? low : cut - low;
  }
}
/* :244-252 DReLU share = (msb + wrap) % 2; :308-311 the multiplexer's selector (ALICE flips hers) */
void mmo_relu_post(int party, const uint8_t *msb, const uint8_t *wrap, size_t n, uint8_t *drelu, uint8_t *sel) {
  for (size_t i = 0; i < n; i++) {
    drelu[i] = (uint8_t)((msb[i] + wrap[i]) % 2);
    sel[i] = party == ALICE ? drelu[i] ^ 1 : drelu[i];
  }
}
/* AuxProtocols::multiplexer (SCI/src/BuildingBlocks/aux-protocols.cpp:62-93): the correlation of its two COTs
 * (:76-78) and the output from the received and the sent COT strings (:86-88) */
void mmo_mux_pre(const uint8_t *sel, const uint64_t *x, size_t n, int bw_y, uint64_t *corr) {
  uint64_t mask = bw_mask(bw_y);
  for (size_t i = 0; i < n; i++) corr[i] = (x[i] * (1 - 2 * (uint64_t)sel[i])) & mask;
}
void mmo_mux_post(const uint8_t *sel, const uint64_t *x, const uint64_t *data_r, const uint64_t *data_s, size_t n,
                  int bw_y, uint64_t *y) {
  uint64_t mask = bw_mask(bw_y);
  for (size_t i = 0; i < n; i++) y[i] = (x[i] * (uint64_t)sel[i] + data_r[i] - data_s[i]) & mask;
}
/* AuxProtocols::B2A (aux-protocols.cpp:95-126): ALICE's correlation -2x (:107-109) and both outputs from the COT
 * string y (:112-114 ALICE x - y, :119-121 BOB x + y); bw_y = 1 is the bit itself (:97-102, y ignored) */
void mmo_b2a_pre(const uint8_t *x, size_t n, int bw_y, uint64_t *corr) {
  uint64_t mask = bw_mask(bw_y);
  for (size_t i = 0; i < n; i++) corr[i] = ((uint64_t)0 - 2 * (uint64_t)x[i]) & mask;
}
void mmo_b2a_post(int party, const uint8_t *x, const uint64_t *y, size_t n, int bw_y, uint64_t *out) {
  uint64_t mask = bw_mask(bw_y);
  for (size_t i = 0; i < n; i++) {
    if (bw_y == 1) out[i] = (uint64_t)x[i] & 1;
    else out[i] = party == ALICE ? ((uint64_t)x[i] - y[i]) & mask : ((uint64_t)x[i] + y[i]) & mask;
  }
}

void mmo_ideal_cot(const uint64_t *x, const uint64_t *corr, const uint8_t *b, size_t n,
                   int l, uint64_t p, int prime, uint64_t *recv) {
  for (size_t i = 0; i < n; i++) {
    if (prime) {
      uint64_t v = x[i] % p;
      if (b[i]) v = (uint64_t)(((u128)v + corr[i] % p) % p);
      recv[i] = v;
    } else {
      uint64_t mask = bw_mask(l);
      recv[i] = (x[i] + (b[i] ? corr[i] : 0)) & mask;
    }
  }
}

uint64_t mmo_modulo_mult(uint64_t a, uint64_t b, uint64_t mod) { /* utils.hpp:353-363, literal */
  uint64_t res = 0;
  a %= mod;
  while (b) {
    if (b & 1) res = (res + a) % mod;
    a = (2 * a) % mod;
    b >>= 1;
  }
  return res;
}
/* cheetah-api.cpp:648-649 */
void mmo_bole_cross_term(uint64_t *out, const uint64_t *x, const uint64_t *y, size_t n, uint64_t p) {
  for (size_t i = 0; i < n; i++) out[i] = (out[i] + mmo_modulo_mult(x[i], y[i], p)) % p;
}

/* ------------------------------------------------- CPU baseline batches */
/* vector = 1: the AVX-512 restatement (mm_oracle_avx512.c) where the host has it; the results are the same bits */
static int g_vector = 0;
int mmo_set_vector_isa(int on) {
  g_vector = on && mmo_have_avx512();
  return g_vector;
}
static void fwd_any(const mmo_ctx *c, size_t l, uint64_t *a) {
  if (g_vector) mmo_ntt_forward_avx512(a, c->N, c->q[l], c->psi_rev[l], c->psi_rev_sh[l]);
  else fwd(c, l, a);
}
static void inv_any(const mmo_ctx *c, size_t l, uint64_t *a) {
  if (g_vector) {
    uint64_t ninv = mmo_invmod((uint64_t)c->N % c->q[l], c->q[l]);
    mmo_ntt_inverse_avx512(a, c->N, c->q[l], c->ipsi_rev[l], c->ipsi_rev_sh[l], ninv, mmo_shoup(ninv, c->q[l]));
  } else {
    inv(c, l, a);
  }
}
void mmo_ntt_forward_batch(const mmo_ctx *c, uint64_t *polys, size_t n_polys, size_t limbs, int threads) {
  long total = (long)(n_polys * limbs);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long b = 0; b < total; b++) fwd_any(c, (size_t)b % limbs, polys + (size_t)b * c->N);
}
void mmo_ntt_inverse_batch(const mmo_ctx *c, uint64_t *polys, size_t n_polys, size_t limbs, int threads) {
  long total = (long)(n_polys * limbs);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long b = 0; b < total; b++) inv_any(c, (size_t)b % limbs, polys + (size_t)b * c->N);
}

/* conv2DOneFilter for every filter (hom_conv2d_ss.cc:563-619, :679-690) in the reference's own
 * schedule -- one multiply_plain (+ add_inplace) per (filter, channel slice, spatial slice), workers
 * over filters like its ThreadPool.  image [Cs*S] cts at level 0, filters [M][Cs] plaintexts,
 * out [M*S] cts.  Used for parity of large layers and as the timed CPU baseline. */
void mmo_conv_mac_reference(const mmo_ctx *c, const uint64_t *image, const uint64_t *filters, size_t M,
                            size_t Cs, size_t S, uint64_t *out, int threads) {
  const size_t N = c->N, ct = 2 * c->L * N;
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (long m = 0; m < (long)M; m++) {
    uint64_t *tmp = (uint64_t *)malloc(ct * 8);
    for (size_t i = 0; i < S; i++) {
      int started = 0;
      uint64_t *dst = out + ((size_t)m * S + i) * ct;
      for (size_t cc = 0; cc < Cs; cc++) {
        const uint64_t *pt = filters + ((size_t)m * Cs + cc) * N;
        int nz = 0;
        for (size_t k = 0; k < N && !nz; k++) nz = pt[k] != 0;
        if (!nz) continue;
        const uint64_t *src = image + (cc * S + i) * ct;
        if (started) {
          mmo_multiply_plain(c, 0, src, pt, N, tmp);
          mmo_ct_add(c, 0, dst, tmp, 2);
        } else {
          mmo_multiply_plain(c, 0, src, pt, N, dst);
          started = 1;
        }
      }
    }
    free(tmp);
  }
}
