/*
 * mm_oracle.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the integer algorithms on M&M's HE-linear +
 * modulus-conversion hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; nothing under openmm_amd/ links or calls it.
 *
 * PARITY UNPINNED: the arithmetic lives in microsoft/SEAL @ 7923472 (v3.7.2,
 * patched by /root/reference/patch/SEAL.patch) and intel/hexl @ 343acab
 * (v1.2.2); neither is present in /root/reference and the reference holds no
 * golden vectors for this path.  The functions below
 * restate the published algorithms of those libraries and are anchored on the
 * reference's own call sites, which are cited per function.  What is pinned:
 * algebraic identities (negacyclic convolution vs schoolbook, decrypt(encrypt)
 * = m, share reconstruction as in tests/bole_test.cpp:164-167, etc.).
 */
#ifndef MM_ORACLE_H_
#define MM_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar modular arithmetic (SEAL util/uintarithsmallmod.h semantics) ---- */
uint64_t mmo_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t mmo_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t mmo_invmod(uint64_t a, uint64_t q);            /* q need not be prime; gcd(a,q)=1 */
int      mmo_is_prime(uint64_t n);
/* SEAL CoeffModulus::Create / util::get_primes rule.  Returns 0 on success. */
int      mmo_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out);
int      mmo_coeff_modulus_create(size_t N, const int *bits, size_t n, uint64_t *out);
/* minimal primitive (2N)-th root of unity mod q (SEAL try_minimal_primitive_root) */
uint64_t mmo_min_primitive_root(uint64_t two_n, uint64_t q);
/* floor(w * 2^64 / q): the Shoup / MultiplyUIntModOperand quotient */
uint64_t mmo_shoup(uint64_t w, uint64_t q);

/* ---- NTT tables & transforms (SEAL util/ntt.cpp, HEXL ntt; cited at
 *      hom_conv2d_ss.cc:306,315,333) ---- */
/* psi_rev[k] = psi^{bitrev(k, logN)}, ipsi_rev[k] = psi_rev[k]^{-1}; both length N */
void mmo_ntt_tables(size_t N, uint64_t q, uint64_t psi, uint64_t *psi_rev, uint64_t *ipsi_rev);
/* in-place, natural-order in, bit-reversed out, fully reduced */
void mmo_ntt_forward(uint64_t *a, size_t N, uint64_t q, const uint64_t *psi_rev);
/* in-place, bit-reversed in, natural out, scaled by N^-1, fully reduced */
void mmo_ntt_inverse(uint64_t *a, size_t N, uint64_t q, const uint64_t *ipsi_rev);
/* fast variants (Shoup + lazy Harvey butterflies) used for the timed CPU baseline;
 * same function, shoup tables = mmo_shoup of the table entries */
void mmo_ntt_forward_fast(uint64_t *a, size_t N, uint64_t q, const uint64_t *psi_rev,
                          const uint64_t *psi_rev_shoup);
void mmo_ntt_inverse_fast(uint64_t *a, size_t N, uint64_t q, const uint64_t *ipsi_rev,
                          const uint64_t *ipsi_rev_shoup);
/* schoolbook negacyclic product c = a*b mod (X^N+1, q): the identity the NTT is pinned by */
void mmo_negacyclic_schoolbook(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t N, uint64_t q);

/* ---- pointwise RNS ops (SEAL util/polyarithsmallmod.h) ---- */
void mmo_dyadic_mul(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n, uint64_t q);
void mmo_add_mod(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n, uint64_t q);
void mmo_sub_mod(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n, uint64_t q);

/* ---- BFV context restatement ---- */
#define MMO_MAX_L 4
typedef struct mmo_ctx {
  size_t   N;
  int      logN;
  size_t   L;                    /* limbs at the first (= key) level, patch/SEAL.patch:28-37 */
  uint64_t q[MMO_MAX_L];
  uint64_t psi[MMO_MAX_L];
  uint64_t *psi_rev[MMO_MAX_L], *ipsi_rev[MMO_MAX_L];
  uint64_t *psi_rev_sh[MMO_MAX_L], *ipsi_rev_sh[MMO_MAX_L];
  uint64_t t;                    /* plain modulus */
  int      t_is_pow2;
  /* per level lv in {0: all L limbs, 1: first limb only (last level)} */
  uint64_t delta_mod_q[2][MMO_MAX_L];   /* floor(Q/t) mod q_i */
  uint64_t q_mod_t[2];                  /* Q mod t */
  uint64_t upper_half_threshold;        /* (t+1)/2 */
  /* BEHZ decrypt_scale_and_round constants, per level */
  uint64_t gamma;
  uint64_t prod_t_gamma_mod_q[2][MMO_MAX_L];
  uint64_t inv_punct_mod_q[2][MMO_MAX_L];      /* (Q/q_i)^-1 mod q_i */
  uint64_t punct_mod_t[2][MMO_MAX_L];          /* Q/q_i mod t */
  uint64_t punct_mod_gamma[2][MMO_MAX_L];      /* Q/q_i mod gamma */
  uint64_t neg_inv_q_mod_t[2], neg_inv_q_mod_gamma[2];
  uint64_t inv_gamma_mod_t;
  /* mod-switch to last level: q_last^-1 mod q_i */
  uint64_t inv_qlast_mod_q[MMO_MAX_L];
  /* BatchEncoder (t prime, t = 1 mod 2N) */
  int      batching;
  uint64_t t_psi;
  uint64_t *t_psi_rev, *t_ipsi_rev;
  uint32_t *slot_map;            /* matrix_reps_index_map, length N */
} mmo_ctx;

/* bits[]: coeff-modulus bit sizes as in cheetah-api.cpp:96-105 ({60,49}); returns NULL on error */
mmo_ctx *mmo_ctx_create(size_t N, const int *bits, size_t L, uint64_t t);
/* explicit moduli instead of bit sizes */
mmo_ctx *mmo_ctx_create_q(size_t N, const uint64_t *q, size_t L, uint64_t t);
void     mmo_ctx_destroy(mmo_ctx *c);
size_t   mmo_level_limbs(const mmo_ctx *c, int lv);

/* fast plain lift of a coefficient-form plaintext (len coeffs, rest zero) into L RNS limbs:
 * out[l*N+i] = m + (m >= (t+1)/2 ? q_l - t : 0)      (SEAL Evaluator::multiply_plain_normal) */
void mmo_plain_lift(const mmo_ctx *c, int lv, const uint64_t *pt, size_t len, uint64_t *out);
/* ct (2 polys, lv limbs, coefficient form) times plaintext -> out (may alias ct); exact negacyclic
 * product per limb (Evaluator::multiply_plain for non-NTT BFV, hom_conv2d_ss.cc:610-613) */
void mmo_multiply_plain(const mmo_ctx *c, int lv, const uint64_t *ct, const uint64_t *pt, size_t len,
                        uint64_t *out);
/* c0 += / -= round(Q m / t)  (util::multiply_add/sub_plain_with_scaling_variant;
 * Evaluator::add_plain hom_conv2d_ss.cc:658, sub_plain :79) */
void mmo_add_plain(const mmo_ctx *c, int lv, uint64_t *ct, const uint64_t *pt, size_t len);
void mmo_sub_plain(const mmo_ctx *c, int lv, uint64_t *ct, const uint64_t *pt, size_t len);
/* ct add (Evaluator::add_inplace) over `polys` polynomials */
void mmo_ct_add(const mmo_ctx *c, int lv, uint64_t *a, const uint64_t *b, size_t polys);
/* mod_switch_to_inplace(first -> last) = divide_and_round_q_last per poly
 * (hom_conv2d_ss.cc:382).  in: polys*L*N, out: polys*(L-1)*N (only L-1 == 1.. supported) */
void mmo_mod_switch_to_last(const mmo_ctx *c, const uint64_t *in, uint64_t *out, size_t polys);
/* RNSTool::decrypt_scale_and_round on n (<= N) coefficients, RNS layout in[l*n+i]
 * (patch/SEAL.patch:125-175; hom_conv2d_ss.cc:220) */
void mmo_decrypt_scale_and_round(const mmo_ctx *c, int lv, const uint64_t *in, size_t n, uint64_t *out);

/* ---- BFV encryption with INJECTED randomness (so oracle and GPU consume identical streams) ----
 * sk_ntt: secret key in NTT form, L*N.  a: uniform poly in coefficient form (L*N, < q_l),
 * e: noise as signed values (N).  Output ct (2*L*N) in coefficient form:
 *   c1 = a ; c0 = -(a*s + e) + round(Q m/t)      (SEAL util/rlwe.cpp encrypt_zero_symmetric
 *   with is_ntt_form=false + Encryptor::encrypt_internal; hom_conv2d_ss.cc:516) */
void mmo_encrypt_symmetric(const mmo_ctx *c, int lv, const uint64_t *sk_ntt, const uint64_t *a,
                           const int64_t *e, const uint64_t *pt, size_t len, uint64_t *ct);
/* pk = (-(a s + e), a) in NTT form at lv 0 (KeyGenerator::create_public_key) */
void mmo_make_public_key(const mmo_ctx *c, const uint64_t *sk_ntt, const uint64_t *a_ntt,
                         const int64_t *e, uint64_t *pk);
/* asymmetric_encrypt_zero, non-NTT output (hom_conv2d_ss.cc:275-340): u ternary signed (N),
 * e0/e1 signed noise (N each; e0 may be NULL = skipped as under USE_APPROX_RESHARE :323-328).
 * pk given at level lv (L_lv limbs, NTT form). out: 2*L_lv*N */
void mmo_encrypt_zero_asymmetric(const mmo_ctx *c, int lv, const uint64_t *pk, const int64_t *u,
                                 const int64_t *e0, const int64_t *e1, uint64_t *out);
/* Decryptor::decrypt (BFV): out = scale_and_round(c0 + c1*s) , N coeffs (not trimmed) */
void mmo_decrypt(const mmo_ctx *c, int lv, const uint64_t *sk_ntt, const uint64_t *ct, uint64_t *out);
/* truncate_for_decryption masks (hom_conv2d_ss.cc:418-436) */
void mmo_truncate_masks(const mmo_ctx *c, uint64_t *mask0, uint64_t *mask1);

/* ---- BatchEncoder (hom_bn_ss.cc:176-180, :395) ---- */
void mmo_batch_encode(const mmo_ctx *c, const uint64_t *vals, size_t n, uint64_t *pt /* N */);
void mmo_batch_decode(const mmo_ctx *c, const uint64_t *pt /* N */, uint64_t *vals /* N */);

/* ---- modulus conversions: local share arithmetic (truncation.cpp:319-695) ----
 * party: 1 = ALICE (COT sender), 2 = BOB.  pre: tmp (and msb, and for ALICE the COT correlation).
 * post: consumes the COT output y exactly as the reference does after send_cot/recv_cot. */
void mmo_r2f_pre(int party, const uint64_t *in, size_t n, int bw, uint64_t p,
                 uint64_t *tmp, uint8_t *msb, uint64_t *corr);
void mmo_r2f_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                  int bw, uint64_t p, uint64_t *out);
void mmo_f2r_pre(int party, const uint64_t *in, size_t n, uint64_t p, int shift,
                 uint64_t *tmp, uint8_t *msb, uint64_t *corr);
void mmo_f2r_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                  uint64_t p, int bw, uint64_t *out);
void mmo_f2r_trunc_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                        uint64_t p, int bw, int shift, uint64_t *out);
void mmo_r2r_pre(int party, const uint64_t *in, size_t n, int bw_a,
                 uint64_t *tmp, uint8_t *msb, uint64_t *corr);
void mmo_r2r_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                  int bw_a, int bw_b, uint64_t *out);
/* 128-bit output as (lo,hi) pairs: out[2*i], out[2*i+1] */
void mmo_r2r128_post(int party, const uint64_t *tmp, const uint8_t *msb, const uint64_t *y, size_t n,
                     int bw_a, int bw_b, uint64_t *out);
/* local arithmetic of Truncation::truncate / truncate_msb / truncate_msb0 (truncation-cheetah.cpp:2-218) and
 * truncate_and_reduce (truncation.cpp:289-317); aw = the B2A output of the wrap bit */
void mmo_trunc_pre(int party, const uint64_t *in, size_t n, int bw, int signed_arith, int heuristic,
                   uint64_t *adj, uint64_t *cmp);
void mmo_trunc_post(int party, const uint64_t *adj, const uint64_t *aw, size_t n, int shift, int bw,
                    int signed_arith, int heuristic, uint64_t *out);
void mmo_trunc_reduce_pre(const uint64_t *in, size_t n, int shift, uint64_t *lower);
void mmo_trunc_reduce_post(const uint64_t *in, const uint64_t *aw, size_t n, int shift, int bw, uint64_t *out);
/* local arithmetic of ReLURingProtocol::relu (USE_CHEETAH build, relu-ring.h:216-252,:308-313) and of the two
 * COT-based helpers it and the truncations use: AuxProtocols::multiplexer and B2A (aux-protocols.cpp:62-126) */
void mmo_relu_pre(int party, const uint64_t *share, size_t n, int l, uint8_t *msb, uint64_t *cmp);
void mmo_relu_post(int party, const uint8_t *msb, const uint8_t *wrap, size_t n, uint8_t *drelu, uint8_t *sel);
void mmo_mux_pre(const uint8_t *sel, const uint64_t *x, size_t n, int bw_y, uint64_t *corr);
void mmo_mux_post(const uint8_t *sel, const uint64_t *x, const uint64_t *data_r, const uint64_t *data_s, size_t n,
                  int bw_y, uint64_t *y);
void mmo_b2a_pre(const uint8_t *x, size_t n, int bw_y, uint64_t *corr);
void mmo_b2a_post(int party, const uint8_t *x, const uint64_t *y, size_t n, int bw_y, uint64_t *out);
/* ideal COT functionality (silent_ot.h:111-116,198-204; pattern SCI/src/OT/ideal.h:106-156):
 * x = sender randomness (given), receiver gets x + b*corr mod 2^l (l<=64) or mod p (prime=1) */
void mmo_ideal_cot(const uint64_t *x, const uint64_t *corr, const uint8_t *b, size_t n,
                   int l, uint64_t p, int prime, uint64_t *recv);
/* sci::moduloMult (utils.hpp:353-363) and the BOLE cross term cheetah-api.cpp:648-649 */
uint64_t mmo_modulo_mult(uint64_t a, uint64_t b, uint64_t mod);
void mmo_bole_cross_term(uint64_t *out, const uint64_t *x, const uint64_t *y, size_t n, uint64_t p);

/* FusedBN (networks/main_resnet50.cpp:1728-1788): filter scaling in double arithmetic + round, and the bias add */
int mmo_fused_bn_filters(const uint64_t *filters, const uint64_t *bn_scales, long fh, long fw, long ci, long co,
                         int bitlength, int k_scale, uint64_t prime_mod, uint64_t *out);
void mmo_fused_bn_bias(uint64_t *out_nhwc, const uint64_t *bn_bias, size_t pixels, size_t co, int k_scale);

/* ---- COT post-processing (mm_cot_oracle.c): SilentOT::send/recv_ot_cam_cc[_prime], silent_ot.h:85-242 ----
 * AES-128 per FIPS-197; MITCCRH<8> restated from emp-tool 44b1dde [upstream-recall, parity unpinned] */
void mmo_aes128_expand(const uint8_t key[16], uint8_t rk[176]);
void mmo_aes128_encrypt(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16]);
void mmo_mitccrh_hash(const uint64_t S[2], uint64_t gid0, const uint64_t *blocks, size_t n, int H, uint64_t *out);
uint64_t mmo_cot_gids(size_t n);
size_t mmo_cot_wire_bytes(size_t n, int l, uint64_t p, int prime);
void mmo_cot_send(const uint64_t *rcm, const uint64_t delta[2], const uint64_t S[2], uint64_t gid0,
                  const uint64_t *corr, size_t n, int l, uint64_t p, int prime, uint64_t *data0, uint8_t *wire);
void mmo_cot_recv(const uint64_t *rcm, const uint8_t *b, const uint64_t S[2], uint64_t gid0, const uint8_t *wire,
                  size_t n, int l, uint64_t p, int prime, uint64_t *data);

/* ---- tensor <-> polynomial geometry (mm_geom_oracle.c) ---- */
int  mmo_conv_out_shape(const long ishape[3], const long fshape[3], int same, int stride, long out_hw[2]);
int  mmo_conv_shape_infer(long N, const long ishape[3], const long fshape[3], int same, int stride,
                          long strided[3], int pads[2], int slice_strides[3]);
long mmo_conv_encode_image(long N, uint64_t t, const uint64_t *img, const long ishape[3], const long fshape[3],
                           int same, int stride, uint64_t *polys);
long mmo_conv_encode_filter(long N, uint64_t t, const uint64_t *filter, const long ishape[3], const long fshape[3],
                            int same, int stride, uint64_t *polys);
typedef struct mmo_conv_indexer mmo_conv_indexer;
mmo_conv_indexer *mmo_conv_indexer_create(long N, const long ishape[3], const long fshape[3], int same, int stride);
void mmo_conv_indexer_destroy(mmo_conv_indexer *x);
int  mmo_conv_indexer_slices(const mmo_conv_indexer *x, int d);
int  mmo_conv_indexer_get(const mmo_conv_indexer *x, int sh, int sw, long out_hw[2], uint64_t *indices);
void mmo_fc_split(long nrows, long ncols, long N, long split[2]);
long mmo_fc_encode_input(long N, uint64_t t, const uint64_t *vec, long len_total, long d1, uint64_t *polys);
long mmo_fc_encode_weights(long N, uint64_t t, const uint64_t *W, long nrows, long ncols, long d0, long d1,
                           uint64_t *polys);

/* ---- timed CPU baseline helpers (bench.py cpu_baseline leg): OpenMP over independent polys */
/* AVX-512 restatement of the two lazy transforms (mm_oracle_avx512.c), same bits as the *_fast functions;
 * mmo_set_vector_isa(1) makes the batch functions below use it where the host has AVX-512F/DQ (returns what is
 * in force) */
int mmo_have_avx512(void);
int mmo_set_vector_isa(int on);
void mmo_ntt_forward_avx512(uint64_t *a, size_t N, uint64_t q, const uint64_t *psi_rev, const uint64_t *psi_rev_sh);
void mmo_ntt_inverse_avx512(uint64_t *a, size_t N, uint64_t q, const uint64_t *ipsi_rev, const uint64_t *ipsi_rev_sh,
                            uint64_t ninv, uint64_t ninv_sh);
void mmo_ntt_forward_batch(const mmo_ctx *c, uint64_t *polys, size_t n_polys, size_t limbs, int threads);
void mmo_ntt_inverse_batch(const mmo_ctx *c, uint64_t *polys, size_t n_polys, size_t limbs, int threads);
void mmo_conv_mac_reference(const mmo_ctx *c, const uint64_t *image, const uint64_t *filters, size_t M,
                            size_t Cs, size_t S, uint64_t *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
