/*
 * mm_hip.h -- C ABI of libmm_hip.so: MI355X (gfx950) kernels for M&M's HE-linear
 * and modulus-conversion hot path.
 *
 * The reference has no FFI for this path: its arithmetic is reached through SEAL
 * 3.7.2 / HEXL 1.2.2 C++ calls made by the .cc files of include/gemini/cheetah and through the
 * plain loops of SCI/src/BuildingBlocks/truncation.cpp.  Each entry point below names
 * the reference call (file:line, relative to /root/reference) it replaces.
 *
 * Conventions
 *  - every function returns an int equal to gemini::Code
 *    (include/gemini/core/types.h:30-40): 0 OK, 1 ERR_CONFIG, 2 ERR_NULL_POINTER,
 *    3 ERR_DIM_MISMATCH, 4 ERR_SEAL_MEMORY, 5 ERR_KEY_MISSING, 6 ERR_OUT_BOUND,
 *    7 ERR_INVALID_ARG, 8 ERR_INTERNAL.  mmh_last_error() gives the text.
 *  - pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA storage), all
 *    others are host pointers.  The library never frees caller memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls
 *    enqueue work and return; use mmh_stream_sync to wait.
 *  - polynomial layout is SEAL's: poly k, limb l, coefficient i at
 *    base[(k*L + l)*N + i]  (hom_conv2d_ss.cc:96-103,134-138,305-311).
 *  - a context is immutable after creation, so calls are re-entrant across host
 *    threads (the reference enters the same const methods from a ThreadPool,
 *    hom_conv2d_ss.cc:511,559,667,793).
 */
#ifndef MM_HIP_H_
#define MM_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MMH_OK = 0, MMH_ERR_CONFIG, MMH_ERR_NULL_POINTER, MMH_ERR_DIM_MISMATCH, MMH_ERR_MEMORY,
  MMH_ERR_KEY_MISSING, MMH_ERR_OUT_BOUND, MMH_ERR_INVALID_ARG, MMH_ERR_INTERNAL
};

typedef struct mmh_ctx mmh_ctx;

const char *mmh_last_error(void);
int mmh_device_count(int *count);

/* ---- context: replaces seal::SEALContext + NTTTables + RNSTool precomputation
 * (cheetah-api.cpp:95-105, :151-162).  q[0..L) = first-level RNS base (no special prime,
 * patch/SEAL.patch:28-37); last level = {q[0]}.  t = plain modulus (0 = NTT-only context,
 * any L <= 4; BFV contexts need L <= 2).  Tables are built on the host and uploaded to
 * `device`. */
int mmh_ctx_create(mmh_ctx **out, size_t N, const uint64_t *q, size_t L, uint64_t t, int device);
/* same, from SEAL CoeffModulus::Create bit sizes, e.g. {60,49} (cheetah-api.cpp:98-99) */
int mmh_ctx_create_bits(mmh_ctx **out, size_t N, const int *bits, size_t L, uint64_t t, int device);
int mmh_ctx_destroy(mmh_ctx *ctx);
/* introspection: moduli, roots, gamma (for tests / host facade) */
int mmh_ctx_info(const mmh_ctx *ctx, size_t *N, size_t *L, uint64_t *q /*L*/, uint64_t *psi /*L*/,
                 uint64_t *t, uint64_t *gamma);

/* ---- device memory helpers for C/C++ callers that do not link HIP themselves */
int mmh_malloc(void **d_ptr, size_t bytes);
int mmh_free(void *d_ptr);
/* Layer calls keep their scratch buffers cached between calls (reused in stream order, no per-call
 * allocation once warm).  This returns the cached blocks of the current device to the driver after a
 * device synchronisation.  No counterpart in the reference (SEAL's MemoryPool plays that role there). */
int mmh_scratch_release(void);
/* Page-locked host memory for staging buffers: copies to and from it run at the link's rate (pageable memory goes
 * through the driver's bounce buffers).  No counterpart in the reference, whose data never leaves the host. */
int mmh_host_alloc(void **h_ptr, size_t bytes);
int mmh_host_free(void *h_ptr);
/* The current device and streams for C/C++ callers that do not link HIP: a host object creates one stream, enqueues
 * its copies and layer calls on it and waits once (mmh_stream_sync) before it hands bytes to the network. */
int mmh_get_device(int *device);
int mmh_set_device(int device);
int mmh_stream_create(void **stream);
int mmh_stream_destroy(void *stream);
/* Events (hipEvent_t as void*, timing disabled): a point in a stream that the host can wait for without waiting for
 * what was enqueued behind it -- how a caller reads an early verdict (mmh_conv_server_tensors_ev) while the rest of
 * its call still runs.  No counterpart in the reference. */
int mmh_event_create(void **event);
int mmh_event_destroy(void *event);
int mmh_event_record(void *event, void *stream);
int mmh_event_sync(void *event);
/* Page-lock memory the caller already owns (a buffer mapped by several processes, a transport's send buffer) so that
 * copies to and from it run asynchronously at the link's rate; fails with ERR_SEAL_MEMORY when the driver refuses the
 * range (the copy then still works, through the driver's bounce buffers).  No counterpart in the reference. */
int mmh_host_register(void *h_ptr, size_t bytes);
int mmh_host_unregister(void *h_ptr);
int mmh_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *stream);
int mmh_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *stream);
int mmh_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
int mmh_memset(void *d_dst, int value, size_t bytes, void *stream);
int mmh_stream_sync(void *stream);

/* ---- batched negacyclic NTT, in place.
 * d_polys holds n_polys polynomials of `limbs` limbs each ([n_polys][limbs][N]); limb l is
 * transformed modulo q[l].  Forward: natural order in, bit-reversed order out, values in
 * [0,q) -- seal::util::ntt_negacyclic_harvey (hom_conv2d_ss.cc:306,333; HEXL
 * NTT::ComputeForward underneath).  Inverse: bit-reversed in, natural out, scaled by N^-1,
 * values in [0,q) -- seal::util::inverse_ntt_negacyclic_harvey (hom_conv2d_ss.cc:315).
 * Inputs must be < 4q (forward) / < 2q (inverse). */
int mmh_ntt_forward(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limbs, void *stream);
int mmh_ntt_inverse(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limbs, void *stream);
/* same transforms modulo the plain modulus t (BatchEncoder, hom_bn_ss.cc:176-180,:395) */
int mmh_ntt_forward_t(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, void *stream);
int mmh_ntt_inverse_t(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, void *stream);

/* the same transforms one limb at a time: every polynomial of the batch [n_polys][N] modulo q[limb_index], as the
 * reference's own loops call them with ntt_tables[i] (hom_conv2d_ss.cc:306,315,333) */
int mmh_ntt_forward_limb(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limb_index, void *stream);
int mmh_ntt_inverse_limb(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limb_index, void *stream);
/* seal::util polynomial arithmetic modulo one word-sized modulus (seal/util/polyarithsmallmod.h), element-wise on n
 * coefficients: op 0 add_poly_coeffmod (hom_conv2d_ss.cc:335,375), 1 sub_poly_coeffmod (:99), 2 dyadic_product_coeffmod
 * (:308), 3 multiply_poly_scalar_coeffmod by `scalar` (hom_bn_ss.cc:637), 4 modulo_poly_coeffs (hom_conv2d_ss.cc:264,
 * tensor_encoder.cc:317).  d_b is unused for ops 3 and 4; d_out may alias an operand. */
int mmh_poly_op(int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t scalar, size_t n, uint64_t modulus,
                uint64_t *d_out, void *stream);

/* =====================================================================================
 * BFV arithmetic on batches of ciphertexts / plaintexts (BFV contexts only: t >= 2, L <= 2).
 * lv selects the level: 0 = first level (all L limbs), 1 = last level ({q0} only,
 * context.last_parms_id() in the reference).  A ciphertext is [2][limbs(lv)][N], a
 * coefficient-form plaintext is [N] (values < t), both in coefficient (non-NTT) form unless
 * a name says ntt.
 * ===================================================================================== */

/* c0 += round(Q m/t) (sub = 0) or c0 -= (sub = 1) for n ciphertexts and n plaintexts.
 * seal::Evaluator::add_plain / add_plain_inplace / sub_plain_inplace
 * (hom_conv2d_ss.cc:658, :79; hom_bn_ss.cc:264, :355).  d_out may equal d_cts. */
int mmh_add_plain(const mmh_ctx *ctx, int lv, const uint64_t *d_cts, const uint64_t *d_pts, size_t n,
                  int sub, uint64_t *d_out, void *stream);

/* "fast plain lift" of n plaintexts into RNS form followed by the forward NTT:
 * d_out[n][limbs][N].  First half of seal::Evaluator::multiply_plain for non-NTT BFV
 * (hom_conv2d_ss.cc:610,613); exposed so that static weights can be transformed once. */
int mmh_plain_lift_ntt(const mmh_ctx *ctx, int lv, const uint64_t *d_pts, size_t n, uint64_t *d_out,
                       void *stream);

/* Dyadic multiply-accumulate on NTT-form operands:
 *   out[o1][o2] = sum_{k<K} ct[k*ct_sk + o2*ct_s2] (.) pt[o1*pt_s1 + k*pt_sk + o2*pt_s2]
 * for o1 < n_o1, o2 < n_o2; out is [n_o1*n_o2] ciphertexts in NTT form, fully reduced.
 * seal::util::dyadic_product_coeffmod + add_poly_coeffmod as reached from
 * conv2DOneFilter (hom_conv2d_ss.cc:597-616), matVecMul (hom_fc_ss.cc:443-456) and
 * bole/bn (hom_bn_ss.cc:264-268, :304-306). */
int mmh_dyadic_mac(const mmh_ctx *ctx, int lv, const uint64_t *d_cts_ntt, size_t n_cts,
                   const uint64_t *d_pts_ntt, size_t n_pts, size_t n_o1, size_t n_o2, size_t K,
                   size_t ct_sk, size_t ct_s2, size_t pt_s1, size_t pt_sk, size_t pt_s2,
                   uint64_t *d_out_ntt, void *stream);

/* The whole multiply_plain + add_inplace schedule on coefficient-form inputs and outputs:
 * NTT(cts) once, lift+NTT(pts) once, dyadic MAC, one inverse NTT per output polynomial.
 * Bit-identical to the reference's per-term multiply_plain/add_inplace loop because every
 * step is exact modulo q_l.  Workspace is taken from the stream-ordered pool. */
int mmh_mul_plain_acc(const mmh_ctx *ctx, int lv, const uint64_t *d_cts, size_t n_cts,
                      const uint64_t *d_pts, size_t n_pts, size_t n_o1, size_t n_o2, size_t K,
                      size_t ct_sk, size_t ct_s2, size_t pt_s1, size_t pt_sk, size_t pt_s2,
                      uint64_t *d_out, void *stream);

/* a += b on n ciphertexts (seal::Evaluator::add_inplace, hom_conv2d_ss.cc:386,611) */
int mmh_ct_add(const mmh_ctx *ctx, int lv, uint64_t *d_a, const uint64_t *d_b, size_t n, void *stream);

/* first level -> last level: RNSTool::divide_and_round_q_last per polynomial
 * (seal::Evaluator::mod_switch_to_inplace, hom_conv2d_ss.cc:382,408; hom_bn_ss.cc:269,307).
 * d_in [n][2][L][N] -> d_out [n][2][1][N]. */
int mmh_mod_switch_to_last(const mmh_ctx *ctx, const uint64_t *d_in, size_t n, uint64_t *d_out, void *stream);

/* Public-key encryption of zero with injected randomness, coefficient-form output:
 * asymmetric_encrypt_zero (hom_conv2d_ss.cc:275-340).  d_pk_ntt [2][limbs(lv)][N] in NTT form,
 * d_u [n][N] ternary in {-1,0,1}, d_e0 / d_e1 [n][N] signed noise (d_e0 may be NULL: the
 * USE_APPROX_RESHARE build skips e0, :323-328).  d_out [n][2][limbs][N]. */
int mmh_encrypt_zero_asym(const mmh_ctx *ctx, int lv, const uint64_t *d_pk_ntt, const int64_t *d_u,
                          const int64_t *d_e0, const int64_t *d_e1, size_t n, uint64_t *d_out, void *stream);

/* Secret-key encryption with injected randomness: c1 = a, c0 = -(a s + e) + round(Q m/t).
 * seal::Encryptor::encrypt_symmetric (hom_conv2d_ss.cc:516, hom_fc_ss.cc:257, hom_bn_ss.cc:180).
 * d_sk_ntt [limbs][N] NTT form; d_a [n][limbs][N] uniform < q_l; d_e [n][N] signed;
 * d_pts [n][N] or NULL (encrypt_zero_symmetric, hom_bn_ss.cc:535). */
int mmh_encrypt_symmetric(const mmh_ctx *ctx, int lv, const uint64_t *d_sk_ntt, const uint64_t *d_a,
                          const int64_t *d_e, const uint64_t *d_pts, size_t n, uint64_t *d_out, void *stream);

/* Public key from injected randomness: pk = (-(a s + e), a) in NTT form at the first level
 * (seal::KeyGenerator::create_public_key, cheetah-api.cpp:113,167). d_a_ntt [L][N], d_e [N]. */
int mmh_make_public_key(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_a_ntt,
                        const int64_t *d_e, uint64_t *d_pk_ntt, void *stream);

/* BFV decryption of n ciphertexts to n plaintexts [n][N] (coefficients < t, NOT trimmed of
 * leading zeros): seal::Decryptor::decrypt (hom_conv2d_ss.cc:906, hom_fc_ss.cc:590,
 * hom_bn_ss.cc:389) = dot product with sk, then RNSTool::decrypt_scale_and_round in the
 * patched form (patch/SEAL.patch:125-175). */
int mmh_decrypt(const mmh_ctx *ctx, int lv, const uint64_t *d_sk_ntt, const uint64_t *d_cts, size_t n,
                uint64_t *d_pts, void *stream);

/* RNSTool::decrypt_scale_and_round on gathered coefficients: out[k][j] =
 * scale_and_round( polys[k][.][targets[j]] ), polys [n][limbs][N], targets [n_targets] shared
 * by all n (approximate re-sharing, sample_random_mask hom_conv2d_ss.cc:209-221).
 * d_targets = NULL means all N coefficients in order. */
int mmh_scale_and_round(const mmh_ctx *ctx, int lv, const uint64_t *d_polys, size_t n,
                        const uint32_t *d_targets, size_t n_targets, uint64_t *d_out, void *stream);

/* ct.c0 -= r for an RNS-form polynomial r [n][limbs][N] (internal::sub_poly_inplace with a
 * parms_id-carrying plaintext, hom_conv2d_ss.cc:83-103) */
int mmh_sub_rns_poly(const mmh_ctx *ctx, int lv, uint64_t *d_cts, const uint64_t *d_r, size_t n, void *stream);

/* truncate_for_decryption (hom_conv2d_ss.cc:392-441) fused with remove_unused_coeffs
 * (:108-140): on n last-level ciphertexts, c0 &= mask0, c1 &= mask1, and when d_keep != NULL
 * every coefficient i with keep[(k % keep_period)*N + i] == 0 is zeroed in c0 (the reference
 * walks ct.data(0) only, so c1 keeps all its coefficients).
 * Masks: mmh_truncate_masks. */
int mmh_truncate_masks(const mmh_ctx *ctx, uint64_t *mask0, uint64_t *mask1);
/* flood_ciphertext's noise step of the USE_APPROX_RESHARE=0 build (hom_conv2d_ss.cc:358-380 with
 * set_poly_coeffs_uniform :142-177): first-level cts [n][2][L][N], c0[l] += (random & bitmask) mod q_l,
 * bitlen = min(64, bits(Q) - bits(t) - 1), bitmask = (1 << (bitlen % 64)) - 1 exactly as written at
 * :154 (so 0 when bitlen = 64).  d_random [n][N] raw words. */
int mmh_flood_add(const mmh_ctx *ctx, uint64_t *d_cts, const uint64_t *d_random, size_t n, void *stream);
/* modulo_poly_coeffs with the plain modulus (hom_conv2d_ss.cc:257-265, hom_bn_ss.cc:715-723) */
int mmh_reduce_mod_t(const mmh_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t count, void *stream);
int mmh_truncate_and_zero(const mmh_ctx *ctx, uint64_t *d_cts, size_t n, const uint8_t *d_keep,
                          size_t keep_period, void *stream);

/* seal::BatchEncoder::encode / decode on n vectors of N slots (hom_bn_ss.cc:179,225,354,390):
 * encode reduces inputs mod t first (ring_to_field outputs are only congruent). */
int mmh_batch_encode(const mmh_ctx *ctx, const uint64_t *d_vals, size_t n, uint64_t *d_pts, void *stream);
int mmh_batch_decode(const mmh_ctx *ctx, const uint64_t *d_pts, size_t n, uint64_t *d_vals, void *stream);
/* mmh_batch_decode without any allocation.  No counterpart in the reference: it is for callers that cannot
 * call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it).  The working
 * copy of the plaintexts lives in a static device array of 2^20 words (8 MiB per GPU, held until the process
 * ends) instead of the scratch cache, so n is limited to 2^20 / N plaintexts (256 at N = 4096;
 * mmh_batch_decode_noalloc_max reports it) and larger n fail with MMH_ERR_OUT_BOUND.  The context's device must
 * be current.  Calls on one GPU share the array: calls that overlap on one GPU race.  Same results as
 * mmh_batch_decode. */
int mmh_batch_decode_noalloc(const mmh_ctx *ctx, const uint64_t *d_pts, size_t n, uint64_t *d_vals, void *stream);
int mmh_batch_decode_noalloc_max(const mmh_ctx *ctx, size_t *max_n);

/* =====================================================================================
 * Randomness on the device.  The reference samples through SEAL's UniformRandomGenerator
 * (Blake2xb): parms.random_generator()->create() at hom_conv2d_ss.cc:759-760, hom_bn_ss.cc:329.
 * Here a ChaCha20 keystream (RFC 8439 block function; block counter in words 12-13, stream id in
 * words 14-15) keyed by the caller feeds the same three distributions.  Streams are not
 * comparable with SEAL's, only the distributions are; parity tests inject explicit randomness.
 * ===================================================================================== */
/* A fresh symmetric ciphertext is (c0, c1 = a) with `a` uniform: the reference ships it as c0 plus the PRNG seed
 * of `a` (seal::Serializable<Ciphertext> from encrypt_symmetric, saved by send_ciphertext, cheetah-api.cpp:11-29 on
 * the output of encryptImage :401-406) and the receiver expands the seed again.  These two move the halves of
 * [n][2][limbs(lv)][N] apart and back together; the expansion itself is mmh_sample_uniform with the shipped key. */
int mmh_ct_split_c0(const mmh_ctx *ctx, int lv, const uint64_t *d_cts, size_t n, uint64_t *d_c0, void *stream);
int mmh_ct_join(const mmh_ctx *ctx, int lv, const uint64_t *d_c0, const uint64_t *d_c1, size_t n, uint64_t *d_cts,
                void *stream);
/* raw keystream as little-endian 64-bit words, starting at block `counter0` */
int mmh_chacha20_words(const uint32_t key[8], uint64_t stream_id, uint64_t counter0, uint64_t *d_out,
                       size_t n_words, void *stream);
/* seal::util::sample_poly_ternary (hom_conv2d_ss.cc:302): n_coeffs values in {-1,0,1} */
int mmh_sample_ternary(const uint32_t key[8], uint64_t stream_id, int64_t *d_out, size_t n_coeffs, void *stream);
/* SEAL_NOISE_SAMPLER = sample_poly_cbd (hom_conv2d_ss.cc:329): popcount(21 bits) - popcount(21 bits) */
int mmh_sample_noise(const uint32_t key[8], uint64_t stream_id, int64_t *d_out, size_t n_coeffs, void *stream);
/* seal::util::sample_poly_uniform (hom_conv2d_ss.cc:206): [n_polys][limbs(lv)][N], limb l < q_l */
int mmh_sample_uniform(const mmh_ctx *ctx, int lv, const uint32_t key[8], uint64_t stream_id, uint64_t *d_out,
                       size_t n_polys, void *stream);
/* The randomness of n re-sharings in ONE launch (conv2DSS / matVecMul draw it per output ciphertext,
 * hom_conv2d_ss.cc:275-340,:180-223): d_u = mmh_sample_ternary(key, stream_id), d_e1 = mmh_sample_noise(key,
 * stream_id + 1), d_mask = mmh_sample_uniform(ctx, 1, key, stream_id + 2), each [n][N]; the same values as the three
 * calls (tests/test_sampler_gpu.py).  The caller advances its stream id by 3. */
int mmh_sample_reshare(const mmh_ctx *ctx, const uint32_t key[8], uint64_t stream_id, size_t n, int64_t *d_u,
                       int64_t *d_e1, uint64_t *d_mask, void *stream);
/* Polynomials [first_poly, first_poly + n) of the same draw, each array [n][N]: the rows mmh_sample_reshare would
 * write at first_poly ...  A layer cut over several devices (LaunchWorks' ranges of output ciphertexts,
 * hom_conv2d_ss.cc:597-607) draws each range where it is used and puts the bytes of the one-device run on the wire. */
int mmh_sample_reshare_range(const mmh_ctx *ctx, const uint32_t key[8], uint64_t stream_id, size_t first_poly, size_t n,
                             int64_t *d_u, int64_t *d_e1, uint64_t *d_mask, void *stream);
/* signed small polynomials [n][N] -> RNS + forward NTT [n][limbs][N]: secret key to NTT form
 * (seal::KeyGenerator, cheetah-api.cpp:109-111) */
int mmh_signed_to_ntt(const mmh_ctx *ctx, int lv, const int64_t *d_signed, size_t n, uint64_t *d_out,
                      void *stream);

/* =====================================================================================
 * HomConv2DSS (include/gemini/cheetah/hom_conv2d_ss.h:27-115) as layer-level calls.
 * A plan fixes the geometry of one layer (HomConv2DSS::Meta: ishape, fshape, n_filters,
 * padding, stride, is_shared_input) and owns the device index tables derived from
 * shape_inference::Conv2D, Conv2DSliceHelper and ConvCoeffIndexCalculator.
 * Tensors are row-major uint64: image C x H x W, filters M x C x fh x fw, result M x oh x ow.
 * ===================================================================================== */
typedef struct mmh_conv_meta {
  int64_t in_c, in_h, in_w;   /* meta.ishape */
  int64_t f_h, f_w;           /* meta.fshape (channels = in_c) */
  int64_t n_filters;
  int32_t padding_same;       /* gemini::Padding: 0 VALID, 1 SAME */
  int32_t stride;
  int32_t is_shared_input;
} mmh_conv_meta;
typedef struct mmh_conv_plan mmh_conv_plan;
int mmh_conv_plan_create(const mmh_ctx *ctx, const mmh_conv_meta *meta, mmh_conv_plan **out);
int mmh_conv_plan_destroy(mmh_conv_plan *plan);
/* counts a caller needs to size buffers: input ciphertexts (= channel slices x spatial slices),
 * output ciphertexts (= n_filters x spatial slices), result shape, slice_strides */
int mmh_conv_plan_info(const mmh_conv_plan *plan, size_t *n_in_cts, size_t *n_channel_slices,
                       size_t *n_spatial_slices, size_t *n_out_cts, int64_t out_shape[3],
                       int64_t slice_strides[3]);
/* What one server call of the plan's layer does, for a caller that prices it against the machine (bench.py's
 * roofline of the network legs): the route the plan's cost model picked (1 = direct product of the rotated input
 * columns with the weight matrix on the matrix cores, 0 = transform route) for filters that arrive with the call and
 * for prepared ones, the taps K = C*fh*fw of a filter and K rounded up to the product kernel's 64-tap step, the
 * coefficient columns the direct product computes per limb (all of c1, of c0 the 64-column tiles that hold a
 * coefficient removeUnusedCoeffs keeps; 0 when the plan cannot take the direct route), and the counts of
 * mmh_conv_plan_info.  No counterpart in the reference (its schedule is fixed: conv2DOneFilter). */
typedef struct mmh_conv_work {
  int32_t direct_fresh, direct_prepared;
  int64_t taps, taps_padded;
  int64_t live_columns[2];
  int64_t n_in, n_out, n_cslices, n_spatial, n_targets;
  int64_t limbs, poly_degree;
} mmh_conv_work;
int mmh_conv_plan_work(const mmh_conv_plan *plan, mmh_conv_work *work);
/* TensorEncoder::EncodeImageShare (tensor_encoder.cc:178-244): image -> n_in plaintexts [n_in][N] */
int mmh_conv_encode_image(const mmh_conv_plan *plan, const uint64_t *d_img, uint64_t *d_pts, void *stream);
/* HomConv2DSS::encodeFilters -> TensorEncoder::EncodeFilter (hom_conv2d_ss.cc:537-561,
 * tensor_encoder.cc:124-176): filters -> [n_filters][n_channel_slices][N] */
int mmh_conv_encode_filters(const mmh_conv_plan *plan, const uint64_t *d_filters, uint64_t *d_pts, void *stream);
/* HomConv2DSS::encryptImage (hom_conv2d_ss.cc:497-522) with injected randomness:
 * d_a [n_in][L][N] uniform, d_e [n_in][N] signed -> d_cts [n_in][2][L][N] */
int mmh_conv_encrypt_image(const mmh_conv_plan *plan, const uint64_t *d_sk_ntt, const uint64_t *d_img,
                           const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream);
/* HomConv2DSS::conv2DSS (hom_conv2d_ss.cc:621-712) in the shipped USE_APPROX_RESHARE build:
 * add_plain of the server's share (when is_shared_input; d_share_pts may be NULL otherwise),
 * per-filter multiply-accumulate, addRandomMask (flood_ciphertext + approximate
 * sample_random_mask + sub_poly), truncate_for_decryption, removeUnusedCoeffs.
 * d_pk_ntt_last: public key restricted to the last level [2][1][N] (NTT form).
 * Randomness per output ciphertext: d_u ternary [n_out][N], d_e1 noise [n_out][N],
 * d_mask uniform in [0,q0) [n_out][N].
 * Outputs: d_out_cts [n_out][2][1][N], d_out_share result tensor M x oh x ow (fully written). */
int mmh_conv_server(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                    const uint64_t *d_filter_pts, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                    const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share,
                    void *stream);
/* conv2DSS as compiled with USE_APPROX_RESHARE=0 (CMakeLists.txt:31-35; scripts/build.sh ships =ON):
 * flood_ciphertext first adds (d_flood & bitmask) mod q_l to c0 (hom_conv2d_ss.cc:358-380, d_flood
 * [n_out][N] raw 64-bit words), the zero encryption keeps e0 (:327), and sample_random_mask :225-272
 * draws the mask mod t (d_mask [n_out][N] raw words, reduced here), subtracts it with sub_plain and
 * returns it as the share.  Everything else as mmh_conv_server. */
int mmh_conv_server_exact(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                          const uint64_t *d_filter_pts, const uint64_t *d_pk_ntt_last, const uint64_t *d_flood,
                          const int64_t *d_u, const int64_t *d_e0, const int64_t *d_e1, const uint64_t *d_mask,
                          uint64_t *d_out_cts, uint64_t *d_out_share, void *stream);
/* The server half of CheetahLinear::conv2d in one call (cheetah-api.cpp:417-445: encodeFilters
 * followed by conv2DSS): as mmh_conv_server but from the raw filters d_filters [M][C][fh][fw].
 * Each filter polynomial is encoded, lifted and transformed on chip, so the sparse plaintexts of
 * mmh_conv_encode_filters are never written; results are bit-identical to the two-call form.
 * The caller keeps the conv2DOneFilter :573-580 rule (a filter that is all zero mod t is an error). */
int mmh_conv_server_weights(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                            const uint64_t *d_filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                            const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                            uint64_t *d_out_share, void *stream);
/* The whole server half of CheetahLinear::conv2d (cheetah-api.cpp:401-445: encodeImage of the server's share,
 * encodeFilters, conv2DSS) in one call: as mmh_conv_server_weights, but the server's share arrives as the raw
 * tensor d_share [C][H][W] (ignored when the plan's input is not shared) and is encoded inside the pass that adds
 * it to the ciphertexts; no share plaintexts are written.  Bit-identical to mmh_conv_encode_image followed by
 * mmh_conv_server_weights.  d_zero_filter (device word, may be null): when given, the call also evaluates the
 * conv2DOneFilter :573-580 rule as mmh_conv_check_filters does (1 when some filter is all zero mod t, else 0);
 * the caller reads it back after enqueuing and treats 1 as the reference's error. */
int mmh_conv_server_tensors(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share,
                            const uint64_t *d_filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                            const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                            uint64_t *d_out_share, uint32_t *d_zero_filter, void *stream);
/* mmh_conv_server_tensors whose zero-filter verdict can be read EARLY: `zero_filter` may point into page-locked host
 * memory (mmh_host_alloc, torch pinned memory: the kernels store to it directly, no atomics), and `verdict_event`
 * (mmh_event_create; may be NULL) is recorded on the stream right behind the launch that settles the word -- the second
 * launch of the call on the direct route -- so the caller waits for that event alone, reads the word and raises the
 * reference's error (conv2DOneFilter :573-580 returns before it computes) while the products and the re-sharing of
 * this call, and whatever it enqueued after it, are still running.  No extra launch, clear or copy for the verdict. */
int mmh_conv_server_tensors_ev(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share,
                               const uint64_t *d_filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                               const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                               uint64_t *d_out_share, uint32_t *zero_filter, void *verdict_event, void *stream);
/* mmh_conv_server_tensors_ev that DRAWS the re-sharing's randomness itself, as conv2DSS does inside the call
 * (sample_poly_ternary :302, SEAL_NOISE_SAMPLER :329, sample_poly_uniform :206): u, e1 and the mask are polynomials
 * [first_poly, first_poly + n_out) of the three streams mmh_sample_reshare(ctx, key, stream_id, ...) draws, so the call
 * equals mmh_sample_reshare_range + mmh_conv_server_tensors_ev bit for bit -- without the sampling launch and without
 * u in memory: on the direct route the call's front launch draws u where it transforms it and e1 / the mask beside
 * its other work, and for a layer of few output ciphertexts it finishes the pk-encryption of zero there as well. */
int mmh_conv_server_tensors_sampled(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share,
                                    const uint64_t *d_filters, const uint64_t *d_pk_ntt_last, const uint32_t key[8],
                                    uint64_t stream_id, size_t first_poly, uint64_t *d_out_cts, uint64_t *d_out_share,
                                    uint32_t *zero_filter, void *verdict_event, void *stream);
/* The conv2DOneFilter :573-580 rule on the raw filters [M][C][fh][fw]: *d_flag (device word) becomes 1
 * when some filter is all zero mod t, else 0.  Asynchronous, so a caller can read the verdict back after
 * it has enqueued the layer. */
int mmh_conv_check_filters(const mmh_conv_plan *plan, const uint64_t *d_filters, uint32_t *d_flag, void *stream);
/* Static weights prepared once per model (the reference re-encodes the filters
 * on every call, cheetah-api.cpp:418-423, which mmh_conv_server_weights reproduces).  The handle keeps
 * what the layer call needs of the filters in device memory -- lifted NTT-form filter polynomials, or the
 * scalar weight matrix for 1x1 layers -- so mmh_conv_server_prepared skips their encoding.  Results are
 * bit-identical to mmh_conv_server_weights on the same filters.  The handle belongs to the plan it was
 * made for; destroy it before the plan. */
typedef struct mmh_conv_filters mmh_conv_filters;
int mmh_conv_filters_prepare(const mmh_conv_plan *plan, const uint64_t *d_filters, mmh_conv_filters **out,
                             void *stream);
int mmh_conv_filters_destroy(mmh_conv_filters *filters);
int mmh_conv_server_prepared(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                             const mmh_conv_filters *filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                             const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                             uint64_t *d_out_share, void *stream);
/* The re-sharing tail split where the layer's input enters it.  addRandomMask's pk-encryption of zero
 * (asymmetric_encrypt_zero, hom_conv2d_ss.cc:275-340, called at :769) depends on the public key and on fresh
 * randomness only: mmh_reshare_zero_prepare makes n of them ahead of the inputs -- d_zero [n][2][N], the products
 * u * pk_j in coefficient form at the last level, without noise (e1 enters in the online pass; e0 is skipped by the
 * USE_APPROX_RESHARE build, :323-328) -- and mmh_conv_server_online is mmh_conv_server_prepared with that part taken
 * from d_zero (n = the plan's n_out): one pass instead of two transforms per output ciphertext, same bits as
 * mmh_conv_server_prepared with the same u.  The reference computes the encryption inside conv2DSS on every call;
 * a deployment that has idle time before the client's ciphertexts arrive moves it there.  Every d_zero must be used
 * once (it is the ciphertext's randomness).  Exactly one of d_share_pts (encoded share, as _prepared takes it) and
 * d_share_raw (the share tensor [C][H][W], as _tensors takes it) for a shared input, neither otherwise. */
int mmh_reshare_zero_prepare(const mmh_ctx *ctx, const uint64_t *d_pk_ntt_last, const int64_t *d_u, size_t n,
                             uint64_t *d_zero, void *stream);
int mmh_conv_server_online(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                           const uint64_t *d_share_raw, const mmh_conv_filters *filters, const uint64_t *d_zero,
                           const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share,
                           void *stream);
/* mmh_conv_server_prepared with the server's share as the raw tensor [C][H][W] (encodeImage, hom_conv2d_ss.cc:524-545,
 * fused into the call the way mmh_conv_server_tensors does it); same bits. */
int mmh_conv_server_prepared_tensors(const mmh_conv_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share,
                                     const mmh_conv_filters *filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                                     const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                                     uint64_t *d_out_share, void *stream);
/* Compact wire form of conv2DSS's result ciphertexts [n_out][2][N] (last level).  The reference sends them through
 * seal::Ciphertext::save (cheetah-api.cpp:11-29, :443-444), whose compression squeezes out what
 * truncate_for_decryption (hom_conv2d_ss.cc:418-436) and remove_unused_coeffs (:800-851) zeroed; this pair does it
 * by construction: per ciphertext the N coefficients of c1 without their cleared low bits, then the kept
 * coefficients of c0 likewise, as two bit streams padded to whole words (words_per_ct in all, the same for every
 * ciphertext of the plan).  unpack(pack(ct)) == ct for every ciphertext the server calls produce. */
int mmh_conv_result_packed_words(const mmh_conv_plan *plan, size_t *words_per_ct);
int mmh_conv_result_pack(const mmh_conv_plan *plan, const uint64_t *d_cts, uint64_t *d_packed, void *stream);
int mmh_conv_result_unpack(const mmh_conv_plan *plan, const uint64_t *d_packed, uint64_t *d_cts, void *stream);
/* HomConv2DSS::decryptToTensor (hom_conv2d_ss.cc:853-941): d_sk_ntt at the last level [1][N] */
int mmh_conv_decrypt(const mmh_conv_plan *plan, const uint64_t *d_sk_ntt, const uint64_t *d_cts,
                     uint64_t *d_out, void *stream);

/* =====================================================================================
 * HomFCSS (include/gemini/cheetah/hom_fc_ss.h:27-121): out = W * in for a rows x cols weight
 * matrix (row-major) and a length-cols input vector, Meta {weight_shape, input_shape,
 * is_shared_input}.
 * ===================================================================================== */
typedef struct mmh_fc_meta {
  int64_t rows, cols;       /* meta.weight_shape; meta.input_shape = cols */
  int32_t is_shared_input;
} mmh_fc_meta;
typedef struct mmh_fc_plan mmh_fc_plan;
int mmh_fc_plan_create(const mmh_ctx *ctx, const mmh_fc_meta *meta, mmh_fc_plan **out);
/* The same with (d0, d1) given instead of taken from getSplit: the plan of the row blocks [b0, b1) of a larger
 * matrix, whose split the sub-matrix (rows = those blocks' rows) has to keep (matVecMul's workers each take a range of
 * row blocks, hom_fc_ss.cc:443-457).  1 <= d0, d1 and d0 * d1 <= N. */
int mmh_fc_plan_create_split(const mmh_ctx *ctx, const mmh_fc_meta *meta, const int64_t split[2], mmh_fc_plan **out);
int mmh_fc_plan_destroy(mmh_fc_plan *plan);
/* split = getSplit's (d0, d1) (hom_fc_ss.cc:25-50); n_in = ceil(cols/d1), n_out = ceil(rows/d0) */
int mmh_fc_plan_info(const mmh_fc_plan *plan, int64_t split[2], size_t *n_in_cts, size_t *n_out_cts);
/* getSplit alone (hom_fc_ss.cc:25-50), host only, no device needed: the block (d0, d1) the plan would use.  The
 * search is pruned (per d0 only the smallest d1 reaching that row's minimum is examined); tests sweep it against the
 * reference's exhaustive double loop. */
int mmh_fc_split(int64_t rows, int64_t cols, size_t N, int64_t split[2]);
/* HomFCSS::encodeInputVector (hom_fc_ss.cc:275-325): -> [n_in][N] */
int mmh_fc_encode_input(const mmh_fc_plan *plan, const uint64_t *d_vec, uint64_t *d_pts, void *stream);
/* HomFCSS::encodeWeightMatrix (hom_fc_ss.cc:327-392): -> [n_out][n_in][N] */
int mmh_fc_encode_weights(const mmh_fc_plan *plan, const uint64_t *d_w, uint64_t *d_pts, void *stream);
/* HomFCSS::encryptInputVector (hom_fc_ss.cc:214-273) with injected randomness */
int mmh_fc_encrypt_input(const mmh_fc_plan *plan, const uint64_t *d_sk_ntt, const uint64_t *d_vec,
                         const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream);
/* HomFCSS::matVecMul (hom_fc_ss.cc:394-471), USE_APPROX_RESHARE build; arguments as mmh_conv_server,
 * randomness per output row block [n_out][N]; d_out_share has `rows` entries */
int mmh_fc_server(const mmh_fc_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                  const uint64_t *d_weight_pts, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                  const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share,
                  void *stream);
/* matVecMul as compiled with USE_APPROX_RESHARE=0; extra arguments as mmh_conv_server_exact */
int mmh_fc_server_exact(const mmh_fc_plan *plan, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                        const uint64_t *d_weight_pts, const uint64_t *d_pk_ntt_last, const uint64_t *d_flood,
                        const int64_t *d_u, const int64_t *d_e0, const int64_t *d_e1, const uint64_t *d_mask,
                        uint64_t *d_out_cts, uint64_t *d_out_share, void *stream);
/* HomFCSS::decryptToVector (hom_fc_ss.cc:572-602) */
int mmh_fc_decrypt(const mmh_fc_plan *plan, const uint64_t *d_sk_ntt, const uint64_t *d_cts, uint64_t *d_out,
                   void *stream);

/* =====================================================================================
 * HomBNSS, SIMD path = BOLE (include/gemini/cheetah/hom_bn_ss.h:25-198; hom_bn_ss.cc:151-406).
 * The context's plain modulus must be a batching prime (setUp_Field, hom_bn_ss.cc:122-126).
 * A length-n vector occupies n_ct = ceil(n / min(n, N)) ciphertexts of N slots each.
 * ===================================================================================== */
/* HomBNSS::encodeVector / encodeScales (hom_bn_ss.cc:203-239): -> [n_ct][N] */
int mmh_bn_encode_vector(const mmh_ctx *ctx, const uint64_t *d_vec, size_t n, uint64_t *d_pts, void *stream);
/* HomBNSS::encryptVector (hom_bn_ss.cc:151-189) with injected randomness a [n_ct][L][N], e [n_ct][N] */
int mmh_bn_encrypt_vector(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_vec, size_t n,
                          const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream);
/* HomBNSS::bn (hom_bn_ss.cc:241-278) + addMask (:318-362): (cts [+ share_pts]) * scale_pts, mod-switch,
 * + encode(r), truncate.  d_random: n raw 64-bit words (r = word % p, :332-334); share = -r mod p.
 * d_share_pts NULL when !is_shared_input. */
int mmh_bn_server(const mmh_ctx *ctx, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                  const uint64_t *d_scale_pts, size_t n, const uint64_t *d_random, uint64_t *d_out_cts,
                  uint64_t *d_out_share, void *stream);
/* HomBNSS::bole (hom_bn_ss.cc:280-316) + addMask: out[i] = ex[i]*py[i] + ey[i]*px[i].
 * d_enc_xy = [ex ; ey] (2*n_ct ciphertexts), d_pts_yx = [py ; px] (2*n_ct plaintexts).
 * The reference destroys ex/ey in place (:304-305); here the inputs are left untouched. */
int mmh_bole_server(const mmh_ctx *ctx, const uint64_t *d_enc_xy, const uint64_t *d_pts_yx, size_t n,
                    const uint64_t *d_random, uint64_t *d_out_cts, uint64_t *d_out_share, void *stream);
/* HomBNSS::decryptToVector (hom_bn_ss.cc:364-406): last-level cts -> n values */
int mmh_bn_decrypt_vector(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_cts, size_t n,
                          uint64_t *d_out, void *stream);

/* -------------------------------------------------------------------------------------
 * HomBNSS direct path (hom_bn_ss.cc:471-790, the BatchNorm() wrapper of
 * library_fixed_uniform_cheetah.cpp:465): a C x H x W tensor mod t = 2^k is coefficient-packed in
 * floor(sqrt(N))-square tiles, one channel per tile (getSplitBN :61-67), and every channel is
 * multiplied by a signed scalar.  Ciphertext cid = (c*dH + hb)*dW + wb.
 * ------------------------------------------------------------------------------------- */
typedef struct mmh_bn_tensor_meta {
  int64_t c, h, w;          /* HomBNSS::Meta::ishape */
  int32_t is_shared_input;
} mmh_bn_tensor_meta;
/* number of plaintexts / ciphertexts dC*dH*dW and the tile shape {1, sh, sw} */
int mmh_bn_tensor_count(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, size_t *n_cts, int64_t split[3]);
/* HomBNSS::encodeTensor (:471-520): entry (h, w) of a tile at coefficient h*sw + w, zero outside */
int mmh_bn_encode_tensor(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_tensor,
                         uint64_t *d_pts, void *stream);
/* HomBNSS::encryptTensor (:522-576) with injected randomness d_a [n][L][N], d_e [n][N] */
int mmh_bn_encrypt_tensor(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_sk_ntt,
                          const uint64_t *d_tensor, const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts,
                          void *stream);
/* HomBNSS::bn_direct (:578-738): add_plain of the server's share (d_share_pts NULL unless
 * is_shared_input), multiply_poly_scalar_coeffmod by (+/-)(|scales[c]| mod t) (scales read as
 * int64), mod-switch, sub_plain of encodeTensor(d_share1), add Encryptor::encrypt_zero at the last
 * level (a first-level pk encryption with u, e0, e1 [n][N] that is mod-switched; d_pk_ntt is the
 * full key [2][L][N]), remove_unused_coeffs on the first vh*vw coefficients, truncate_for_decryption.
 * d_share1 [C][H][W] is the server's output share, already uniform mod t (the reference draws it
 * at :712-723); outputs d_out_cts [n][2][1][N]. */
int mmh_bn_direct_server(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_in_cts,
                         const uint64_t *d_share_pts, const uint64_t *d_scales, const uint64_t *d_pk_ntt,
                         const int64_t *d_u, const int64_t *d_e0, const int64_t *d_e1, const uint64_t *d_share1,
                         uint64_t *d_out_cts, void *stream);
/* HomBNSS::decryptToTensor (:740-790): valid entries of each tile from consecutive coefficients */
int mmh_bn_decrypt_tensor(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_sk_ntt,
                          const uint64_t *d_cts, uint64_t *d_out, void *stream);

/* =====================================================================================
 * Modulus conversion: local share arithmetic of SCI's Truncation class
 * (SCI/src/BuildingBlocks/truncation.h:29-210), split around its single COT call.
 * party: 1 = ALICE (the COT sender, truncation.cpp:549,607,665,351,412), 2 = BOB.
 *   *_pre : d_tmp = input (+ "big number" for ALICE), d_msb = choice / wrap bit per element,
 *           d_corr = sender correlation (ALICE only; may be NULL for BOB).
 *   host  : ALICE send_cot[_prime](y, corr), BOB recv_cot[_prime](y, msb)  -- stays on host.
 *   *_post: consumes y exactly as the reference does after the COT returns.
 * No context is needed; work is enqueued on the current device.  in/out never alias-modified.
 * ===================================================================================== */
/* Truncation::ring_to_field (truncation.cpp:518-581); outputs are congruent mod p, not
 * canonical, exactly like the reference's secure_sub on l-bit inputs (:538-541). */
int mmh_r2f_pre(int party, const uint64_t *d_in, size_t n, int bw, uint64_t p, uint64_t *d_tmp,
                uint8_t *d_msb, uint64_t *d_corr, void *stream);
int mmh_r2f_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y, size_t n,
                 int bw, uint64_t p, uint64_t *d_out, void *stream);
/* Truncation::field_to_ring (:583-637) and field_to_ring_with_truncate (:639-695); the pre
 * step is shared, `shift` only enters the correlation p >> shift (0 for the plain variant) */
int mmh_f2r_pre(int party, const uint64_t *d_in, size_t n, uint64_t p, int shift, uint64_t *d_tmp,
                uint8_t *d_msb, uint64_t *d_corr, void *stream);
int mmh_f2r_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y, size_t n,
                 uint64_t p, int bw, uint64_t *d_out, void *stream);
int mmh_f2r_trunc_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y, size_t n,
                       uint64_t p, int bw, int shift, uint64_t *d_out, void *stream);
/* Conversions fused with the HomBNSS vector step next to them.  CheetahLinear::bn / ::BOLE
 * convert their shares with ring_to_field, hand the field elements to encodeVector / encryptVector, and convert
 * decryptToVector's output back with field_to_ring[_with_truncate] (cheetah-api.cpp:560-660,
 * library_fixed_uniform_cheetah.cpp:492-565).  With everything resident the element need not travel through memory
 * between the two: _r2f = mmh_r2f_post (p = the context's field modulus t) + mmh_bn_encode_vector / _encrypt_vector,
 * _f2r = mmh_bn_decrypt_vector + mmh_f2r_pre (shift = 0 for the plain variant); same bits as the calls they replace,
 * 25 instead of 57 bytes of traffic per element on the encode side. */
int mmh_bn_encode_vector_r2f(const mmh_ctx *ctx, int party, const uint64_t *d_tmp, const uint8_t *d_msb,
                             const uint64_t *d_y, size_t n, int bw, uint64_t *d_pts, void *stream);
int mmh_bn_encrypt_vector_r2f(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, int party, const uint64_t *d_tmp,
                              const uint8_t *d_msb, const uint64_t *d_y, size_t n, int bw, const uint64_t *d_a,
                              const int64_t *d_e, uint64_t *d_cts, void *stream);
int mmh_bn_decrypt_vector_f2r(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_cts, size_t n, int party,
                              int shift, uint64_t *d_tmp, uint8_t *d_msb, uint64_t *d_corr, void *stream);
/* Truncation::ring_to_ring (:319-381) and ring_to_ring128 (:383-442; d_out holds (lo,hi) pairs) */
int mmh_r2r_pre(int party, const uint64_t *d_in, size_t n, int bw_a, uint64_t *d_tmp, uint8_t *d_msb,
                uint64_t *d_corr, void *stream);
int mmh_r2r_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y, size_t n,
                 int bw_a, int bw_b, uint64_t *d_out, void *stream);
int mmh_r2r128_post(int party, const uint64_t *d_tmp, const uint8_t *d_msb, const uint64_t *d_y, size_t n,
                    int bw_a, int bw_b, uint64_t *d_out, void *stream);
/* Local share arithmetic of Truncation::truncate / truncate_msb / truncate_msb0 in the shipped USE_CHEETAH build
 * (SCI/src/BuildingBlocks/truncation-cheetah.cpp:2-218) around their wrap and B2A sub-protocols (OT-based, host).
 * pre: d_adj = the input after the msb0 heuristic's +2^(bw-3) on BOB (:23-29) and the signed case's +2^(bw-1)
 * on ALICE (:42-47, :102-107, :183-188); d_cmp (may be null) = the operand of mill->compare (:51-56).
 * post: d_out from d_adj and the B2A output d_aw (:64-75, :133-144, :201-212, heuristic :32-34). */
int mmh_trunc_pre(int party, const uint64_t *d_in, size_t n, int bw, int signed_arith, int heuristic,
                  uint64_t *d_adj, uint64_t *d_cmp, void *stream);
int mmh_trunc_post(int party, const uint64_t *d_adj, const uint64_t *d_aw, size_t n, int shift, int bw,
                   int signed_arith, int heuristic, uint64_t *d_out, void *stream);
/* Truncation::truncate_and_reduce (truncation.cpp:289-317): the operand of wrap_computation (:300-304) and the
 * result in the ring of bw - shift bits from the B2A output d_aw (:311-313) */
int mmh_trunc_reduce_pre(const uint64_t *d_in, size_t n, int shift, uint64_t *d_lower, void *stream);
int mmh_trunc_reduce_post(const uint64_t *d_in, const uint64_t *d_aw, size_t n, int shift, int bw,
                          uint64_t *d_out, void *stream);
/* Local share arithmetic of ReLURingProtocol::relu in the USE_CHEETAH build (SCI/src/NonLinear/relu-ring.h):
 * pre (:216-238) the msb of the share and the operand of millionaire->compare on l-1 bits; post (:244-252,
 * :308-311) the DReLU share (msb + wrap) % 2 and the selector handed to aux->multiplexer (ALICE flips hers). */
int mmh_relu_pre(int party, const uint64_t *d_share, size_t n, int l, uint8_t *d_msb, uint64_t *d_cmp, void *stream);
int mmh_relu_post(int party, const uint8_t *d_msb, const uint8_t *d_wrap, size_t n, uint8_t *d_drelu,
                  uint8_t *d_sel, void *stream);
/* AuxProtocols::multiplexer (SCI/src/BuildingBlocks/aux-protocols.cpp:62-93) around its two COTs: the correlation
 * x * (1 - 2 sel) (:76-78) and y = x * sel + data_R - data_S (:86-88), mod 2^bw_y */
int mmh_mux_pre(const uint8_t *d_sel, const uint64_t *d_x, size_t n, int bw_y, uint64_t *d_corr, void *stream);
int mmh_mux_post(const uint8_t *d_sel, const uint64_t *d_x, const uint64_t *d_data_r, const uint64_t *d_data_s,
                 size_t n, int bw_y, uint64_t *d_y, void *stream);
/* AuxProtocols::B2A (aux-protocols.cpp:95-126) around its COT: ALICE's correlation -2x (:107-109) and the outputs
 * x - y (ALICE, :112-114) / x + y (BOB, :119-121) mod 2^bw_y; for bw_y = 1 the bit itself (:97-102, d_y unused) */
int mmh_b2a_pre(const uint8_t *d_x, size_t n, int bw_y, uint64_t *d_corr, void *stream);
int mmh_b2a_post(int party, const uint8_t *d_x, const uint64_t *d_y, size_t n, int bw_y, uint64_t *d_out,
                 void *stream);
/* out[i] = (out[i] + moduloMult(x[i], y[i], p)) % p : the local cross term that closes
 * CheetahLinear::BOLE (cheetah-api.cpp:648-649; sci::moduloMult utils.hpp:353-363) */
int mmh_bole_cross_term(uint64_t *d_out, const uint64_t *d_x, const uint64_t *d_y, size_t n, uint64_t p,
                        void *stream);

/* Layout changes of the layer adapters on the device (SCI/src/library_fixed_uniform_cheetah.cpp:236-262 filters
 * [FH][FW][CI][CO] -> per filter [CI][FH][FW]; :284-287 input [H][W][CI] -> [CI][H][W]; :293-299 output back), with
 * getRingElt's reduction `& ring_mask` on the way in.  Out of place. */
int mmh_repack_nhwc_to_chw(const uint64_t *d_in, size_t h, size_t w, size_t c, uint64_t ring_mask, uint64_t *d_out,
                           void *stream);
int mmh_repack_chw_to_nhwc(const uint64_t *d_in, size_t c, size_t h, size_t w, uint64_t *d_out, void *stream);
int mmh_repack_hwio_to_oihw(const uint64_t *d_in, size_t fh, size_t fw, size_t ci, size_t co, uint64_t ring_mask,
                            uint64_t *d_out, void *stream);

/* FusedBN of the shipped networks (USE_FUSED_BN 1; networks/main_resnet50.cpp:1728-1788): the server
 * folds a batch-norm scale into the filters before Conv2DWrapper: scaled = neg_mod(round(f/2^k * s/2^k * 2^k), 2^l)
 * in double arithmetic with getSignValue (:21-28) on both operands -- the only floating point on the path, evaluated
 * in the reference's order so the bits agree.  d_filters [fh][fw][ci][co], d_scales [co]; d_out in the same layout
 * (oihw = 0, what FusedBN hands to Conv2DWrapper) or as [co][ci][fh][fw] (oihw = 1, what mmh_conv_server_tensors /
 * mmh_conv_filters_prepare read).  *d_any_nonzero (device word) becomes 1 unless every scaled entry is zero, the
 * case in which the reference warns and exits (:1754-1762); the caller reads it back.  Products beyond int64_t are
 * undefined in the reference (static_cast of the double) and unspecified here. */
int mmh_fused_bn_filters(const uint64_t *d_filters, const uint64_t *d_scales, int fh, int fw, int ci, int co,
                         int bitlength, int k_scale, uint64_t prime_mod, int oihw, uint64_t *d_out,
                         uint32_t *d_any_nonzero, void *stream);
/* the bias the server adds to its output share afterwards (:1771-1786): out[pixel][m] += bias[m] << k (ScaleUp1 in the
 * SCI_OT build, library_fixed_uniform.cpp:1394-1402; SecretAdd :1861-1867), NHWC */
int mmh_fused_bn_bias(uint64_t *d_out_nhwc, const uint64_t *d_bias, size_t pixels, int co, int k_scale, void *stream);

/* ---- COT post-processing: what SilentOT does between Ferret's output and the values the
 * conversions above consume.  One launch per side: MITCCRH hash of the Ferret blocks (AES-128 under one fresh
 * key per element, key_j = seed ^ makeBlock(gid0 + j, 0); emp-tool mitccrh.h at 44b1dde, restated -- that
 * dependency is absent from the reference tree, so the hash is pinned by FIPS-197 only), then
 *   prime = 0: SilentOT::send_ot_cam_cc / recv_ot_cam_cc (SCI/src/OT/ferret/silent_ot.h:85-127,:168-207):
 *              low 64 bits & (2^l - 1), correction, sci::pack_cot_messages / unpack_cot_messages
 *              (SCI/src/OT/ot-utils.h:154-216) per group of ot_bsize = 8 elements;
 *   prime = 1: send_ot_cam_cc_prime / recv_ot_cam_cc_prime (silent_ot.h:129-163,:209-242): 128-bit hash mod p,
 *              correction mod p, ceil(log2 p / 8) bytes per element on the wire.
 * d_rcm holds n 16-byte blocks in memory order (send_ot_rcm_cc / recv_ot_rcm_cc output, which stay on the
 * host with Ferret), delta = ferret->Delta, seed = the block `s` the sender draws and sends (silent_ot.h:91-94),
 * gid0 = the hash's key counter when the call starts; a call advances it by mmh_cot_gids(n) = 8*ceil(n/8)
 * (every hash<8,H> call renews eight keys).  d_wire: mmh_cot_wire_bytes bytes, 8-byte aligned, exactly the
 * bytes send_data / recv_data carry after the seed block.  l is ignored when prime = 1, p when prime = 0. */
int mmh_cot_wire_bytes(size_t n, int l, uint64_t p, int prime, size_t *bytes);
int mmh_cot_gids(size_t n, uint64_t *gids);
int mmh_cot_send(const uint64_t *d_rcm, const uint64_t delta[2], const uint64_t seed[2], uint64_t gid0,
                 const uint64_t *d_corr, size_t n, int l, uint64_t p, int prime, uint64_t *d_data0,
                 uint8_t *d_wire, void *stream);
int mmh_cot_recv(const uint64_t *d_rcm, const uint8_t *d_choice, const uint64_t seed[2], uint64_t gid0,
                 const uint8_t *d_wire, size_t n, int l, uint64_t p, int prime, uint64_t *d_data, void *stream);
/* A conversion of the COT sender's (ALICE of the Truncation object) in ONE launch: the pre step, the sender's
 * post-processing and the post step fused, so tmp / msb / corr / data0 never exist in memory.  Same outputs and the
 * same wire bytes as the calls they replace (ElemWiseActModelVectorMult runs both around CheetahLinear::bn,
 * library_fixed_uniform_cheetah.cpp:567-600):
 *   mmh_cot_send_f2r         Truncation::field_to_ring (shift < 0, truncation.cpp:583-637) / field_to_ring_with_truncate
 *                            (shift >= 0, :639-695) = mmh_f2r_pre + mmh_cot_send(prime = 0, l = bw) + mmh_f2r[_trunc]_post;
 *   mmh_cot_send_r2f_encode  Truncation::ring_to_field (:518-581) feeding HomBNSS::encodeVector (hom_bn_ss.cc:203-233)
 *                            = mmh_r2f_pre + mmh_cot_send(prime = 1, p = ctx's t) + mmh_bn_encode_vector_r2f: d_pts are the
 *                            plaintexts CheetahLinear::bn adds to the client's ciphertexts. */
int mmh_cot_send_f2r(const uint64_t *d_rcm, const uint64_t delta[2], const uint64_t seed[2], uint64_t gid0,
                     const uint64_t *d_share, size_t n, uint64_t p, int bw, int shift, uint64_t *d_out, uint8_t *d_wire,
                     void *stream);
int mmh_cot_send_r2f_encode(const mmh_ctx *ctx, const uint64_t *d_rcm, const uint64_t delta[2], const uint64_t seed[2],
                            uint64_t gid0, const uint64_t *d_x, size_t n, int bw, uint64_t *d_pts, uint8_t *d_wire,
                            void *stream);
/* ALICE's (the model owner's) side of one batch-norm layer, ElemWiseActModelVectorMult around CheetahLinear::bn
 * (library_fixed_uniform_cheetah.cpp:567-600, cheetah-api.cpp:621-660, hom_bn_ss.cc:236-276 and :333-357), as two calls
 * that meet at the client's ciphertexts, with the launches of the step-by-step entry points merged (13 -> 5):
 *   mmh_bn_alice_prepare  everything that does not need the client's ciphertexts, two launches.  The first runs side by
 *                         side: Ring2Field of her ring share d_x (n elements, bw bits) as the COT sender into
 *                         encodeVector's plaintexts (= mmh_cot_send_r2f_encode: blocks d_rcm_r2f, seed_r2f, key counter
 *                         gid0, wire d_wire_r2f); HomBNSS::encodeVector of the scales d_scales (n field elements);
 *                         addMask's r drawn on the device -- element e gets word e of mmh_chacha20_words(key,
 *                         stream_id, 0, ...), r = word mod p, her share -r mod p (stored to d_out_share unless null) --
 *                         and, on that share, Field2Ring (shift < 0) or Field2Ring_with_Truncate as the COT sender
 *                         (= mmh_cot_send_f2r: blocks d_rcm_f2r, seed_f2r, key counter gid0 + mmh_cot_gids(n), wire
 *                         d_wire_f2r, her ring share of the layer's result in d_out_ring): her share does not depend on
 *                         the ciphertexts, so neither does its conversion.  The second is BatchEncoder's inverse
 *                         transform of the three plaintext batches d_pts3 [3][n_ct][N] (her share, the scales, r), n_ct =
 *                         ceil(n / min(n, N)).  A call advances the key counter by 2 * mmh_cot_gids(n).
 *   mmh_bn_alice_finish   add_plain, multiply_plain by the scales, mod-switch to the last level, add_plain of r and
 *                         truncate_for_decryption on the client's ciphertexts d_in_cts [n_ct][2][2][N] into d_out_cts
 *                         [n_ct][2][N].  Three launches.
 * Bit for bit mmh_cot_send_r2f_encode + mmh_bn_encode_vector + mmh_chacha20_words + mmh_bn_server + mmh_cot_send_f2r,
 * wire bytes included; two moduli only. */
int mmh_bn_alice_prepare(const mmh_ctx *ctx, const uint64_t *d_rcm_r2f, const uint64_t *d_rcm_f2r,
                         const uint64_t delta[2], const uint64_t seed_r2f[2], const uint64_t seed_f2r[2], uint64_t gid0,
                         const uint64_t *d_x, size_t n, int bw, int shift, const uint64_t *d_scales,
                         const uint32_t key[8], uint64_t stream_id, uint64_t *d_pts3, uint64_t *d_out_share,
                         uint64_t *d_out_ring, uint8_t *d_wire_r2f, uint8_t *d_wire_f2r, void *stream);
int mmh_bn_alice_finish(const mmh_ctx *ctx, const uint64_t *d_in_cts, const uint64_t *d_pts3, size_t n,
                        uint64_t *d_out_cts, void *stream);

#ifdef __cplusplus
}
#endif
#endif
