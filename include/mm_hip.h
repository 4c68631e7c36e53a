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
int mmh_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *stream);
int mmh_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *stream);
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

#ifdef __cplusplus
}
#endif
#endif
