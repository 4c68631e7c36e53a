// Device-side randomness: ChaCha20 keystream (RFC 8439 block function, 64-bit block counter in
// words 12-13, 64-bit stream id in words 14-15) and the three distributions the path draws from:
// uniform mod q (seal::util::sample_poly_uniform, hom_conv2d_ss.cc:206), ternary
// (sample_poly_ternary, :302) and centred binomial noise (SEAL_NOISE_SAMPLER, :329; SEAL's
// sample_poly_cbd: popcount of 21 bits minus popcount of 21 bits).  The reference draws from
// SEAL's Blake2xb generator; streams are not comparable across implementations, only the
// distributions are, and every consumer of randomness in this library takes it as an argument.
#include "common.hpp"
#include "sampler_device.hpp"

namespace mmh {
// Raw keystream: block b = words 8b .. 8b+7 (what the RFC's test vector pins)
__global__ __launch_bounds__(256) void sample_raw_kernel(ChaChaKey key, u64 stream_id, u64 counter0, u64 *out,
                                                         size_t n_words) {
  const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (blk * 8 >= n_words) return;
  u64 w[8];
  chacha20_block(key, stream_id, counter0 + blk, w);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (blk * 8 + i < n_words) out[blk * 8 + i] = w[i];
}

// MODE 1: ternary in {-1,0,1}; 2: centred binomial (21 - 21 bits)
template <int MODE>
__global__ __launch_bounds__(256) void sample_words_kernel(ChaChaKey key, u64 stream_id, u64 counter0, u64 *out,
                                                           size_t n_words) {
  if (MODE == 1) ternary_block(key, stream_id, counter0, blockIdx.x, threadIdx.x, (int64_t *)out, n_words);
  if (MODE == 2) noise_block(key, stream_id, counter0, blockIdx.x, threadIdx.x, (int64_t *)out, n_words);
}

// uniform in [0, q_l)
__global__ __launch_bounds__(256) void sample_uniform_kernel(ChaChaKey key, u64 stream_id, u64 counter0, u64 *out,
                                                             size_t total, int logN, int limbs,
                                                             const DevMod *__restrict__ mods) {
  uniform_block(key, stream_id, counter0, blockIdx.x, threadIdx.x, out, total, logN, limbs, mods);
}

// the randomness of n re-sharings in one launch: u = ternary(stream_id), e1 = noise(stream_id + 1), mask =
// uniform at the last level (stream_id + 2), n polynomials of N coefficients each: coefficients [first, first + total)
// of the three streams
struct ReshareTiles {
  unsigned count_u, count_e;    // blocks [0, count_u) draw u, the next count_e draw e1, the rest the mask
  unsigned lo_u, lo_e, lo_m;    // the streams' first tiles that meet the range
};
__global__ __launch_bounds__(256) void sample_reshare_kernel(ChaChaKey key, u64 stream_id, int64_t *u, int64_t *e1,
                                                             u64 *mask, size_t first, size_t total, int logN,
                                                             ReshareTiles tl, const DevMod *__restrict__ mods) {
  const unsigned b = blockIdx.x;
  if (b < tl.count_u) ternary_block(key, stream_id, 0, tl.lo_u + b, threadIdx.x, u, total, first);
  else if (b < tl.count_u + tl.count_e)
    noise_block(key, stream_id + 1, 0, tl.lo_e + (b - tl.count_u), threadIdx.x, e1, total, first);
  else
    uniform_block(key, stream_id + 2, 0, tl.lo_m + (b - tl.count_u - tl.count_e), threadIdx.x, mask, total, logN, 1, mods,
                  first);
}

__global__ __launch_bounds__(256) void signed_to_rns_kernel2(const int64_t *__restrict__ v, u64 *__restrict__ out,
                                                             size_t total, int logN, int limbs,
                                                             const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const int64_t x = v[idx];
  for (int l = 0; l < limbs; ++l) {
    const u64 q = mods[l].q;
    out[((k * limbs + l) << logN) + i] = x >= 0 ? (u64)x : q - (u64)(-x);
  }
}
int ntt_dispatch(const mmh_ctx *ctx, bool inverse, u64 *d_polys, size_t n_polys, int limbs, int mod_base,
                 hipStream_t s);
static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace mmh

using namespace mmh;

static unsigned tiles_of(size_t n, int per_block) { return (unsigned)((n + 256 * (size_t)per_block - 1) / (256 * (size_t)per_block)); }

static int sample_words(int mode, const uint32_t key[8], uint64_t stream_id, uint64_t counter0, uint64_t *d_out,
                        size_t n_words, void *stream) {
  MMH_REQUIRE(key && (d_out || n_words == 0), MMH_ERR_NULL_POINTER);
  if (n_words == 0) return MMH_OK;
  ChaChaKey k;
  for (int i = 0; i < 8; ++i) k.k[i] = key[i];
  if (mode == 0)
    hipLaunchKernelGGL(sample_raw_kernel, dim3(nblk((n_words + 7) / 8)), dim3(256), 0, (hipStream_t)stream, k, stream_id,
                       counter0, d_out, n_words);
  else if (mode == 1)
    hipLaunchKernelGGL(sample_words_kernel<1>, dim3(tiles_of(n_words, kTernaryPerBlock)), dim3(256), 0,
                       (hipStream_t)stream, k, stream_id, counter0, d_out, n_words);
  else
    hipLaunchKernelGGL(sample_words_kernel<2>, dim3(tiles_of(n_words, kNoisePerBlock)), dim3(256), 0,
                       (hipStream_t)stream, k, stream_id, counter0, d_out, n_words);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_chacha20_words(const uint32_t key[8], uint64_t stream_id, uint64_t counter0, uint64_t *d_out,
                                  size_t n_words, void *stream) {
  return sample_words(0, key, stream_id, counter0, d_out, n_words, stream);
}
extern "C" int mmh_sample_ternary(const uint32_t key[8], uint64_t stream_id, int64_t *d_out, size_t n_coeffs,
                                  void *stream) {
  return sample_words(1, key, stream_id, 0, (uint64_t *)d_out, n_coeffs, stream);
}
extern "C" int mmh_sample_noise(const uint32_t key[8], uint64_t stream_id, int64_t *d_out, size_t n_coeffs,
                                void *stream) {
  return sample_words(2, key, stream_id, 0, (uint64_t *)d_out, n_coeffs, stream);
}
extern "C" int mmh_sample_uniform(const mmh_ctx *ctx, int lv, const uint32_t key[8], uint64_t stream_id,
                                  uint64_t *d_out, size_t n_polys, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && key && (d_out || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(lv == 0 || lv == 1, MMH_ERR_INVALID_ARG);
  if (n_polys == 0) return MMH_OK;
  ChaChaKey k;
  for (int i = 0; i < 8; ++i) k.k[i] = key[i];
  const int limbs = lv == 0 ? (int)ctx->L : 1;
  const size_t total = n_polys * limbs * ctx->N;
  hipLaunchKernelGGL(sample_uniform_kernel, dim3(tiles_of(total, kUniformPerBlock)), dim3(256), 0, (hipStream_t)stream,
                     k, stream_id, (u64)0, d_out, total, ctx->logN, limbs, ctx->d_mods);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
extern "C" int mmh_sample_reshare_range(const mmh_ctx *ctx, const uint32_t key[8], uint64_t stream_id, size_t first_poly,
                                        size_t n, int64_t *d_u, int64_t *d_e1, uint64_t *d_mask, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && key && (n == 0 || (d_u && d_e1 && d_mask)), MMH_ERR_NULL_POINTER);
  if (n == 0) return MMH_OK;
  ChaChaKey k;
  for (int i = 0; i < 8; ++i) k.k[i] = key[i];
  const size_t first = first_poly * ctx->N, total = n * ctx->N;
  auto lo = [&](int per) { return (unsigned)(first / (256 * (size_t)per)); };
  auto count = [&](int per) { return tiles_of(first + total, per) - lo(per); };
  ReshareTiles tl{count(kTernaryPerBlock), count(kNoisePerBlock), lo(kTernaryPerBlock), lo(kNoisePerBlock),
                  lo(kUniformPerBlock)};
  hipLaunchKernelGGL(sample_reshare_kernel, dim3(tl.count_u + tl.count_e + count(kUniformPerBlock)), dim3(256), 0,
                     (hipStream_t)stream, k, stream_id, d_u, d_e1, d_mask, first, total, ctx->logN, tl, ctx->d_mods);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
extern "C" int mmh_sample_reshare(const mmh_ctx *ctx, const uint32_t key[8], uint64_t stream_id, size_t n, int64_t *d_u,
                                  int64_t *d_e1, uint64_t *d_mask, void *stream) {
  return mmh_sample_reshare_range(ctx, key, stream_id, 0, n, d_u, d_e1, d_mask, stream);
}
extern "C" int mmh_signed_to_ntt(const mmh_ctx *ctx, int lv, const int64_t *d_signed, size_t n, uint64_t *d_out,
                                 void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && (n == 0 || (d_signed && d_out)), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(lv == 0 || lv == 1, MMH_ERR_INVALID_ARG);
  if (n == 0) return MMH_OK;
  const int limbs = lv == 0 ? (int)ctx->L : 1;
  hipLaunchKernelGGL(signed_to_rns_kernel2, dim3(nblk(n * ctx->N)), dim3(256), 0, (hipStream_t)stream, d_signed,
                     d_out, n * ctx->N, ctx->logN, limbs, ctx->d_mods);
  MMH_HIP(hipGetLastError());
  return ntt_dispatch(ctx, false, d_out, n, limbs, 0, (hipStream_t)stream);
}
