// Device-side randomness: ChaCha20 keystream (RFC 8439 block function, 64-bit block counter in
// words 12-13, 64-bit stream id in words 14-15) and the three distributions the path draws from:
// uniform mod q (seal::util::sample_poly_uniform, hom_conv2d_ss.cc:206), ternary
// (sample_poly_ternary, :302) and centred binomial noise (SEAL_NOISE_SAMPLER, :329; SEAL's
// sample_poly_cbd: popcount of 21 bits minus popcount of 21 bits).  The reference draws from
// SEAL's Blake2xb generator; streams are not comparable across implementations, only the
// distributions are, and every consumer of randomness in this library takes it as an argument.
#include "common.hpp"

namespace mmh {
struct ChaChaKey {
  u32 k[8];
};

__device__ __forceinline__ u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define MMH_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); \
  c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  \
  c += d; b ^= c; b = rotl32(b, 7);

// one 64-byte block as 8 little-endian u64 words
__device__ __forceinline__ void chacha20_block(const ChaChaKey &key, u64 stream_id, u64 counter, u64 out[8]) {
  const u32 in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                      key.k[4], key.k[5], key.k[6], key.k[7], (u32)counter, (u32)(counter >> 32), (u32)stream_id,
                      (u32)(stream_id >> 32)};
  u32 x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = in[i];
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    MMH_QR(x[0], x[4], x[8], x[12]) MMH_QR(x[1], x[5], x[9], x[13]) MMH_QR(x[2], x[6], x[10], x[14])
    MMH_QR(x[3], x[7], x[11], x[15]) MMH_QR(x[0], x[5], x[10], x[15]) MMH_QR(x[1], x[6], x[11], x[12])
    MMH_QR(x[2], x[7], x[8], x[13]) MMH_QR(x[3], x[4], x[9], x[14])
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (u64)(x[2 * i] + in[2 * i]) | ((u64)(x[2 * i + 1] + in[2 * i + 1]) << 32);
}

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

// The three distributions, each spending the keystream bits its samples need and no more (a ChaCha20 block costs
// ~1000 vector instructions per lane; the re-sharing of one output ciphertext draws 3 x 4096 samples):
//   ternary  16 per block: a 32-bit word w -> floor(3 w / 2^32) - 1   (each value within 2^-32 of 1/3)
//   noise    12 per block: 42 bits -> popcount(21 bits) - popcount(21 bits)   (SEAL's sample_poly_cbd, exact)
//   uniform   8 per block: a 64-bit word w below the largest multiple of q -> w mod q (exact); the word is rejected
//             with probability < q / 2^64 and the sample then comes from 128 bits of a block of its own
// Block -> sample index: thread t of the 256-thread tile T computes block counter0 + 256 T + t, and its j-th sample
// is element (256 W) T + 256 j + t of the output (W samples per block), so every store instruction of a wave writes
// 64 consecutive words.  The fused launch below and the stand-alone entry points share these functions, hence agree.
constexpr int kTernaryPerBlock = 16, kNoisePerBlock = 12, kUniformPerBlock = 8;

// `tile` is the tile's index in the whole stream; of its samples those with index in [first, first + n) are stored, at
// out[index - first] (a range of a longer draw equals that part of the whole draw: the host's device shards use it)
__device__ __forceinline__ void ternary_block(const ChaChaKey &key, u64 sid, u64 counter0, size_t tile, int t,
                                              int64_t *out, size_t n, size_t first = 0) {
  u64 w[8];
  chacha20_block(key, sid, counter0 + tile * 256 + t, w);
  const size_t base = tile * (256 * kTernaryPerBlock) + t;
#pragma unroll
  for (int j = 0; j < kTernaryPerBlock; ++j) {
    const u32 r = (u32)(w[j >> 1] >> ((j & 1) * 32));
    const size_t idx = base + (size_t)j * 256;
    if (idx - first < n) out[idx - first] = (int64_t)__umulhi(r, 3u) - 1;   // (unsigned: idx < first wraps above n)
  }
}
__device__ __forceinline__ void noise_block(const ChaChaKey &key, u64 sid, u64 counter0, size_t tile, int t,
                                            int64_t *out, size_t n, size_t first = 0) {
  u64 w[8];
  chacha20_block(key, sid, counter0 + tile * 256 + t, w);
  const size_t base = tile * (256 * kNoisePerBlock) + t;
#pragma unroll
  for (int j = 0; j < kNoisePerBlock; ++j) {
    const int bit = 42 * j, a = bit >> 6, sh = bit & 63;
    u64 v = w[a] >> sh;
    if (sh > 22) v |= w[a + 1] << (64 - sh);   // (a + 1 <= 7: 42 * 12 = 504 bits)
    const size_t idx = base + (size_t)j * 256;
    if (idx - first < n) out[idx - first] = (int64_t)__popcll(v & 0x1FFFFF) - (int64_t)__popcll((v >> 21) & 0x1FFFFF);
  }
}
__device__ __forceinline__ void uniform_block(const ChaChaKey &key, u64 sid, u64 counter0, size_t tile, int t, u64 *out,
                                              size_t n, int logN, int limbs, const DevMod *__restrict__ mods,
                                              size_t first = 0) {
  u64 w[8];
  chacha20_block(key, sid, counter0 + tile * 256 + t, w);
  const size_t base = tile * (256 * kUniformPerBlock) + t;
#pragma unroll
  for (int j = 0; j < kUniformPerBlock; ++j) {
    const size_t idx = base + (size_t)j * 256;
    if (idx - first >= n) continue;
    const DevMod &q = mods[(idx >> logN) % limbs];
    u64 v;
    if (w[j] < q.q * q.barrett_hi) {   // barrett_hi = floor(2^64 / q): below the largest multiple of q
      v = reduce64(w[j], q);
    } else {                           // (probability < q / 2^64) 128 fresh bits from a block only this sample uses
      u64 x[8];
      chacha20_block(key, sid, (u64(1) << 63) | idx, x);
      v = csub(barrett128(reduce64(x[1], q), x[0], q.q, q.barrett_hi, q.barrett_lo), q.q);
    }
    out[idx - first] = v;
  }
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
