// Device-side randomness shared by sampler.hip (the stand-alone sampling launches) and by the kernels that draw the
// randomness they consume in place (conv.hip's front launch, ntt_device.hpp's SRC_TERNARY): the ChaCha20 block function
// and the three distributions.  One definition, so a draw made in place equals the same draw made by mmh_sample_*.
#pragma once
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


// The 16 ternary samples thread t of tile `tile` draws (coefficients tile * 4096 + 256 j + t of stream sid), in a
// register array: what ternary_block stores, for a consumer that wants them where they are (the transform of u).
__device__ __forceinline__ void ternary_samples(const ChaChaKey &key, u64 sid, size_t tile, int t, int64_t (&v)[16]) {
  u64 w[8];
  chacha20_block(key, sid, tile * 256 + t, w);
#pragma unroll
  for (int j = 0; j < kTernaryPerBlock; ++j) v[j] = (int64_t)__umulhi((u32)(w[j >> 1] >> ((j & 1) * 32)), 3u) - 1;
}
}  // namespace mmh
