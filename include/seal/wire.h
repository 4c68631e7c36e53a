// seal/wire.h -- the byte layout of SEAL 3.7.2's serialisation as M&M puts it on the wire, host code
// only (no device calls): what `ct.save(os)` / `ct.load(context, is)` / `pk.save(os)` of SCI/src/cheetah/cheetah-api.cpp
// :11-29,:113-135,:723-737 produce and accept, with patch/SEAL.patch:43-71's n_special_primes word in the parms_id hash.
//
// PARITY UNPINNED.  SEAL's sources are not in the reference tree (scripts/build-deps.sh fetches v3.7.2) and the
// reference holds no serialised object, so this is a restatement of SEAL's published format from its documented
// layout, anchored only where an independent implementation exists in this image: BLAKE2b against RFC 7693's vector
// and Python's hashlib, BLAKE2xb against hashlib's tree parameters (tests/test_seal_wire.py).  What it restates:
//   * Serialization::SEALHeader, 16 bytes: magic 0xA15E, header size 0x10, version 3.7, compr_mode, reserved, u64 total
//     size (header included);
//   * Ciphertext::save_members: parms_id (4 x u64), is_ntt_form (1 byte), size, poly_modulus_degree,
//     coeff_modulus_size, correction_factor (all u64), scale (f64), then the DynArray as an object of its own
//     (SEALHeader, u64 count, words) -- or, for a seeded ciphertext (Serializable<Ciphertext> of a symmetric
//     encryption), the DynArray of c0 alone followed by a UniformRandomGeneratorInfo object (SEALHeader, prng type
//     byte, 64-byte seed) from which the loader regenerates c1;
//   * the seed expansion: SEAL's default Blake2xb generator (4096-byte buffers, buffer i = BLAKE2xb(key = seed,
//     input = u64 counter i)) feeding sample_poly_uniform (one word per coefficient, limb after limb, a word at or
//     above the largest multiple of q redrawn from the same stream);
//   * parms_id = BLAKE2b-256 over the words {scheme, N, n_special_primes, q_0 .. q_{L-1}, t}.
// Compression: this writer emits compr_mode none, which every SEAL build loads.  An unmodified reference party writes
// compr_mode zstd (SEAL's default when built with it): the loader inflates it with the decoder of
// seal/zstd_decode.h (libzstd is not in this image as a C library; the decoder is checked against the libzstd inside
// pyarrow).  zlib streams are refused with a message that says so.  PublicKey::save is Ciphertext::save of its data.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "seal/zstd_decode.h"

namespace seal {
namespace wire {

// ------------------------------------------------------------------------------------------ BLAKE2b / BLAKE2xb
// RFC 7693.  The parameter block is passed whole so that BLAKE2xb (the BLAKE2X paper, section 2) can set its tree
// fields; P[0] = digest length, P[1] = key length.
class Blake2b {
 public:
  explicit Blake2b(const std::uint8_t (&P)[64]) {
    static const std::uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                        0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                        0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    for (int i = 0; i < 8; ++i) h_[i] = IV[i] ^ load64(P + 8 * i);
    outlen_ = P[0];
  }
  void update(const void *in, std::size_t n) {
    const std::uint8_t *p = static_cast<const std::uint8_t *>(in);
    while (n) {
      if (fill_ == 128) {  // the buffer is compressed only once more input follows: the last block is final
        count(128);
        compress(false);
        fill_ = 0;
      }
      const std::size_t take = n < 128 - fill_ ? n : 128 - fill_;
      std::memcpy(buf_ + fill_, p, take);
      fill_ += take, p += take, n -= take;
    }
  }
  void final(void *out) {
    count(fill_);
    std::memset(buf_ + fill_, 0, 128 - fill_);
    compress(true);
    std::uint8_t full[64];
    for (int i = 0; i < 8; ++i) store64(full + 8 * i, h_[i]);
    std::memcpy(out, full, outlen_);
  }
  static std::uint64_t load64(const std::uint8_t *p) {  // little-endian word
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
    std::uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
#else
    std::uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
#endif
  }
  static void store64(std::uint8_t *p, std::uint64_t v) {
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
    std::memcpy(p, &v, 8);
#else
    for (int i = 0; i < 8; ++i) p[i] = (std::uint8_t)(v >> (8 * i));
#endif
  }

 private:
  static std::uint64_t rotr(std::uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void count(std::size_t n) {
    t_[0] += n;
    if (t_[0] < n) ++t_[1];
  }
  void compress(bool last) {
    static const std::uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                        0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                        0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    std::uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) m[i] = load64(buf_ + 8 * i);
    for (int i = 0; i < 8; ++i) v[i] = h_[i], v[i + 8] = IV[i];
    v[12] ^= t_[0], v[13] ^= t_[1];
    if (last) v[14] = ~v[14];
    // (the message schedule as compile-time constants: the twelve rounds unroll into straight-line code, which is
    // what the seed expansion of every fresh ciphertext spends its time in)
#define MM_B2_G(a, b, c, d, x, y)     \
  v[a] = v[a] + v[b] + m[x];          \
  v[d] = rotr(v[d] ^ v[a], 32);       \
  v[c] = v[c] + v[d];                 \
  v[b] = rotr(v[b] ^ v[c], 24);       \
  v[a] = v[a] + v[b] + m[y];          \
  v[d] = rotr(v[d] ^ v[a], 16);       \
  v[c] = v[c] + v[d];                 \
  v[b] = rotr(v[b] ^ v[c], 63);
#define MM_B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  MM_B2_G(0, 4, 8, 12, s0, s1) MM_B2_G(1, 5, 9, 13, s2, s3) MM_B2_G(2, 6, 10, 14, s4, s5) MM_B2_G(3, 7, 11, 15, s6, s7) \
  MM_B2_G(0, 5, 10, 15, s8, s9) MM_B2_G(1, 6, 11, 12, s10, s11) MM_B2_G(2, 7, 8, 13, s12, s13) MM_B2_G(3, 4, 9, 14, s14, s15)
    MM_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    MM_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    MM_B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    MM_B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    MM_B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    MM_B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    MM_B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    MM_B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    MM_B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    MM_B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    MM_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    MM_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
#undef MM_B2_ROUND
#undef MM_B2_G
    for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[i + 8];
  }
  std::uint64_t h_[8], t_[2] = {0, 0};
  std::uint8_t buf_[128];
  std::size_t fill_ = 0, outlen_ = 64;
};

// sequential BLAKE2b: digest of `outlen` bytes, optional key (RFC 7693 section 2.5: the key is the first block)
inline void blake2b(void *out, std::size_t outlen, const void *in, std::size_t inlen, const void *key = nullptr,
                    std::size_t keylen = 0) {
  if (outlen == 0 || outlen > 64 || keylen > 64) throw std::invalid_argument("blake2b: lengths");
  std::uint8_t P[64] = {0};
  P[0] = (std::uint8_t)outlen, P[1] = (std::uint8_t)keylen, P[2] = 1, P[3] = 1;  // fanout 1, depth 1
  Blake2b s(P);
  if (keylen) {
    std::uint8_t block[128] = {0};
    std::memcpy(block, key, keylen);
    s.update(block, 128);
  }
  s.update(in, inlen);
  s.final(out);
}

// BLAKE2xb with a known output length below 2^32 - 1: root hash with xof_length set, then output block i =
// BLAKE2b(root) under parameters (digest length of the block, fanout 0, depth 0, leaf length 64, node offset i,
// xof_length, inner length 64)
inline void blake2xb(void *out, std::size_t outlen, const void *in, std::size_t inlen, const void *key,
                     std::size_t keylen) {
  if (outlen == 0 || outlen >= 0xFFFFFFFFull || keylen > 64) throw std::invalid_argument("blake2xb: lengths");
  auto put32 = [](std::uint8_t *p, std::uint32_t v) {
    for (int i = 0; i < 4; ++i) p[i] = (std::uint8_t)(v >> (8 * i));
  };
  std::uint8_t root[64];
  {
    std::uint8_t P[64] = {0};
    P[0] = 64, P[1] = (std::uint8_t)keylen, P[2] = 1, P[3] = 1;
    put32(P + 12, (std::uint32_t)outlen);  // xof_length sits in the upper half of BLAKE2b's node offset
    Blake2b s(P);
    if (keylen) {
      std::uint8_t block[128] = {0};
      std::memcpy(block, key, keylen);
      s.update(block, 128);
    }
    s.update(in, inlen);
    s.final(root);
  }
  std::uint8_t *o = static_cast<std::uint8_t *>(out);
  const std::uint32_t xof_length = (std::uint32_t)outlen;
  for (std::uint32_t i = 0; outlen; ++i) {
    const std::size_t take = outlen < 64 ? outlen : 64;
    std::uint8_t P[64] = {0};
    P[0] = (std::uint8_t)take;
    put32(P + 4, 64);                            // leaf length
    put32(P + 8, i);                             // node offset
    put32(P + 12, xof_length);                   // the whole output's length
    P[17] = 64;                                  // inner length (P[16] = node depth 0)
    Blake2b s(P);
    s.update(root, 64);
    s.final(o);
    o += take, outlen -= take;
  }
}

// EncryptionParameters::compute_parms_id with patch/SEAL.patch:43-71: BLAKE2b-256 of the parameter words
inline std::array<std::uint64_t, 4> parms_id_hash(std::uint8_t scheme, std::uint64_t poly_modulus_degree,
                                                  std::uint64_t n_special_primes,
                                                  const std::vector<std::uint64_t> &coeff_modulus,
                                                  std::uint64_t plain_modulus) {
  std::vector<std::uint64_t> w{scheme, poly_modulus_degree, n_special_primes};
  w.insert(w.end(), coeff_modulus.begin(), coeff_modulus.end());
  w.push_back(plain_modulus);
  std::vector<std::uint8_t> bytes(w.size() * 8);
  for (std::size_t i = 0; i < w.size(); ++i) Blake2b::store64(bytes.data() + 8 * i, w[i]);
  std::uint8_t d[32];
  blake2b(d, 32, bytes.data(), bytes.size());
  return {Blake2b::load64(d), Blake2b::load64(d + 8), Blake2b::load64(d + 16), Blake2b::load64(d + 24)};
}

// ------------------------------------------------------------------------------------------ SEAL's default PRNG
constexpr std::size_t kSeedBytes = 64;        // prng_seed_type: 8 x u64
enum : std::uint8_t { prng_unknown = 0, prng_blake2xb = 1, prng_shake256 = 2 };
class Blake2xbPRNG {
 public:
  explicit Blake2xbPRNG(const std::uint8_t (&seed)[kSeedBytes]) { std::memcpy(seed_, seed, kSeedBytes); }
  void generate(std::size_t n, void *dst) {
    std::uint8_t *o = static_cast<std::uint8_t *>(dst);
    while (n) {
      if (head_ == kBuffer) refill();
      const std::size_t take = n < kBuffer - head_ ? n : kBuffer - head_;
      std::memcpy(o, buf_ + head_, take);
      head_ += take, o += take, n -= take;
    }
  }

 private:
  static constexpr std::size_t kBuffer = 4096;
  void refill() {
    std::uint8_t ctr[8];
    Blake2b::store64(ctr, counter_++);
    blake2xb(buf_, kBuffer, ctr, 8, seed_, kSeedBytes);
    head_ = 0;
  }
  std::uint8_t seed_[kSeedBytes], buf_[kBuffer];
  std::uint64_t counter_ = 0;
  std::size_t head_ = kBuffer;
};
// util::sample_poly_uniform: dst [L][N], one word per coefficient drawn for the whole array first, then limb by limb
// every word at or above the largest multiple of q_l is redrawn (from the stream's current position) and reduced
inline void sample_poly_uniform(Blake2xbPRNG &prng, const std::vector<std::uint64_t> &q, std::size_t N,
                                std::uint64_t *dst) {
  prng.generate(q.size() * N * 8, dst);
  for (std::size_t l = 0; l < q.size(); ++l) {
    const std::uint64_t max_multiple = ~std::uint64_t(0) - (~std::uint64_t(0) % q[l]) - 1;
    for (std::size_t i = 0; i < N; ++i) {
      std::uint64_t r = dst[l * N + i];
      while (r >= max_multiple) prng.generate(8, &r);
      dst[l * N + i] = r % q[l];
    }
  }
}

// ------------------------------------------------------------------------------------------ framing
enum : std::uint8_t { compr_none = 0, compr_zlib = 1, compr_zstd = 2 };
constexpr std::uint16_t kMagic = 0xA15E;
constexpr std::size_t kHeaderBytes = 16;
constexpr std::size_t kCtMemberBytes = 32 + 1 + 5 * 8;       // parms_id .. scale
constexpr std::size_t kPrngInfoBytes = kHeaderBytes + 1 + kSeedBytes;

struct Writer {
  std::string bytes;
  void raw(const void *p, std::size_t n) { bytes.append(static_cast<const char *>(p), n); }
  void u8(std::uint8_t v) { bytes.push_back((char)v); }
  void u16(std::uint16_t v) { u8((std::uint8_t)v), u8((std::uint8_t)(v >> 8)); }
  void u64(std::uint64_t v) {
    std::uint8_t b[8];
    Blake2b::store64(b, v);
    raw(b, 8);
  }
  void words(const std::uint64_t *v, std::size_t n) {  // n little-endian words
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
    raw(v, n * 8);
#else
    for (std::size_t i = 0; i < n; ++i) u64(v[i]);
#endif
  }
  void header(std::uint64_t total) { u16(kMagic), u8(0x10), u8(3), u8(7), u8(compr_none), u16(0), u64(total); }
};
struct Reader {
  const std::uint8_t *p, *end;
  void need(std::size_t n) const {
    if ((std::size_t)(end - p) < n) throw std::logic_error("seal wire: truncated stream");
  }
  std::uint8_t u8() { return need(1), *p++; }
  std::uint16_t u16() {
    need(2);
    const std::uint16_t v = (std::uint16_t)(p[0] | (p[1] << 8));
    return p += 2, v;
  }
  std::uint64_t u64() {
    need(8);
    const std::uint64_t v = Blake2b::load64(p);
    return p += 8, v;
  }
  void raw(void *dst, std::size_t n) {
    need(n);
    std::memcpy(dst, p, n);
    p += n;
  }
  // one SEALHeader; returns the bytes that follow it inside its object and the object's compr_mode
  std::size_t header(std::uint8_t *compr_mode = nullptr) {
    need(kHeaderBytes);
    const std::uint16_t magic = u16();
    const std::uint8_t hsize = u8(), vmaj = u8(), vmin = u8(), compr = u8();
    (void)u16();
    const std::uint64_t total = u64();
    if (magic != kMagic || hsize != 0x10) throw std::logic_error("seal wire: not a SEAL object (bad header)");
    // 3.7 only: load_ciphertext reads 3.7's member list (with correction_factor), which a 3.6 object does not have
    if (vmaj != 3 || vmin != 7) throw std::logic_error("seal wire: SEAL version other than 3.7");
    if (compr == compr_zlib)
      throw std::logic_error("seal wire: compressed stream (compr_mode zlib): no zlib decoder in this build; the peer "
                             "must save with compr_mode_type::zstd or none");
    if (compr != compr_none && compr != compr_zstd) throw std::logic_error("seal wire: unknown compr_mode");
    if (compr == compr_zstd && !compr_mode)
      throw std::logic_error("seal wire: compressed stream (compr_mode zstd) where SEAL writes none");
    if (compr_mode) *compr_mode = compr;
    if (total < kHeaderBytes) throw std::logic_error("seal wire: size field below the header size");
    need(total - kHeaderBytes);
    return (std::size_t)total - kHeaderBytes;
  }
};

// The largest ciphertext object a context of this facade can hold (16 polynomials of N = 8192 with two moduli, plus
// headers): what a peer-declared zstd content size may make the loader reserve before any metadata is validated.
static constexpr std::size_t kMaxCtBytes = 16 * 8192 * 2 * 8 + 4096;

struct CtMeta {
  std::array<std::uint64_t, 4> parms_id{};
  bool is_ntt_form = false;
  std::uint64_t size = 0, poly_modulus_degree = 0, coeff_modulus_size = 0, correction_factor = 1;
  double scale = 1.0;
};

// Ciphertext::save with compr_mode none.  `seed` non-null: the seeded form (data = c0 alone, size stays 2).
inline std::string save_ciphertext(const CtMeta &m, const std::uint64_t *data, const std::uint8_t *seed = nullptr) {
  const std::size_t per_poly = (std::size_t)(m.poly_modulus_degree * m.coeff_modulus_size);
  const std::size_t words = seed ? per_poly : (std::size_t)m.size * per_poly;
  const std::size_t array_bytes = kHeaderBytes + 8 + words * 8;
  const std::size_t total = kHeaderBytes + kCtMemberBytes + array_bytes + (seed ? kPrngInfoBytes : 0);
  Writer w;
  w.bytes.reserve(total);
  w.header(total);
  for (int i = 0; i < 4; ++i) w.u64(m.parms_id[i]);
  w.u8(m.is_ntt_form ? 1 : 0);
  w.u64(m.size), w.u64(m.poly_modulus_degree), w.u64(m.coeff_modulus_size), w.u64(m.correction_factor);
  std::uint64_t scale_bits;
  std::memcpy(&scale_bits, &m.scale, 8);
  w.u64(scale_bits);
  w.header(array_bytes);
  w.u64(words);
  w.words(data, words);
  if (seed) {
    w.header(kPrngInfoBytes);
    w.u8(prng_blake2xb);
    w.raw(seed, kSeedBytes);
  }
  return w.bytes;
}

// Ciphertext::load_members up to the data: fills `m`, returns the words present and sets seeded / seed.  `data`
// receives size * N * L words; for a seeded object the second polynomial is left for the caller to expand.
inline std::size_t load_ciphertext(const std::uint8_t *bytes, std::size_t n, CtMeta &m, std::vector<std::uint64_t> &data,
                                   bool &seeded, std::uint8_t (&seed)[kSeedBytes], std::size_t *consumed = nullptr,
                                   std::size_t max_inflated = kMaxCtBytes) {
  Reader outer{bytes, bytes + n};
  std::uint8_t compr = compr_none;
  std::size_t body = outer.header(&compr);
  if (consumed) *consumed = kHeaderBytes + body;
  // SEAL's default: the members behind the header are one Zstandard stream (the objects nested inside are saved with
  // compr_mode none); inflate it and read the members from the result
  std::vector<std::uint8_t> inflated;
  Reader r{outer.p, outer.p + body};
  if (compr == compr_zstd) {
    try {
      inflated = zstd::decompress(outer.p, body, max_inflated);
    } catch (const zstd::Error &e) {
      throw std::logic_error(std::string("seal wire: ") + e.what());
    }
    r = Reader{inflated.data(), inflated.data() + inflated.size()};
    body = inflated.size();
  }
  const std::uint8_t *obj_end = r.p + body;
  if (body < kCtMemberBytes) throw std::logic_error("seal wire: ciphertext members missing");
  for (int i = 0; i < 4; ++i) m.parms_id[i] = r.u64();
  m.is_ntt_form = r.u8() != 0;
  m.size = r.u64(), m.poly_modulus_degree = r.u64(), m.coeff_modulus_size = r.u64(), m.correction_factor = r.u64();
  const std::uint64_t scale_bits = r.u64();
  std::memcpy(&m.scale, &scale_bits, 8);
  if (m.size < 2 || m.size > 16 || m.poly_modulus_degree > 32768 || m.coeff_modulus_size > 64)
    throw std::logic_error("seal wire: ciphertext metadata out of range");
  const std::size_t per_poly = (std::size_t)(m.poly_modulus_degree * m.coeff_modulus_size), total = (std::size_t)m.size * per_poly;
  const std::size_t array_body = r.header();
  const std::uint64_t words = r.u64();
  if (array_body != 8 + words * 8) throw std::logic_error("seal wire: DynArray size mismatch");
  seeded = words == per_poly && m.size == 2;
  if (!seeded && words != total) throw std::logic_error("seal wire: ciphertext data has the wrong length");
  data.assign(total, 0);
  r.need(words * 8);
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
  std::memcpy(data.data(), r.p, words * 8), r.p += words * 8;
#else
  for (std::size_t i = 0; i < words; ++i) data[i] = r.u64();
#endif
  if (seeded) {
    const std::size_t info_body = r.header();
    if (info_body != 1 + kSeedBytes) throw std::logic_error("seal wire: UniformRandomGeneratorInfo size mismatch");
    const std::uint8_t type = r.u8();
    if (type != prng_blake2xb)
      throw std::logic_error("seal wire: seeded ciphertext with a generator other than Blake2xb (type " +
                             std::to_string(type) + ")");
    r.raw(seed, kSeedBytes);
  }
  if (r.p != obj_end) throw std::logic_error("seal wire: ciphertext object size mismatch");
  return (std::size_t)words;
}

// the size field of the next object in a stream whose first 16 bytes are given (what a stream loader must read first)
inline std::size_t peek_total(const std::uint8_t (&head)[kHeaderBytes]) {
  if ((std::uint16_t)(head[0] | (head[1] << 8)) != kMagic || head[2] != 0x10)
    throw std::logic_error("seal wire: not a SEAL object (bad header)");
  return (std::size_t)Blake2b::load64(head + 8);
}

}  // namespace wire
}  // namespace seal
