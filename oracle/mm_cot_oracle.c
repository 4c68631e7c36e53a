/* CPU oracle (TEST INFRASTRUCTURE ONLY, see mm_oracle.h) for the post-processing of the correlated OT:
 * SilentOT::send_ot_cam_cc / send_ot_cam_cc_prime / recv_ot_cam_cc / recv_ot_cam_cc_prime
 * (SCI/src/OT/ferret/silent_ot.h:85-242) with sci::pack_cot_messages / unpack_cot_messages
 * (SCI/src/OT/ot-utils.h:154-216).  Input is what Ferret hands over (rcm_data, Delta, the choice bits) and
 * the hash seed S the sender draws; output is data0 / data and the bytes of the correction message.
 *
 * Pinning.  The mask / mod-p / correction / packing arithmetic follows the reference's own lines cited
 * below.  The hash is emp-tool's MITCCRH<8> (emp-tool/utils/mitccrh.h at 44b1dde, the commit
 * scripts/build-deps.sh:19 checks out), a dependency that is ABSENT from /root/reference: it is restated
 * from its published form [upstream-recall] -- key_i = S xor makeBlock(gid++, 0), eight keys scheduled per
 * renewal, out = blk xor AES-128_{key_i}(blk), key i used for the H blocks of element i -- and is PARITY
 * UNPINNED beyond AES-128 itself, which is pinned by the FIPS-197 known answers (tests/test_cot_oracle.py).
 * A block is 16 bytes in memory order; as two little-endian 64-bit words (lo, hi) here. */
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

/* ---- AES-128 (FIPS-197), byte-wise; state bytes in memory order like AESENC sees an xmm register ---- */
static uint8_t g_sbox[256];
static int g_sbox_ready = 0;
static uint8_t gf_mul(uint8_t a, uint8_t b) {
  uint8_t r = 0;
  while (b) {
    if (b & 1) r ^= a;
    a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
    b >>= 1;
  }
  return r;
}
static void sbox_init(void) {
  if (g_sbox_ready) return;
  for (int x = 0; x < 256; x++) {
    uint8_t inv = 0;
    for (int y = 1; y < 256 && x; y++)
      if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
    uint8_t s = inv, r = inv;
    for (int k = 0; k < 4; k++) {
      r = (uint8_t)((r << 1) | (r >> 7));
      s ^= r;
    }
    g_sbox[x] = s ^ 0x63;
  }
  g_sbox_ready = 1;
}

void mmo_aes128_expand(const uint8_t key[16], uint8_t rk[176]) {
  sbox_init();
  memcpy(rk, key, 16);
  uint8_t rcon = 1;
  for (int i = 4; i < 44; i++) {
    uint8_t t[4];
    memcpy(t, rk + 4 * (i - 1), 4);
    if (i % 4 == 0) {
      const uint8_t t0 = t[0];
      t[0] = g_sbox[t[1]] ^ rcon;
      t[1] = g_sbox[t[2]];
      t[2] = g_sbox[t[3]];
      t[3] = g_sbox[t0];
      rcon = (uint8_t)((rcon << 1) ^ ((rcon & 0x80) ? 0x1b : 0));
    }
    for (int k = 0; k < 4; k++) rk[4 * i + k] = rk[4 * (i - 4) + k] ^ t[k];
  }
}

void mmo_aes128_encrypt(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16]) {
  sbox_init();
  uint8_t s[16], t[16];
  for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
  for (int r = 1; r <= 10; r++) {
    /* SubBytes + ShiftRows: byte (row i, column c) sits at s[4c + i] */
    for (int c = 0; c < 4; c++)
      for (int i = 0; i < 4; i++) t[4 * c + i] = g_sbox[s[4 * ((c + i) & 3) + i]];
    if (r < 10) {
      for (int c = 0; c < 4; c++) {
        const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
        s[4 * c + 0] = (uint8_t)(gf_mul(a0, 2) ^ gf_mul(a1, 3) ^ a2 ^ a3);
        s[4 * c + 1] = (uint8_t)(a0 ^ gf_mul(a1, 2) ^ gf_mul(a2, 3) ^ a3);
        s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ gf_mul(a2, 2) ^ gf_mul(a3, 3));
        s[4 * c + 3] = (uint8_t)(gf_mul(a0, 3) ^ a1 ^ a2 ^ gf_mul(a3, 2));
      }
    } else {
      memcpy(s, t, 16);
    }
    for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
  }
  memcpy(out, s, 16);
}

/* MITCCRH<8>: one hash<8,H> call per batch of eight elements [upstream-recall, see header].  blocks and out
 * hold n*H blocks, element j's H blocks adjacent; element j is hashed under key S ^ makeBlock(gid0 + j, 0)
 * (the gid in the HIGH 64 bits: makeBlock(high, low) = _mm_set_epi64x(high, low)). */
void mmo_mitccrh_hash(const uint64_t S[2], uint64_t gid0, const uint64_t *blocks, size_t n, int H, uint64_t *out) {
  uint8_t rk[176];
  for (size_t j = 0; j < n; j++) {
    uint64_t key[2] = {S[0], S[1] ^ (gid0 + j)};
    mmo_aes128_expand((const uint8_t *)key, rk);
    for (int h = 0; h < H; h++) {
      const uint64_t *b = blocks + 2 * (j * (size_t)H + h);
      uint64_t e[2];
      mmo_aes128_encrypt(rk, (const uint8_t *)b, (uint8_t *)e);
      out[2 * (j * (size_t)H + h)] = b[0] ^ e[0];
      out[2 * (j * (size_t)H + h) + 1] = b[1] ^ e[1];
    }
  }
}

/* the gids one call consumes: every hash<8,H> call renews eight keys, a ragged last batch included
 * (silent_ot.h:103-109 hashes the whole pad array) */
uint64_t mmo_cot_gids(size_t n) { return 8 * (uint64_t)((n + 7) / 8); }

static int prime_bytes(uint64_t p) { /* silent_ot.h:139: ceil(log2(p) / 8.0) */
  int bits = 0;
  while (bits < 64 && (p >> bits) > 1) bits++; /* floor(log2 p) */
  if (p & (p - 1)) bits++;                     /* ceil */
  return (bits + 7) / 8;
}

/* bytes on the wire: ring -- per batch of 8 elements ceil(bsize*l/64) 64-bit words (silent_ot.h:117-123);
 * prime -- ceil(log2 p / 8) bytes per element (silent_ot.h:139,157-159) */
size_t mmo_cot_wire_bytes(size_t n, int l, uint64_t p, int prime) {
  if (prime) return n * (size_t)prime_bytes(p);
  const size_t full = n / 8, rem = n % 8;
  return 8 * (full * (size_t)((8 * l + 63) / 64) + (rem * (size_t)l + 63) / 64);
}

/* ot-utils.h:154-186 */
static void pack_cot(uint64_t *y, const uint64_t *corr_data, int ysize, int bsize, int bitsize) {
  const uint64_t mask = bitsize == 64 ? ~0ull : (1ull << bitsize) - 1;
  for (int i = 0; i < ysize; i++) y[i] = 0;
  for (int i = 0; i < bsize; i++) {
    const uint64_t start = (uint64_t)i * bitsize, end = start + bitsize - 1;
    const uint64_t sb = start / 64, eb = end / 64, v = corr_data[i] & mask;
    y[sb] ^= v << (start % 64);
    if (sb != eb) y[eb] ^= v >> (64 - start % 64);
  }
}
/* ot-utils.h:188-216 (a straddling value keeps its high garbage bits, the caller masks) */
static void unpack_cot(uint64_t *corr_data, const uint64_t *recvd, int bsize, int bitsize) {
  const uint64_t mask = bitsize == 64 ? ~0ull : (1ull << bitsize) - 1;
  for (int i = 0; i < bsize; i++) {
    const uint64_t start = (uint64_t)i * bitsize, end = start + bitsize - 1;
    const uint64_t sb = start / 64, eb = end / 64;
    if (sb == eb)
      corr_data[i] = (recvd[sb] >> (start % 64)) & mask;
    else
      corr_data[i] = (recvd[sb] >> (start % 64)) ^ (recvd[eb] << (64 - start % 64));
  }
}

/* silent_ot.h:85-127 (prime = 0) and :129-163 (prime = 1).  rcm [n] blocks, corr [n], data0 [n],
 * wire mmo_cot_wire_bytes(...) bytes. */
void mmo_cot_send(const uint64_t *rcm, const uint64_t delta[2], const uint64_t S[2], uint64_t gid0,
                  const uint64_t *corr, size_t n, int l, uint64_t p, int prime, uint64_t *data0, uint8_t *wire) {
  const uint64_t mask = l == 64 ? ~0ull : (1ull << l) - 1;
  const int lb = prime ? prime_bytes(p) : 0;
  uint64_t pad[4 * 8], hashed[4 * 8], corr_data[8], y[8];
  size_t woff = 0;
  for (size_t i = 0; i < n; i += 8) {
    const int bs = (int)(n - i < 8 ? n - i : 8);
    for (int j = 0; j < bs; j++) { /* :104-107 */
      pad[4 * j] = rcm[2 * (i + j)];
      pad[4 * j + 1] = rcm[2 * (i + j) + 1];
      pad[4 * j + 2] = rcm[2 * (i + j)] ^ delta[0];
      pad[4 * j + 3] = rcm[2 * (i + j) + 1] ^ delta[1];
    }
    mmo_mitccrh_hash(S, gid0 + i, pad, (size_t)bs, 2, hashed); /* :109 */
    if (!prime) {
      for (int j = 0; j < bs; j++) { /* :111-116 */
        data0[i + j] = hashed[4 * j] & mask;
        corr_data[j] = (corr[i + j] + data0[i + j] + hashed[4 * j + 2]) & mask;
      }
      const int ysize = (bs * l + 63) / 64; /* :117-123 */
      pack_cot(y, corr_data, ysize, bs, l);
      memcpy(wire + woff, y, 8 * (size_t)ysize);
      woff += 8 * (size_t)ysize;
    } else {
      for (int j = 0; j < bs; j++) { /* :152-158 */
        const unsigned __int128 h0 = ((unsigned __int128)hashed[4 * j + 1] << 64) | hashed[4 * j];
        const unsigned __int128 h1 = ((unsigned __int128)hashed[4 * j + 3] << 64) | hashed[4 * j + 2];
        const unsigned __int128 t0 = h0 % p, t1 = h1 % p;
        data0[i + j] = (uint64_t)t0;
        const uint64_t c = (uint64_t)((t0 + t1 + corr[i + j]) % p);
        memcpy(wire + woff, &c, (size_t)lb);
        woff += (size_t)lb;
      }
    }
  }
}

/* silent_ot.h:168-207 (prime = 0) and :209-242 (prime = 1).  rcm [n] blocks (the receiver's), b [n] */
void mmo_cot_recv(const uint64_t *rcm, const uint8_t *b, const uint64_t S[2], uint64_t gid0, const uint8_t *wire,
                  size_t n, int l, uint64_t p, int prime, uint64_t *data) {
  const uint64_t mask = l == 64 ? ~0ull : (1ull << l) - 1;
  const int lb = prime ? prime_bytes(p) : 0;
  uint64_t hashed[2 * 8], corr_data[8], recvd[8];
  size_t woff = 0;
  for (size_t i = 0; i < n; i += 8) {
    const int bs = (int)(n - i < 8 ? n - i : 8);
    mmo_mitccrh_hash(S, gid0 + i, rcm + 2 * i, (size_t)bs, 1, hashed); /* :191-192 / :223-224 */
    if (!prime) {
      const int rsize = (bs * l + 63) / 64;
      memcpy(recvd, wire + woff, 8 * (size_t)rsize);
      woff += 8 * (size_t)rsize;
      unpack_cot(corr_data, recvd, bs, l);
      for (int j = 0; j < bs; j++) /* :198-204 */
        data[i + j] = (b[i + j] ? corr_data[j] - hashed[2 * j] : hashed[2 * j]) & mask;
    } else {
      for (int j = 0; j < bs; j++) { /* :227-239 */
        uint64_t c = 0;
        memcpy(&c, wire + woff, (size_t)lb);
        woff += (size_t)lb;
        const unsigned __int128 h = ((unsigned __int128)hashed[2 * j + 1] << 64) | hashed[2 * j];
        const unsigned __int128 t = h % p;
        data[i + j] = b[i + j] ? (uint64_t)(((unsigned __int128)c + (p - t)) % p) : (uint64_t)t;
      }
    }
  }
}
