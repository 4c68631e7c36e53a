// Post-processing of the correlated OT on device: SilentOT::send_ot_cam_cc / send_ot_cam_cc_prime /
// recv_ot_cam_cc / recv_ot_cam_cc_prime (SCI/src/OT/ferret/silent_ot.h:85-242) from the blocks Ferret hands
// over to data0 / data and the bytes of the correction message, one launch per side: MITCCRH hash (AES-128
// under one fresh key per element), low 64 bits & mask or 128-bit value mod p, correction, and the wire
// packing of sci::pack_cot_messages / unpack_cot_messages (SCI/src/OT/ot-utils.h:154-216).  Ferret itself
// (send/recv_ot_rcm_cc) and the socket stay on the host.
//
// AES on CDNA4: one element per lane, key schedule on the fly (a key serves one or two blocks only, so round
// keys are never stored), SubBytes+MixColumns through one 256-entry table T0[x] = (2s, s, s, 3s) in LDS whose
// other three rotations are v_alignbit.  The table is kept in 32 interleaved copies, entry x of copy c at
// word 32x + c with c = lane % 32, so that the 32 lanes an LDS cycle serves always hit 32 different banks
// whatever bytes they look up (32 KiB per workgroup; workgroups walk the elements grid-stride so the fill is
// paid once per workgroup).  LDS-bound: 160 lookups per block + 40 per key.
#include <algorithm>

#include "common.hpp"
#include "convert_device.hpp"
#include "sampler_device.hpp"

namespace mmh {
namespace {
struct AesT0 {
  u32 v[256];
};
constexpr u32 rotl8(u32 x, int r) { return ((x << r) | (x >> (8 - r))) & 0xff; }
constexpr u32 xtime(u32 a) { return ((a << 1) ^ ((a & 0x80) ? 0x1b : 0)) & 0xff; }
// S-box by walking the multiplicative group with generator 3 (p) and its inverse (q); FIPS-197 5.1.1
constexpr AesT0 make_t0() {
  AesT0 t{};
  u32 sbox[256] = {};
  u32 p = 1, q = 1;
  do {
    p = (p ^ xtime(p)) & 0xff;
    q ^= q << 1;
    q ^= q << 2;
    q ^= q << 4;
    q &= 0xff;
    if (q & 0x80) q ^= 0x09;
    sbox[p] = (q ^ rotl8(q, 1) ^ rotl8(q, 2) ^ rotl8(q, 3) ^ rotl8(q, 4)) ^ 0x63;
  } while (p != 1);
  sbox[0] = 0x63;
  for (int x = 0; x < 256; x++) {
    const u32 s = sbox[x], s2 = xtime(s);
    t.v[x] = s2 | (s << 8) | (s << 16) | ((s2 ^ s) << 24);
  }
  return t;
}
__device__ const AesT0 kT0 = make_t0();

constexpr int kCopies = 32;
constexpr int kCotThreads = 512;
constexpr int kBatch = 8;  // ot_bsize: elements per hash call and per packed group on the wire

__device__ __forceinline__ u32 rotl(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }
// hipcc leaves a five-term xor as four v_xor_b32
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // v_bitop3_b32 with the truth table of a ^ b ^ c
}

struct Aes {
  const u32 *t;  // LDS table + (lane % 32)
  __device__ __forceinline__ u32 T(u32 byte) const { return t[byte << 5]; }
  __device__ __forceinline__ u32 b0(u32 w) const { return w & 0xff; }
  __device__ __forceinline__ u32 b1(u32 w) const { return (w >> 8) & 0xff; }
  __device__ __forceinline__ u32 b2(u32 w) const { return (w >> 16) & 0xff; }
  __device__ __forceinline__ u32 b3(u32 w) const { return w >> 24; }
  // next round key in place: w[0] ^= SubWord(RotWord(w[3])) ^ rcon, then the running xors
  __device__ __forceinline__ void next_key(u32 (&w)[4], u32 rcon) const {
    const u32 x = w[3];
    const u32 sub = ((T(b1(x)) >> 8) & 0xff) | (T(b2(x)) & 0xff00) | (T(b3(x)) & 0xff0000) |
                    ((T(b0(x)) << 8) & 0xff000000u);
    w[0] ^= sub ^ rcon;
    w[1] ^= w[0];
    w[2] ^= w[1];
    w[3] ^= w[2];
  }
  __device__ __forceinline__ void round(u32 (&s)[4], const u32 (&k)[4]) const {
    u32 o[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
      o[c] = xor3(xor3(T(b0(s[c])), rotl(T(b1(s[(c + 1) & 3])), 8), rotl(T(b2(s[(c + 2) & 3])), 16)),
                  rotl(T(b3(s[(c + 3) & 3])), 24), k[c]);
#pragma unroll
    for (int c = 0; c < 4; c++) s[c] = o[c];
  }
  __device__ __forceinline__ void last_round(u32 (&s)[4], const u32 (&k)[4]) const {
    u32 o[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
      o[c] = (((T(b0(s[c])) >> 8) & 0xff) | (T(b1(s[(c + 1) & 3])) & 0xff00) | (T(b2(s[(c + 2) & 3])) & 0xff0000) |
              ((T(b3(s[(c + 3) & 3])) << 8) & 0xff000000u)) ^
             k[c];
#pragma unroll
    for (int c = 0; c < 4; c++) s[c] = o[c];
  }
  // blk[i] = blk[i] ^ AES_key(blk[i]) for NB blocks under one key (MITCCRH::hash)
  template <int NB>
  __device__ __forceinline__ void hash(u32 (&key)[4], u32 (&blk)[NB][4]) const {
    u32 s[NB][4];
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
      for (int c = 0; c < 4; c++) s[i][c] = blk[i][c] ^ key[c];
    u32 rcon = 1;
#pragma unroll
    for (int r = 1; r <= 10; r++) {
      next_key(key, rcon);
      rcon = r == 8 ? 0x1b : rcon << 1;
#pragma unroll
      for (int i = 0; i < NB; i++) {
        if (r < 10)
          round(s[i], key);
        else
          last_round(s[i], key);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
      for (int c = 0; c < 4; c++) blk[i][c] ^= s[i][c];
  }
};

__device__ __forceinline__ void fill_table(u32 *lds) {
  for (int i = threadIdx.x; i < 256 * kCopies; i += kCotThreads) lds[i] = kT0.v[i >> 5];
  __syncthreads();
}

struct PrimeMod {
  u64 p, r_hi, r_lo;  // floor(2^128 / p)
  int bytes;          // ceil(log2 p / 8): bytes per element on the wire (silent_ot.h:139)
};
// (hi:lo) mod p.  p < 2^63: Barrett with the exact quotient estimate floor(x*ratio/2^128) >= floor(x/p) - 1;
// larger p: shift-and-subtract.
__device__ __forceinline__ u64 mod128(u64 hi, u64 lo, const PrimeMod &m) {
  if (m.p >> 63) {
    u64 r = hi >= m.p ? hi - m.p : hi;
    for (int i = 63; i >= 0; --i) {
      const u64 top = r >> 63;
      r = (r << 1) | ((lo >> i) & 1);
      if (top | (r >= m.p)) r -= m.p;
    }
    return r;
  }
  return barrett128(hi, lo, m.p, m.r_hi, m.r_lo);
}

__device__ __forceinline__ u64 words64(u32 lo, u32 hi) { return make64(lo, hi); }

// A modulus conversion of ALICE's (the COT sender's, truncation.cpp:518-695) run INSIDE the sender's launch: the pre
// step makes the correlation from the share instead of reading it, the post step takes data0 as it is computed, so
// the conversion is one launch instead of three and tmp / msb / corr / data0 never exist in memory.
//   CONV_R2F_ENCODE  ring_to_field (:518-581) whose field elements go straight into BatchEncoder's slot order for
//                    HomBNSS::encodeVector (hom_bn_ss.cc:203-233; the inverse transform follows): out = pts [n_ct][N]
//   CONV_F2R         field_to_ring (:583-637) or, truncating, field_to_ring_with_truncate (:639-695): out = [n]
//   CONV_F2R_DRAW    CONV_F2R on the share -r of HomBNSS::addMask (hom_bn_ss.cc:333-340), r DRAWN here: element e takes
//                    word e of the ChaCha20 stream (key, sid) -- the first 64 lanes of a workgroup compute the 64 blocks
//                    its 512 elements need and hand them over through `stage` -- r = word mod p also goes into
//                    BatchEncoder's slot order (pts_r) and, if asked for, -r into share_out
enum { CONV_NONE = 0, CONV_R2F_ENCODE = 1, CONV_F2R = 2, CONV_F2R_DRAW = 3 };
struct CotConv {
  const u64 *in;
  u64 *out;
  u64 p, cv, big_mod_p;      // r2f: cv = (2^bw - 1) % p + 1, big = 2^(bw-2) % p; f2r: cv = p or p >> shift
  int bw, shift, truncating;
  const u32 *map;            // encode: BatchEncoder's slot map, elements per ciphertext, log2 N, Barrett constants of t = p
  size_t sub_len;
  int logN;
  ModC t;
  ChaChaKey key;             // CONV_F2R_DRAW
  u64 sid;
  u64 *pts_r, *share_out;
};
struct CotSendArgs {
  const uint4 *rcm;
  uint4 delta, S;
  u64 gid0;
  const u64 *corr;
  size_t n;
  int l;
  PrimeMod pm;
  u64 *data0;
  uint8_t *wire;
  CotConv cv;
};
// Sender.  rcm [n] blocks; element j is hashed under key S ^ (gid0 + j in the high half).
// Ring: wire holds, per group of 8 elements, ceil(8*l/64) words (a ragged last group ceil(rem*l/64)).
// Prime: wire holds `bytes` bytes per element, back to back.
// The sender's work of workgroup `block` of `nblocks` (a kernel of its own below, a range of workgroups in
// bn_alice_front_kernel); `table` is filled, `stage` holds kCotThreads words.
template <bool PRIME, int CONV>
__device__ __forceinline__ void cot_send_body(const CotSendArgs &a, unsigned block, unsigned nblocks, const u32 *table,
                                              u64 *stage) {
  const uint4 *__restrict__ rcm = a.rcm;
  const uint4 delta = a.delta, S = a.S;
  const u64 gid0 = a.gid0;
  const u64 *__restrict__ corr = a.corr;
  const size_t n = a.n;
  const int l = a.l;
  const PrimeMod &pm = a.pm;
  u64 *__restrict__ data0 = a.data0;
  uint8_t *__restrict__ wire = a.wire;
  const CotConv &cv = a.cv;
  const Aes aes{table + (threadIdx.x & (kCopies - 1))};
  const u64 mask = l >= 64 ? ~u64(0) : (u64(1) << l) - 1;
  const int wpg = (kBatch * l + 63) / 64;  // words per full group
  const size_t n_chunks = (n + kCotThreads - 1) / kCotThreads;
  for (size_t chunk = block; chunk < n_chunks; chunk += nblocks) {
    const size_t base = chunk * kCotThreads, j = base + threadIdx.x;
    const int live = (int)min((size_t)kCotThreads, n - base);
    u64 cd = 0;
    if (CONV == CONV_F2R_DRAW) {
      __syncthreads();  // the previous chunk's packing is done with stage
      const size_t blk = base / 8 + threadIdx.x;   // (base is a multiple of 8: word e of the stream is element e's)
      if (threadIdx.x < kCotThreads / 8 && blk * 8 < n) {
        u64 w[8];
        chacha20_block(cv.key, cv.sid, blk, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) stage[threadIdx.x * 8 + k] = w[k];
      }
      __syncthreads();
    }
    if (j < n) {
      const uint4 r = rcm[j];
      const u64 gid = gid0 + j;
      u32 key[4] = {S.x, S.y, S.z ^ lo32(gid), S.w ^ hi32(gid)};
      u32 blk[2][4] = {{r.x, r.y, r.z, r.w}, {r.x ^ delta.x, r.y ^ delta.y, r.z ^ delta.z, r.w ^ delta.w}};
      aes.hash<2>(key, blk);
      u64 c, tmpv = 0;
      bool m = false;
      if (CONV == CONV_R2F_ENCODE) {   // truncation.cpp:525-531, :561-564, :546-548 (ALICE)
        tmpv = (cv.in[j] + (u64(1) << (cv.bw - 2))) & bw_mask(cv.bw);
        m = (tmpv >> (cv.bw - 1)) & 1;
        c = m ? 0 : cv.cv;
      } else if (CONV == CONV_F2R) {   // :601-607 / :657-665 (ALICE)
        tmpv = f2r_pre_elem(kAlice, cv.in[j], cv.p, m);
        c = m ? 0 : cv.cv;
      } else if (CONV == CONV_F2R_DRAW) {   // hom_bn_ss.cc:333-340, then as CONV_F2R on the share
        const u64 r = barrett128(0, stage[threadIdx.x], cv.t.q, cv.t.barrett_hi, cv.t.barrett_lo);
        const u64 share = r ? cv.p - r : 0;
        const size_t k = j / cv.sub_len, i = j - k * cv.sub_len;
        cv.pts_r[(k << cv.logN) + cv.map[i]] = r;
        if (cv.share_out) cv.share_out[j] = share;
        tmpv = f2r_pre_elem(kAlice, share, cv.p, m);
        c = m ? 0 : cv.cv;
      } else {
        c = corr[j];
      }
      if (PRIME) {  // silent_ot.h:153-156
        const u64 t0 = mod128(words64(blk[0][2], blk[0][3]), words64(blk[0][0], blk[0][1]), pm);
        const u64 t1 = mod128(words64(blk[1][2], blk[1][3]), words64(blk[1][0], blk[1][1]), pm);
        if (CONV == CONV_R2F_ENCODE) {   // :551-554, :571, then the chunking and slot scatter of encodeVector
          const u64 f = r2f_post_elem(kAlice, tmpv, m, t0, cv.p, cv.cv, cv.big_mod_p);
          const size_t k = j / cv.sub_len, i = j - k * cv.sub_len;
          cv.out[(k << cv.logN) + cv.map[i]] = barrett128(0, f, cv.t.q, cv.t.barrett_hi, cv.t.barrett_lo);
        } else {
          data0[j] = t0;
        }
        // (t0 + t1 + corr) % p over 128 bits: corr is any 64-bit value
        const u64 s = t0 + t1, lo = s + c;
        const u64 hi = (u64)(s < t0) + (u64)(lo < s);
        cd = mod128(hi, lo, pm);
      } else {  // silent_ot.h:112-115
        const u64 d0 = words64(blk[0][0], blk[0][1]) & mask;
        if (CONV == CONV_F2R || CONV == CONV_F2R_DRAW) {   // truncation.cpp:609-611, :626-634 / :667-669, :684-692 (ALICE); mask = 2^bw - 1 = 2^l - 1
          const u64 yy = m ? ((cv.cv - d0) & mask) : ((0 - d0) & mask), big = cv.p / 4;
          const u64 o = cv.truncating ? ((tmpv - big) >> cv.shift) - yy : tmpv - yy - big;
          cv.out[j] = o & mask;
        } else {
          data0[j] = d0;
        }
        cd = (c + d0 + words64(blk[1][0], blk[1][1])) & mask;
      }
    }
    __syncthreads();  // previous chunk's readers are done with stage
    stage[threadIdx.x] = cd;
    __syncthreads();
    if (PRIME) {
      // live*bytes bytes starting at base*bytes (a multiple of 4): whole 32-bit words, then the last bytes
      const int nbytes = live * pm.bytes;
      uint8_t *dst = wire + base * (size_t)pm.bytes;
      const uint8_t *sb = reinterpret_cast<const uint8_t *>(stage);
      for (int w = threadIdx.x; w < (nbytes + 3) / 4; w += kCotThreads) {
        u32 word = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int byte = 4 * w + k;
          if (byte < nbytes) {
            const int e = byte / pm.bytes, o = byte - e * pm.bytes;
            word |= (u32)sb[8 * e + o] << (8 * k);
          }
        }
        if (4 * w + 4 <= nbytes) {
          reinterpret_cast<u32 *>(dst)[w] = word;
        } else {
          for (int k = 0; 4 * w + k < nbytes; k++) dst[4 * w + k] = (uint8_t)(word >> (8 * k));
        }
      }
    } else {
      // ot-utils.h:154-186: word w of group g gathers the elements whose l bits overlap [64w, 64w+64)
      const int groups = (live + kBatch - 1) / kBatch;
      u64 *dst = reinterpret_cast<u64 *>(wire) + (base / kBatch) * (size_t)wpg;
      for (int t = threadIdx.x; t < groups * wpg; t += kCotThreads) {
        const int g = t / wpg, w = t - g * wpg;
        const int bs = min(kBatch, live - g * kBatch);
        if (w >= (bs * l + 63) / 64) continue;  // ragged last group is shorter
        u64 word = 0;
        const int first = (64 * w) / l;
        for (int i = first; i < bs && i * l < 64 * w + 64; i++) {
          const u64 v = stage[g * kBatch + i];
          const int start = i * l - 64 * w;  // bit position of the element relative to this word
          word ^= start >= 0 ? v << start : v >> (-start);
        }
        dst[(size_t)g * wpg + w] = word;
      }
    }
  }
}

template <bool PRIME, int CONV = CONV_NONE>
__global__ __launch_bounds__(kCotThreads) void cot_send_kernel(const uint4 *__restrict__ rcm, uint4 delta, uint4 S,
                                                                u64 gid0, const u64 *__restrict__ corr, size_t n,
                                                                int l, PrimeMod pm, u64 *__restrict__ data0,
                                                                uint8_t *__restrict__ wire, CotConv cv = CotConv{}) {
  __shared__ u32 table[256 * kCopies];
  __shared__ u64 stage[kCotThreads];
  fill_table(table);
  const CotSendArgs a{rcm, delta, S, gid0, corr, n, l, pm, data0, wire, cv};
  cot_send_body<PRIME, CONV>(a, blockIdx.x, gridDim.x, table, stage);
}

// ALICE's work on a batch-norm layer that needs no ciphertext, as ONE launch (mmh_bn_alice_prepare, fc_bn.hip): workgroups
// [0, nb_a) are the sender of ring_to_field feeding encodeVector (CONV_R2F_ENCODE), the next nb_b the sender of
// field_to_ring[_with_truncate] on the share -r they draw (CONV_F2R_DRAW), the rest chunk-scatter the model's scales
// into BatchEncoder's slot order (encodeVector, hom_bn_ss.cc:203-233) and write the zeros of the three batches' slots
// past the vector's end.  The three are independent and each alone is too small to fill the chip.
struct BnScalesArgs {
  const u64 *scales;
  u64 *pts_in, *pts_scales, *pts_r;
  size_t n_ct, n, sub_len;
  int logN;
  const u32 *map;
  ModC t;
};
__global__ __launch_bounds__(kCotThreads) void bn_alice_front_kernel(CotSendArgs a, CotSendArgs b, unsigned nb_a,
                                                                      unsigned nb_b, BnScalesArgs sc) {
  __shared__ u32 table[256 * kCopies];
  __shared__ u64 stage[kCotThreads];
  if (blockIdx.x >= nb_a + nb_b) {
    const size_t idx = (size_t)(blockIdx.x - nb_a - nb_b) * kCotThreads + threadIdx.x;
    if (idx >= (sc.n_ct << sc.logN)) return;
    const size_t k = idx >> sc.logN, i = idx & ((size_t(1) << sc.logN) - 1), src = k * sc.sub_len + i;
    const size_t slot = (k << sc.logN) + sc.map[i];
    if (i < sc.sub_len && src < sc.n) {
      sc.pts_scales[slot] = barrett128(0, sc.scales[src], sc.t.q, sc.t.barrett_hi, sc.t.barrett_lo);
    } else {
      sc.pts_scales[slot] = 0, sc.pts_in[slot] = 0, sc.pts_r[slot] = 0;
    }
    return;
  }
  fill_table(table);
  if (blockIdx.x < nb_a) cot_send_body<true, CONV_R2F_ENCODE>(a, blockIdx.x, nb_a, table, stage);
  else cot_send_body<false, CONV_F2R_DRAW>(b, blockIdx.x - nb_a, nb_b, table, stage);
}

// Receiver.  rcm [n] blocks (its own), b [n] choice bits as bytes.
template <bool PRIME>
__global__ __launch_bounds__(kCotThreads) void cot_recv_kernel(const uint4 *__restrict__ rcm,
                                                                const uint8_t *__restrict__ b, uint4 S, u64 gid0,
                                                                const uint8_t *__restrict__ wire, size_t n, int l,
                                                                PrimeMod pm, u64 *__restrict__ data) {
  __shared__ u32 table[256 * kCopies];
  fill_table(table);
  const Aes aes{table + (threadIdx.x & (kCopies - 1))};
  const u64 mask = l >= 64 ? ~u64(0) : (u64(1) << l) - 1;
  const int wpg = (kBatch * l + 63) / 64;
  const size_t n_chunks = (n + kCotThreads - 1) / kCotThreads;
  for (size_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const size_t j = chunk * kCotThreads + threadIdx.x;
    if (j >= n) continue;
    const uint4 r = rcm[j];
    const u64 gid = gid0 + j;
    u32 key[4] = {S.x, S.y, S.z ^ lo32(gid), S.w ^ hi32(gid)};
    u32 blk[1][4] = {{r.x, r.y, r.z, r.w}};
    aes.hash<1>(key, blk);
    const bool choice = b[j] != 0;
    if (PRIME) {  // silent_ot.h:227-239
      u64 c = 0;
      const uint8_t *src = wire + j * (size_t)pm.bytes;
      for (int k = 0; k < pm.bytes; k++) c |= (u64)src[k] << (8 * k);
      const u64 t = mod128(words64(blk[0][2], blk[0][3]), words64(blk[0][0], blk[0][1]), pm);
      u64 out = t;
      if (choice) {  // (c + (p - t)) % p over 128 bits: c is whatever the wire held
        const u64 d = pm.p - t, lo = c + d;
        out = mod128((u64)(lo < c), lo, pm);
      }
      data[j] = out;
    } else {  // ot-utils.h:188-216, silent_ot.h:198-204
      const u64 h = words64(blk[0][0], blk[0][1]);
      const u64 *grp = reinterpret_cast<const u64 *>(wire) + (j / kBatch) * (size_t)wpg;
      const int start = (int)(j % kBatch) * l, sb = start >> 6, eb = (start + l - 1) >> 6, sh = start & 63;
      u64 c = grp[sb] >> sh;
      if (eb != sb) c ^= grp[eb] << (64 - sh);
      data[j] = (choice ? c - h : h) & mask;
    }
  }
}

int prime_bytes(u64 p) {
  int bits = 0;
  while (bits < 64 && (p >> bits) > 1) bits++;
  if (p & (p - 1)) bits++;
  return (bits + 7) / 8;
}
PrimeMod make_prime(u64 p) {
  PrimeMod m{};
  m.p = p;
  if (p >= 2) {
    using u128 = unsigned __int128;
    const u128 ratio = p & (p - 1) ? ~(u128)0 / p : (~(u128)0 / p) + 1;
    m.r_hi = (u64)(ratio >> 64);
    m.r_lo = (u64)ratio;
    m.bytes = prime_bytes(p);
  }
  return m;
}
unsigned cot_grid(size_t n) {
  const size_t chunks = (n + kCotThreads - 1) / kCotThreads;
  return (unsigned)std::min<size_t>(chunks, 256 * 4);  // up to 4 workgroups of 34 KiB LDS per CU
}
uint4 as_block(const uint64_t v[2]) { return uint4{(u32)v[0], (u32)(v[0] >> 32), (u32)v[1], (u32)(v[1] >> 32)}; }
}  // namespace
// ALICE's ring_to_field through the sender's launch, field elements scattered into BatchEncoder's slot order (used by
// mmh_cot_send_r2f_encode, fc_bn.hip, which runs the inverse transform behind it).  pts must hold zeros where no
// element lands (a ragged last ciphertext): the caller clears those.
int cot_send_r2f_encode_launch(const u64 *d_rcm, const uint64_t delta[2], const uint64_t seed[2], u64 gid0, const u64 *d_x,
                               size_t n, int bw, u64 p, u64 *d_pts, const u32 *d_map, size_t sub_len, int logN,
                               const ModC &t, uint8_t *d_wire, hipStream_t s) {
  const u64 mask = bw >= 64 ? ~u64(0) : (u64(1) << bw) - 1;
  CotConv cv{};
  cv.in = d_x, cv.out = d_pts, cv.p = p, cv.cv = mask % p + 1, cv.big_mod_p = (u64(1) << (bw - 2)) % p, cv.bw = bw;
  cv.map = d_map, cv.sub_len = sub_len, cv.logN = logN, cv.t = t;
  hipLaunchKernelGGL((cot_send_kernel<true, CONV_R2F_ENCODE>), dim3(cot_grid(n)), dim3(kCotThreads), 0, s,
                     (const uint4 *)d_rcm, as_block(delta), as_block(seed), gid0, (const u64 *)nullptr, n, 64, make_prime(p),
                     (u64 *)nullptr, d_wire, cv);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
// mmh_bn_alice_prepare's one launch (see bn_alice_front_kernel); gid0 is the first conversion's, the second follows it
int bn_alice_front_launch(const u64 *d_rcm_r2f, const u64 *d_rcm_f2r, const uint64_t delta[2], const uint64_t seed_r2f[2],
                          const uint64_t seed_f2r[2], u64 gid0, const u64 *d_x, size_t n, int bw, int shift, u64 p,
                          const u64 *d_scales, const uint32_t key[8], u64 sid, u64 *d_pts3, size_t n_ct, const u32 *d_map,
                          size_t sub_len, int logN, const ModC &t, u64 *d_share, u64 *d_out_ring, uint8_t *d_wire_r2f,
                          uint8_t *d_wire_f2r, hipStream_t s) {
  const u64 mask = bw >= 64 ? ~u64(0) : (u64(1) << bw) - 1;
  const size_t N = size_t(1) << logN;
  u64 *pts_in = d_pts3, *pts_sc = d_pts3 + n_ct * N, *pts_r = d_pts3 + 2 * n_ct * N;
  CotSendArgs a{}, b{};
  a.rcm = (const uint4 *)d_rcm_r2f, a.delta = as_block(delta), a.S = as_block(seed_r2f), a.gid0 = gid0, a.n = n, a.l = 64;
  a.pm = make_prime(p), a.wire = d_wire_r2f;
  a.cv.in = d_x, a.cv.out = pts_in, a.cv.p = p, a.cv.cv = mask % p + 1, a.cv.big_mod_p = (u64(1) << (bw - 2)) % p;
  a.cv.bw = bw, a.cv.map = d_map, a.cv.sub_len = sub_len, a.cv.logN = logN, a.cv.t = t;
  b.rcm = (const uint4 *)d_rcm_f2r, b.delta = a.delta, b.S = as_block(seed_f2r);
  b.gid0 = gid0 + kBatch * (u64)((n + kBatch - 1) / kBatch), b.n = n, b.l = bw, b.pm = make_prime(2), b.wire = d_wire_f2r;
  b.cv.out = d_out_ring, b.cv.p = p, b.cv.truncating = shift >= 0, b.cv.shift = shift >= 0 ? shift : 0;
  b.cv.cv = shift >= 0 ? p >> shift : p, b.cv.bw = bw, b.cv.map = d_map, b.cv.sub_len = sub_len, b.cv.logN = logN, b.cv.t = t;
  for (int i = 0; i < 8; ++i) b.cv.key.k[i] = key[i];
  b.cv.sid = sid, b.cv.pts_r = pts_r, b.cv.share_out = d_share;
  const BnScalesArgs sc{d_scales, pts_in, pts_sc, pts_r, n_ct, n, sub_len, logN, d_map, t};
  const unsigned nb = cot_grid(n), nb_sc = (unsigned)(((n_ct << logN) + kCotThreads - 1) / kCotThreads);
  hipLaunchKernelGGL(bn_alice_front_kernel, dim3(nb + nb + nb_sc), dim3(kCotThreads), 0, s, a, b, nb, nb, sc);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
}  // namespace mmh

using namespace mmh;

extern "C" int mmh_cot_wire_bytes(size_t n, int l, uint64_t p, int prime, size_t *bytes) {
  MMH_REQUIRE(bytes, MMH_ERR_NULL_POINTER);
  if (prime) {
    MMH_REQUIRE(p >= 2, MMH_ERR_INVALID_ARG);
    *bytes = n * (size_t)prime_bytes(p);
    return MMH_OK;
  }
  MMH_REQUIRE(l >= 1 && l <= 64, MMH_ERR_INVALID_ARG);
  const size_t full = n / kBatch, rem = n % kBatch;
  *bytes = 8 * (full * (size_t)((kBatch * l + 63) / 64) + (rem * (size_t)l + 63) / 64);
  return MMH_OK;
}

extern "C" int mmh_cot_gids(size_t n, uint64_t *gids) {
  MMH_REQUIRE(gids, MMH_ERR_NULL_POINTER);
  *gids = kBatch * (uint64_t)((n + kBatch - 1) / kBatch);
  return MMH_OK;
}

static int cot_check(size_t n, int l, uint64_t p, int prime) {
  MMH_REQUIRE(n <= ((size_t)1 << 40), MMH_ERR_OUT_BOUND);
  if (prime)
    MMH_REQUIRE(p >= 2, MMH_ERR_INVALID_ARG);
  else
    MMH_REQUIRE(l >= 1 && l <= 64, MMH_ERR_INVALID_ARG);
  return MMH_OK;
}

extern "C" int mmh_cot_send(const uint64_t *d_rcm, const uint64_t delta[2], const uint64_t seed[2], uint64_t gid0,
                            const uint64_t *d_corr, size_t n, int l, uint64_t p, int prime, uint64_t *d_data0,
                            uint8_t *d_wire, void *stream) {
  MMH_REQUIRE(delta && seed, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(n == 0 || (d_rcm && d_corr && d_data0 && d_wire), MMH_ERR_NULL_POINTER);
  if (int rc = cot_check(n, l, p, prime)) return rc;
  if (n == 0) return MMH_OK;
  const PrimeMod pm = make_prime(prime ? p : 2);
  const dim3 grid(cot_grid(n)), block(kCotThreads);
  if (prime)
    hipLaunchKernelGGL(cot_send_kernel<true>, grid, block, 0, (hipStream_t)stream, (const uint4 *)d_rcm,
                       as_block(delta), as_block(seed), gid0, d_corr, n, 64, pm, d_data0, d_wire);
  else
    hipLaunchKernelGGL(cot_send_kernel<false>, grid, block, 0, (hipStream_t)stream, (const uint4 *)d_rcm,
                       as_block(delta), as_block(seed), gid0, d_corr, n, l, pm, d_data0, d_wire);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_cot_recv(const uint64_t *d_rcm, const uint8_t *d_choice, const uint64_t seed[2], uint64_t gid0,
                            const uint8_t *d_wire, size_t n, int l, uint64_t p, int prime, uint64_t *d_data,
                            void *stream) {
  MMH_REQUIRE(seed, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(n == 0 || (d_rcm && d_choice && d_wire && d_data), MMH_ERR_NULL_POINTER);
  if (int rc = cot_check(n, l, p, prime)) return rc;
  if (n == 0) return MMH_OK;
  const PrimeMod pm = make_prime(prime ? p : 2);
  const dim3 grid(cot_grid(n)), block(kCotThreads);
  if (prime)
    hipLaunchKernelGGL(cot_recv_kernel<true>, grid, block, 0, (hipStream_t)stream, (const uint4 *)d_rcm, d_choice,
                       as_block(seed), gid0, d_wire, n, 64, pm, d_data);
  else
    hipLaunchKernelGGL(cot_recv_kernel<false>, grid, block, 0, (hipStream_t)stream, (const uint4 *)d_rcm, d_choice,
                       as_block(seed), gid0, d_wire, n, l, pm, d_data);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

// ALICE's field_to_ring (shift < 0) / field_to_ring_with_truncate (shift >= 0) as the COT sender in ONE launch:
// mmh_f2r_pre, mmh_cot_send (mod 2^bw) and mmh_f2r_post / mmh_f2r_trunc_post give the same d_out and the same wire.
extern "C" int mmh_cot_send_f2r(const uint64_t *d_rcm, const uint64_t delta[2], const uint64_t seed[2], uint64_t gid0,
                                const uint64_t *d_share, size_t n, uint64_t p, int bw, int shift, uint64_t *d_out,
                                uint8_t *d_wire, void *stream) {
  MMH_REQUIRE(delta && seed, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(n == 0 || (d_rcm && d_share && d_out && d_wire), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 1 && bw <= 64 && p >= 4 && shift < 64, MMH_ERR_INVALID_ARG);
  if (int rc = cot_check(n, bw, p, 0)) return rc;
  if (n == 0) return MMH_OK;
  CotConv cv{};
  cv.in = d_share, cv.out = d_out, cv.p = p, cv.truncating = shift >= 0, cv.shift = shift >= 0 ? shift : 0;
  cv.cv = shift >= 0 ? p >> shift : p, cv.bw = bw;
  hipLaunchKernelGGL((cot_send_kernel<false, CONV_F2R>), dim3(cot_grid(n)), dim3(kCotThreads), 0, (hipStream_t)stream,
                     (const uint4 *)d_rcm, as_block(delta), as_block(seed), gid0, (const uint64_t *)nullptr, n, bw,
                     make_prime(2), (uint64_t *)nullptr, d_wire, cv);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
