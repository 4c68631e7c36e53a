// seal/zstd_decode.h -- a Zstandard decoder (RFC 8878), host code, for the one place the hot path's wire needs it:
// an unmodified reference party saves its ciphertexts and keys with SEAL's default compr_mode_type::zstd
// (cheetah-api.cpp:11-29 calls ct.save(os) without a mode), i.e. a SEALHeader followed by one or more standard
// Zstandard frames.  libzstd is not in this image as a C library, so the loader of seal/wire.h brings its own
// decoder: frames (with or without content size, checksum skipped, skippable frames, no dictionaries), raw / RLE /
// compressed blocks, literals (raw, RLE, Huffman with direct or FSE-compressed weights, one or four streams,
// treeless), sequences (predefined / RLE / FSE / repeat tables, repeat offsets), window across blocks.
// Written from the format specification, not from libzstd's text.  Pinned against a real libzstd: the one inside
// pyarrow compresses the test inputs (tests/test_seal_wire.py: levels 1-19, one-shot and streaming frames, random,
// repetitive, sparse and ciphertext-shaped data, multi-block sizes).  Decoding only; the writer of seal/wire.h
// emits compr_mode none.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace seal {
namespace wire {
namespace zstd {

struct Error : std::runtime_error {
  explicit Error(const std::string &m) : std::runtime_error("zstd: " + m) {}
};

// forward byte reader
struct In {
  const std::uint8_t *p, *end;
  std::size_t left() const { return (std::size_t)(end - p); }
  void need(std::size_t n) const {
    if (left() < n) throw Error("truncated input");
  }
  std::uint64_t le(int bytes) {
    need((std::size_t)bytes);
    std::uint64_t v = 0;
    for (int i = 0; i < bytes; ++i) v |= (std::uint64_t)p[i] << (8 * i);
    p += bytes;
    return v;
  }
};

// forward bit reader (FSE table descriptions): little-endian bit order
struct FwdBits {
  const std::uint8_t *p;
  std::size_t n, bit = 0;
  std::uint32_t read(int bits) {
    std::uint32_t v = 0;
    for (int i = 0; i < bits; ++i, ++bit) {
      if ((bit >> 3) >= n) throw Error("truncated table description");
      v |= (std::uint32_t)((p[bit >> 3] >> (bit & 7)) & 1) << i;
    }
    return v;
  }
  void rewind(int bits) { bit -= (std::size_t)bits; }
  std::size_t bytes_used() const { return (bit + 7) >> 3; }
};

// backward bit reader (Huffman and sequence streams): the stream ends with a 1 bit followed by zero padding; bits are
// taken from the top.  Reading past the start yields zeros (the format allows it for the last symbols) and is
// remembered.
struct BackBits {
  const std::uint8_t *p;
  long long pos;   // number of unread bits
  bool over = false;
  BackBits(const std::uint8_t *data, std::size_t n) : p(data) {
    if (n == 0 || data[n - 1] == 0) throw Error("bitstream without end mark");
    int top = 7;
    while (!((data[n - 1] >> top) & 1)) --top;
    pos = (long long)(n - 1) * 8 + top;
  }
  std::uint64_t read(int bits) {
    std::uint64_t v = 0;
    for (int i = 0; i < bits; ++i) {
      --pos;
      std::uint64_t b = 0;
      if (pos >= 0)
        b = (p[pos >> 3] >> (pos & 7)) & 1;
      else
        over = true;
      v = (v << 1) | b;
    }
    return v;
  }
  // the next `bits` bits without consuming them (zero-filled past the start)
  std::uint32_t peek(int bits) const {
    std::uint32_t v = 0;
    long long q = pos;
    for (int i = 0; i < bits; ++i) {
      --q;
      v = (v << 1) | (q >= 0 ? (std::uint32_t)((p[q >> 3] >> (q & 7)) & 1) : 0u);
    }
    return v;
  }
  void skip(int bits) {
    pos -= bits;
    if (pos < 0) over = true;
  }
};

// ------------------------------------------------------------------------------------------ FSE
struct FseTable {
  int log = 0;
  std::vector<std::uint8_t> symbol, nbits;
  std::vector<std::uint16_t> base;
};

inline int highbit(std::uint32_t v) {
  int r = -1;
  while (v) ++r, v >>= 1;
  return r;
}

// decoding table from normalised counts (-1 = "less than one" probability)
inline void fse_build(FseTable &t, const std::vector<int> &norm, int log) {
  const int size = 1 << log;
  t.log = log;
  t.symbol.assign((std::size_t)size, 0), t.nbits.assign((std::size_t)size, 0), t.base.assign((std::size_t)size, 0);
  std::vector<int> next(norm.size());
  int high = size - 1;
  for (std::size_t s = 0; s < norm.size(); ++s) {
    if (norm[s] == -1) {
      t.symbol[(std::size_t)high--] = (std::uint8_t)s;
      next[s] = 1;
    } else {
      next[s] = norm[s];
    }
  }
  const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  int pos = 0;
  for (std::size_t s = 0; s < norm.size(); ++s)
    for (int i = 0; i < norm[s]; ++i) {
      t.symbol[(std::size_t)pos] = (std::uint8_t)s;
      do pos = (pos + step) & mask;
      while (pos > high);
    }
  if (pos != 0) throw Error("FSE table does not close");
  for (int i = 0; i < size; ++i) {
    const int s = t.symbol[(std::size_t)i], x = next[(std::size_t)s]++;
    const int nb = log - highbit((std::uint32_t)x);
    t.nbits[(std::size_t)i] = (std::uint8_t)nb;
    t.base[(std::size_t)i] = (std::uint16_t)((x << nb) - size);
  }
}

// FSE table description (forward bitstream); returns the bytes consumed
inline std::size_t fse_read(FseTable &t, const std::uint8_t *p, std::size_t n, int max_log, int max_symbols) {
  FwdBits b{p, n};
  const int log = (int)b.read(4) + 5;
  if (log > max_log) throw Error("FSE accuracy log too large");
  int remaining = 1 << log;
  std::vector<int> norm;
  while (remaining > 0 && (int)norm.size() < max_symbols) {
    const int bits = highbit((std::uint32_t)(remaining + 1)) + 1;
    int val = (int)b.read(bits);
    const int lower_mask = (1 << (bits - 1)) - 1, threshold = (1 << bits) - 1 - (remaining + 1);
    if ((val & lower_mask) < threshold) {
      b.rewind(1);
      val &= lower_mask;
    } else if (val > lower_mask) {
      val -= threshold;
    }
    const int prob = val - 1;
    remaining -= prob < 0 ? -prob : prob;
    norm.push_back(prob);
    if (prob == 0) {
      for (;;) {
        const int rep = (int)b.read(2);
        for (int i = 0; i < rep && (int)norm.size() < max_symbols; ++i) norm.push_back(0);
        if (rep != 3) break;
      }
    }
  }
  if (remaining != 0) throw Error("FSE probabilities do not sum to the table size");
  fse_build(t, norm, log);
  return b.bytes_used();
}

inline void fse_rle(FseTable &t, std::uint8_t sym) {
  t.log = 0;
  t.symbol.assign(1, sym), t.nbits.assign(1, 0), t.base.assign(1, 0);
}

// ------------------------------------------------------------------------------------------ Huffman
struct HufTable {
  int max_bits = 0;
  std::vector<std::uint8_t> symbol, nbits;   // indexed by the next max_bits bits
  bool valid = false;
};

inline void huf_build(HufTable &h, const std::vector<std::uint8_t> &weights_in) {
  std::vector<std::uint8_t> w = weights_in;
  std::uint32_t sum = 0;
  for (auto x : w) {
    if (x > 11) throw Error("Huffman weight too large");
    if (x) sum += 1u << (x - 1);
  }
  if (sum == 0) throw Error("empty Huffman tree");
  const int max_bits = highbit(sum) + 1;
  if (max_bits > 11) throw Error("Huffman tree too deep");
  const std::uint32_t left = (1u << max_bits) - sum;
  if (left == 0 || (left & (left - 1))) throw Error("last Huffman weight is not a power of two");
  if (w.size() >= 256) throw Error("too many Huffman symbols");
  w.push_back((std::uint8_t)(highbit(left) + 1));
  h.max_bits = max_bits;
  h.symbol.assign((std::size_t)1 << max_bits, 0), h.nbits.assign((std::size_t)1 << max_bits, 0);
  // codes are handed out by increasing weight (= decreasing length), symbols in order inside a weight
  std::uint32_t code = 0;
  for (int weight = 1; weight <= max_bits; ++weight) {
    const std::uint32_t span = 1u << (weight - 1);
    for (std::size_t s = 0; s < w.size(); ++s)
      if (w[s] == weight) {
        for (std::uint32_t i = 0; i < span; ++i) {
          h.symbol[code + i] = (std::uint8_t)s;
          h.nbits[code + i] = (std::uint8_t)(max_bits + 1 - weight);
        }
        code += span;
      }
  }
  h.valid = true;
}

// Huffman tree description; returns bytes consumed
inline std::size_t huf_read(HufTable &h, const std::uint8_t *p, std::size_t n) {
  if (n < 1) throw Error("truncated Huffman header");
  const int hb = p[0];
  std::vector<std::uint8_t> w;
  std::size_t used;
  if (hb >= 128) {
    const int count = hb - 127;
    used = 1 + (std::size_t)(count + 1) / 2;
    if (n < used) throw Error("truncated Huffman weights");
    for (int i = 0; i < count; ++i) w.push_back((i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4));
  } else {
    used = 1 + (std::size_t)hb;
    if (n < used || hb == 0) throw Error("truncated Huffman weights");
    FseTable t;
    const std::size_t th = fse_read(t, p + 1, (std::size_t)hb, 6, 256);
    if (th >= (std::size_t)hb) throw Error("no room for Huffman weight stream");
    BackBits b(p + 1 + th, (std::size_t)hb - th);
    std::uint32_t s1 = (std::uint32_t)b.read(t.log), s2 = (std::uint32_t)b.read(t.log);
    for (;;) {   // two interleaved states; the stream ends when a state update would run past its start
      w.push_back(t.symbol[s1]);
      if (b.pos < t.nbits[s1]) {
        w.push_back(t.symbol[s2]);
        break;
      }
      s1 = t.base[s1] + (std::uint32_t)b.read(t.nbits[s1]);
      w.push_back(t.symbol[s2]);
      if (b.pos < t.nbits[s2]) {
        w.push_back(t.symbol[s1]);
        break;
      }
      s2 = t.base[s2] + (std::uint32_t)b.read(t.nbits[s2]);
      if (w.size() > 255) throw Error("too many Huffman weights");
    }
  }
  huf_build(h, w);
  return used;
}

inline void huf_stream(const HufTable &h, const std::uint8_t *p, std::size_t n, std::uint8_t *out, std::size_t count) {
  BackBits b(p, n);
  for (std::size_t i = 0; i < count; ++i) {
    const std::uint32_t idx = b.peek(h.max_bits);
    out[i] = h.symbol[idx];
    b.skip(h.nbits[idx]);
  }
  if (b.pos != 0 || b.over) throw Error("Huffman stream does not end where it should");
}

// ------------------------------------------------------------------------------------------ sequences
static const std::uint32_t kLLBase[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,   13,   14,   15,   16,   18,
                                          20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const std::uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                         1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const std::uint32_t kMLBase[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16,  17,  18,  19,  20,
                                          21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,  35,  37,  39,  41,
                                          43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const std::uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                   2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

// what persists from block to block inside a frame
struct FrameState {
  HufTable huf;
  FseTable ll, of, ml;
  bool have_ll = false, have_of = false, have_ml = false;
  std::uint64_t rep[3] = {1, 4, 8};
};

inline void seq_table(FseTable &t, bool &have, int mode, In &in, const int *def, int def_n, int def_log, int max_log,
                      int max_sym) {
  switch (mode) {
    case 0: {
      fse_build(t, std::vector<int>(def, def + def_n), def_log);
      have = true;
      break;
    }
    case 1:
      in.need(1);
      if (*in.p >= max_sym) throw Error("RLE symbol out of range");
      fse_rle(t, *in.p++);
      have = true;
      break;
    case 2:
      in.p += fse_read(t, in.p, in.left(), max_log, max_sym);
      have = true;
      break;
    default:
      if (!have) throw Error("repeat mode without a previous table");
  }
}

inline void decode_block(FrameState &st, const std::uint8_t *src, std::size_t n, std::vector<std::uint8_t> &out,
                         std::size_t frame_start) {
  In in{src, src + n};
  // ---- literals
  in.need(1);
  const int lt = in.p[0] & 3, sf = (in.p[0] >> 2) & 3;
  std::size_t regen, comp = 0;
  int streams = 1;
  if (lt < 2) {
    if (sf == 1) regen = (std::size_t)(in.le(2) >> 4);
    else if (sf == 3) regen = (std::size_t)(in.le(3) >> 4);
    else regen = (std::size_t)(in.le(1) >> 3);
  } else {
    if (sf < 2) {
      const std::uint64_t v = in.le(3);
      regen = (v >> 4) & 0x3FF, comp = (v >> 14) & 0x3FF, streams = sf == 0 ? 1 : 4;
    } else if (sf == 2) {
      const std::uint64_t v = in.le(4);
      regen = (v >> 4) & 0x3FFF, comp = (v >> 18) & 0x3FFF, streams = 4;
    } else {
      const std::uint64_t v = in.le(5);
      regen = (v >> 4) & 0x3FFFF, comp = (v >> 22) & 0x3FFFF, streams = 4;
    }
  }
  if (regen > (std::size_t)128 * 1024) throw Error("literals larger than a block");
  std::vector<std::uint8_t> lit(regen);
  if (lt == 0) {
    in.need(regen);
    std::memcpy(lit.data(), in.p, regen);
    in.p += regen;
  } else if (lt == 1) {
    in.need(1);
    std::memset(lit.data(), *in.p++, regen);
  } else {
    in.need(comp);
    const std::uint8_t *p = in.p;
    std::size_t left = comp;
    if (lt == 2) {
      const std::size_t used = huf_read(st.huf, p, left);
      p += used, left -= used;
    } else if (!st.huf.valid) {
      throw Error("treeless literals without a previous tree");
    }
    if (streams == 1) {
      huf_stream(st.huf, p, left, lit.data(), regen);
    } else {
      if (left < 6) throw Error("truncated jump table");
      const std::size_t s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
      if (6 + s1 + s2 + s3 > left) throw Error("jump table past the literals section");
      const std::size_t s4 = left - 6 - s1 - s2 - s3, per = (regen + 3) / 4;
      if (3 * per > regen) throw Error("four streams for too few literals");
      p += 6;
      huf_stream(st.huf, p, s1, lit.data(), per);
      huf_stream(st.huf, p + s1, s2, lit.data() + per, per);
      huf_stream(st.huf, p + s1 + s2, s3, lit.data() + 2 * per, per);
      huf_stream(st.huf, p + s1 + s2 + s3, s4, lit.data() + 3 * per, regen - 3 * per);
    }
    in.p += comp;
  }
  // ---- sequences
  in.need(1);
  std::size_t nseq = *in.p++;
  if (nseq >= 128) {
    if (nseq == 255) {
      nseq = (std::size_t)in.le(2) + 0x7F00;
    } else {
      in.need(1);
      nseq = ((nseq - 128) << 8) + *in.p++;
    }
  }
  std::size_t lit_pos = 0;
  const std::size_t block_start = out.size();
  constexpr std::size_t kBlockMax = (std::size_t)128 * 1024;   // Block_Maximum_Size: what a block may regenerate
  if (nseq) {
    in.need(1);
    const int modes = *in.p++;
    if (modes & 3) throw Error("reserved bits in the compression modes");
    seq_table(st.ll, st.have_ll, (modes >> 6) & 3, in, kLLDefault, 36, 6, 9, 36);
    seq_table(st.of, st.have_of, (modes >> 4) & 3, in, kOFDefault, 29, 5, 8, 32);
    seq_table(st.ml, st.have_ml, (modes >> 2) & 3, in, kMLDefault, 53, 6, 9, 53);
    BackBits b(in.p, in.left());
    std::uint32_t sl = (std::uint32_t)b.read(st.ll.log), so = (std::uint32_t)b.read(st.of.log),
                  sm = (std::uint32_t)b.read(st.ml.log);
    for (std::size_t i = 0; i < nseq; ++i) {
      const int oc = st.of.symbol[so], mc = st.ml.symbol[sm], lc = st.ll.symbol[sl];
      if (oc > 31 || mc > 52 || lc > 35) throw Error("sequence code out of range");
      const std::uint64_t ov = ((std::uint64_t)1 << oc) + b.read(oc);
      const std::size_t ml = kMLBase[mc] + (std::size_t)b.read(kMLBits[mc]);
      const std::size_t ll = kLLBase[lc] + (std::size_t)b.read(kLLBits[lc]);
      std::uint64_t off;
      if (ov > 3) {
        off = ov - 3;
        st.rep[2] = st.rep[1], st.rep[1] = st.rep[0], st.rep[0] = off;
      } else {
        std::uint64_t idx = ov - 1 + (ll == 0 ? 1 : 0);   // 0..3
        if (idx == 0) {
          off = st.rep[0];
        } else {
          off = idx < 3 ? st.rep[idx] : st.rep[0] - 1;
          if (off == 0) throw Error("zero offset");
          if (idx > 1) st.rep[2] = st.rep[1];
          st.rep[1] = st.rep[0], st.rep[0] = off;
        }
      }
      if (i + 1 < nseq) {
        sl = st.ll.base[sl] + (std::uint32_t)b.read(st.ll.nbits[sl]);
        sm = st.ml.base[sm] + (std::uint32_t)b.read(st.ml.nbits[sm]);
        so = st.of.base[so] + (std::uint32_t)b.read(st.of.nbits[so]);
      }
      if (b.over) throw Error("sequence bitstream exhausted");
      if (lit_pos + ll > lit.size()) throw Error("sequence wants more literals than the block has");
      out.insert(out.end(), lit.begin() + (long)lit_pos, lit.begin() + (long)(lit_pos + ll));
      lit_pos += ll;
      if (off > out.size() - frame_start) throw Error("match offset before the start of the frame");
      if (out.size() - block_start + ml > kBlockMax) throw Error("block expands past the maximum block size");
      const std::size_t from = out.size() - (std::size_t)off;
      for (std::size_t k = 0; k < ml; ++k) {   // byte by byte: overlapping copies repeat the pattern
        const std::uint8_t v = out[from + k];
        out.push_back(v);
      }
    }
    if (b.pos != 0) throw Error("sequence bitstream not consumed");
  }
  out.insert(out.end(), lit.begin() + (long)lit_pos, lit.end());
}

// every frame in [src, src + n) decoded and concatenated
inline std::vector<std::uint8_t> decompress(const std::uint8_t *src, std::size_t n, std::size_t max_out = (std::size_t)1 << 31) {
  std::vector<std::uint8_t> out;
  In in{src, src + n};
  while (in.left()) {
    const std::uint32_t magic = (std::uint32_t)in.le(4);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {   // skippable frame
      const std::size_t len = (std::size_t)in.le(4);
      in.need(len);
      in.p += len;
      continue;
    }
    if (magic != 0xFD2FB528u) throw Error("bad magic number");
    const int fhd = (int)in.le(1);
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 8) throw Error("reserved bit set in the frame header");
    if (!single) (void)in.le(1);   // window descriptor: the output vector is the window
    static const int did_bytes[4] = {0, 1, 2, 4};
    if (did && in.le(did_bytes[did]) != 0) throw Error("dictionaries are not supported");
    static const int fcs_bytes[4] = {0, 2, 4, 8};
    std::uint64_t content = ~(std::uint64_t)0;
    const int fb = fcs_flag == 0 ? (single ? 1 : 0) : fcs_bytes[fcs_flag];
    if (fb) content = in.le(fb) + (fb == 2 ? 256 : 0);
    const std::size_t frame_start = out.size();
    if (content != ~(std::uint64_t)0 && content <= max_out) out.reserve(out.size() + (std::size_t)content);
    FrameState st;
    for (;;) {
      const std::uint32_t bh = (std::uint32_t)in.le(3);
      const int last = bh & 1, type = (bh >> 1) & 3;
      const std::size_t size = bh >> 3;
      if (type == 0) {
        in.need(size);
        out.insert(out.end(), in.p, in.p + size);
        in.p += size;
      } else if (type == 1) {
        in.need(1);
        out.insert(out.end(), size, *in.p++);
      } else if (type == 2) {
        if (size > (std::size_t)128 * 1024) throw Error("compressed block too large");
        in.need(size);
        decode_block(st, in.p, size, out, frame_start);
        in.p += size;
      } else {
        throw Error("reserved block type");
      }
      if (out.size() > max_out) throw Error("output larger than the caller allows");
      if (last) break;
    }
    if (content != ~(std::uint64_t)0 && out.size() - frame_start != content) throw Error("frame content size mismatch");
    if (checksum) (void)in.le(4);   // xxhash64 of the content, low 32 bits: not verified
  }
  return out;
}

}  // namespace zstd
}  // namespace wire
}  // namespace seal
