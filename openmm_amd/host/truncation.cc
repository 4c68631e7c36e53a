// mm::Truncation: the five conversion methods of SCI's Truncation over libmm_hip's pre/post
// kernels, with the COT behind the mm::COT interface (SCI/src/BuildingBlocks/truncation.cpp:319-695).
#include "mm_gemini/truncation.h"

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "mm_hip.h"

namespace mm {
namespace {
void must(int rc, const char *what) {
  if (rc != MMH_OK) throw std::runtime_error(std::string(what) + ": " + mmh_last_error());
}
struct Bufs {  // device scratch of one conversion
  uint64_t *in = nullptr, *tmp = nullptr, *corr = nullptr, *y = nullptr, *out = nullptr;
  uint8_t *msb = nullptr;
  explicit Bufs(size_t n, size_t out_words = 1) {
    auto get = [&](size_t bytes) {
      void *p = nullptr;
      must(mmh_malloc(&p, bytes ? bytes : 1), "mmh_malloc");
      return p;
    };
    in = (uint64_t *)get(n * 8), tmp = (uint64_t *)get(n * 8), corr = (uint64_t *)get(n * 8);
    y = (uint64_t *)get(n * 8), out = (uint64_t *)get(n * 8 * out_words), msb = (uint8_t *)get(n);
  }
  ~Bufs() {
    for (void *p : {(void *)in, (void *)tmp, (void *)corr, (void *)y, (void *)out, (void *)msb}) mmh_free(p);
  }
};
void h2d(void *d, const void *h, size_t bytes) { must(mmh_memcpy_h2d(d, h, bytes, nullptr), "h2d"); }
void d2h(void *h, const void *d, size_t bytes) {
  must(mmh_memcpy_d2h(h, d, bytes, nullptr), "d2h");
  must(mmh_stream_sync(nullptr), "sync");
}
}  // namespace

// the one interactive step: ALICE is the sender (truncation.cpp:351, :549, :607, :665)
static void run_cot(int party, COT *cot, Bufs &b, int dim, int l, uint64_t p) {
  std::vector<uint64_t> y(dim);
  if (party == ALICE) {
    std::vector<uint64_t> corr(dim);
    d2h(corr.data(), b.corr, dim * 8ull);
    if (p) cot->send_cot_prime(y.data(), corr.data(), dim, p);
    else cot->send_cot(y.data(), corr.data(), dim, l);
    cot->flush();
  } else {
    std::vector<uint8_t> msb(dim);
    d2h(msb.data(), b.msb, dim);
    static_assert(sizeof(bool) == 1, "choice bits are passed as bool*");
    if (p) cot->recv_cot_prime(y.data(), reinterpret_cast<const bool *>(msb.data()), dim, p);
    else cot->recv_cot(y.data(), reinterpret_cast<const bool *>(msb.data()), dim, l);
  }
  h2d(b.y, y.data(), dim * 8ull);
}

void Truncation::ring_to_field(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t ring_bw, uint64_t field_mod) {
  Bufs b(dim);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_r2f_pre(party, b.in, dim, ring_bw, field_mod, b.tmp, b.msb, b.corr, nullptr), "ring_to_field");
  run_cot(party, cot_, b, dim, 0, field_mod);
  must(mmh_r2f_post(party, b.tmp, b.msb, b.y, dim, ring_bw, field_mod, b.out, nullptr), "ring_to_field");
  d2h(outB, b.out, dim * 8ull);
}

void Truncation::field_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw) {
  Bufs b(dim);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_f2r_pre(party, b.in, dim, field_mod, 0, b.tmp, b.msb, b.corr, nullptr), "field_to_ring");
  run_cot(party, cot_, b, dim, ring_bw, 0);
  must(mmh_f2r_post(party, b.tmp, b.msb, b.y, dim, field_mod, ring_bw, b.out, nullptr), "field_to_ring");
  d2h(outB, b.out, dim * 8ull);
}

void Truncation::field_to_ring_with_truncate(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod,
                                             int32_t ring_bw, int32_t shift_bw) {
  Bufs b(dim);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_f2r_pre(party, b.in, dim, field_mod, shift_bw, b.tmp, b.msb, b.corr, nullptr), "field_to_ring_with_truncate");
  run_cot(party, cot_, b, dim, ring_bw, 0);
  must(mmh_f2r_trunc_post(party, b.tmp, b.msb, b.y, dim, field_mod, ring_bw, shift_bw, b.out, nullptr),
       "field_to_ring_with_truncate");
  d2h(outB, b.out, dim * 8ull);
}

void Truncation::ring_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t bw_A, int32_t bw_B) {
  if (bw_A >= bw_B) {  // truncation.cpp:327-330
    const uint64_t mask = bw_B == 64 ? ~0ULL : (1ULL << bw_B) - 1;
    std::transform(inA, inA + dim, outB, [mask](uint64_t v) { return v & mask; });
    return;
  }
  Bufs b(dim);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_r2r_pre(party, b.in, dim, bw_A, b.tmp, b.msb, b.corr, nullptr), "ring_to_ring");
  run_cot(party, cot_, b, dim, bw_B - bw_A, 0);
  must(mmh_r2r_post(party, b.tmp, b.msb, b.y, dim, bw_A, bw_B, b.out, nullptr), "ring_to_ring");
  d2h(outB, b.out, dim * 8ull);
}

void Truncation::ring_to_ring128(int32_t dim, uint64_t *inA, unsigned __int128 *outB, int32_t bw_A, int32_t bw_B) {
  Bufs b(dim, 2);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_r2r_pre(party, b.in, dim, bw_A, b.tmp, b.msb, b.corr, nullptr), "ring_to_ring128");
  run_cot(party, cot_, b, dim, bw_B - bw_A, 0);
  must(mmh_r2r128_post(party, b.tmp, b.msb, b.y, dim, bw_A, bw_B, b.out, nullptr), "ring_to_ring128");
  d2h(outB, b.out, dim * 16ull);  // (lo, hi) pairs are the little-endian image of __uint128_t
}
// ---- the truncate family (truncation-cheetah.cpp): pre kernel, wrap sub-protocol, B2A, post kernel ----
void Truncation::truncate_common(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw,
                                 bool signed_arithmetic, bool heuristic, Wrap how, const uint8_t *msb_x) {
  if (shift == 0) {  // :8-11
    std::copy(inA, inA + dim, outB);
    return;
  }
  if (!aux_) throw std::runtime_error("Truncation::truncate: no AuxProtocols");
  if (!(bw - shift > 0) || inA == outB) throw std::invalid_argument("Truncation::truncate: bad shift / aliasing");  // :12-14
  Bufs b(dim);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_trunc_pre(party, b.in, dim, bw, signed_arithmetic, heuristic, b.tmp, b.corr, nullptr), "truncate");
  std::vector<uint64_t> adj(dim), cmp(dim), aw(dim);
  std::vector<uint8_t> wrap(dim);
  d2h(adj.data(), b.tmp, dim * 8ull);
  if (how == Wrap::kCompare) {  // :58
    d2h(cmp.data(), b.corr, dim * 8ull);
    aux_->compare(wrap.data(), cmp.data(), dim, bw);
  } else if (how == Wrap::kMsb) {  // :118-127
    std::vector<uint8_t> m(msb_x, msb_x + dim);
    if (signed_arithmetic)
      for (auto &v : m) v ^= party == ALICE ? 1 : 0;
    aux_->MSB_to_Wrap(adj.data(), m.data(), wrap.data(), dim, bw);
  } else if (signed_arithmetic) {  // :192-195
    aux_->msb1_to_wrap(adj.data(), wrap.data(), dim, bw);
  } else {
    aux_->msb0_to_wrap(adj.data(), wrap.data(), dim, bw);
  }
  aux_->B2A(wrap.data(), aw.data(), dim, shift);  // :61
  h2d(b.y, aw.data(), dim * 8ull);
  must(mmh_trunc_post(party, b.tmp, b.y, dim, shift, bw, signed_arithmetic, heuristic, b.out, nullptr), "truncate");
  d2h(outB, b.out, dim * 8ull);
}

void Truncation::truncate(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw,
                          bool signed_arithmetic, uint8_t *msb_x, bool apply_msb0_heuristic) {
  if (msb_x) return truncate_msb(dim, inA, outB, shift, bw, signed_arithmetic, msb_x);  // :5-6
  if (apply_msb0_heuristic)  // :20-39: truncate_msb0 around BOB's +2^(bw-3) / -2^(bw-3-shift)
    return truncate_common(dim, inA, outB, shift, bw, signed_arithmetic, true, Wrap::kMsb0, nullptr);
  truncate_common(dim, inA, outB, shift, bw, signed_arithmetic, false, Wrap::kCompare, nullptr);
}
void Truncation::truncate_msb(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw,
                              bool signed_arithmetic, uint8_t *msb_x) {
  truncate_common(dim, inA, outB, shift, bw, signed_arithmetic, false, Wrap::kMsb, msb_x);
}
void Truncation::truncate_msb0(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw,
                               bool signed_arithmetic) {
  truncate_common(dim, inA, outB, shift, bw, signed_arithmetic, false, Wrap::kMsb0, nullptr);
}

// truncation.cpp:289-317
void Truncation::truncate_and_reduce(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw) {
  if (shift == 0) {
    std::copy(inA, inA + dim, outB);
    return;
  }
  if (!aux_) throw std::runtime_error("Truncation::truncate_and_reduce: no AuxProtocols");
  if (!(bw - shift > 0)) throw std::invalid_argument("Truncation::truncate_and_reduce: bad shift");
  Bufs b(dim);
  h2d(b.in, inA, dim * 8ull);
  must(mmh_trunc_reduce_pre(b.in, dim, shift, b.tmp, nullptr), "truncate_and_reduce");
  std::vector<uint64_t> lower(dim), aw(dim);
  std::vector<uint8_t> wrap(dim);
  d2h(lower.data(), b.tmp, dim * 8ull);
  aux_->wrap_computation(lower.data(), wrap.data(), dim, shift);
  aux_->B2A(wrap.data(), aw.data(), dim, bw - shift);
  h2d(b.y, aw.data(), dim * 8ull);
  must(mmh_trunc_reduce_post(b.in, b.y, dim, shift, bw, b.out, nullptr), "truncate_and_reduce");
  d2h(outB, b.out, dim * 8ull);
}
}  // namespace mm
