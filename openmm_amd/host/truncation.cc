// mm::Truncation: the five conversion methods of SCI's Truncation over libmm_hip's pre/post
// kernels, with the COT on the host (SCI/src/BuildingBlocks/truncation.cpp:319-695).
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
}  // namespace mm
