// mm::SilentOT over mmh_cot_send / mmh_cot_recv (SCI/src/OT/ferret/silent_ot.h:85-242): host arrays in and out
// like the reference; blocks, corrections and the wire cross PCIe once each.
#include "mm_gemini/silent_ot.h"

#include <cstdio>
#include <stdexcept>
#include <string>

#include "mm_hip.h"

namespace mm {
namespace {
void must(int rc, const char *what) {
  if (rc != MMH_OK) throw std::runtime_error(std::string(what) + ": " + mmh_last_error());
}
struct DevBuf {
  void *p = nullptr;
  explicit DevBuf(size_t bytes) { must(mmh_malloc(&p, bytes ? bytes : 1), "mmh_malloc"); }
  ~DevBuf() { mmh_free(p); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};
void h2d(void *d, const void *h, size_t bytes) { must(mmh_memcpy_h2d(d, h, bytes, nullptr), "h2d"); }
void d2h(void *h, const void *d, size_t bytes) {
  must(mmh_memcpy_d2h(h, d, bytes, nullptr), "d2h");
  must(mmh_stream_sync(nullptr), "sync");
}
size_t wire_bytes(int64_t n, int l, uint64_t p) {
  size_t b = 0;
  must(mmh_cot_wire_bytes((size_t)n, l, p, p ? 1 : 0, &b), "mmh_cot_wire_bytes");
  return b;
}
uint64_t gids(int64_t n) {
  uint64_t g = 0;
  must(mmh_cot_gids((size_t)n, &g), "mmh_cot_gids");
  return g;
}
}  // namespace

void SilentOT::fresh_seed(uint64_t seed[2]) {
  if (seeded_) {
    seed[0] = prg_(), seed[1] = prg_();
    return;
  }
  std::FILE *f = std::fopen("/dev/urandom", "rb");
  const bool ok = f && std::fread(seed, 8, 2, f) == 2;
  if (f) std::fclose(f);
  if (!ok) throw std::runtime_error("SilentOT: cannot read /dev/urandom");
}

void SilentOT::send(uint64_t *data0, const uint64_t *corr, int64_t n, int l, uint64_t p) {
  if (n <= 0) return;
  std::vector<Block> rcm(n);
  ferret_->send_ot_rcm_cc(rcm.data(), n);                      // silent_ot.h:89 / :131
  uint64_t seed[2];
  fresh_seed(seed);
  io_->send_data(seed, 16);                                    // :91-95
  const Block dl = ferret_->Delta();
  const uint64_t delta[2] = {dl.lo, dl.hi};
  const size_t nb = wire_bytes(n, l, p);
  DevBuf d_rcm(n * 16ull), d_corr(n * 8ull), d_data0(n * 8ull), d_wire((nb + 7) / 8 * 8);
  h2d(d_rcm.p, rcm.data(), n * 16ull);
  h2d(d_corr.p, corr, n * 8ull);
  must(mmh_cot_send((const uint64_t *)d_rcm.p, delta, seed, gid, (const uint64_t *)d_corr.p, (size_t)n, l, p,
                    p ? 1 : 0, (uint64_t *)d_data0.p, (uint8_t *)d_wire.p, nullptr), "mmh_cot_send");
  gid += gids(n);
  std::vector<uint8_t> wire(nb);
  d2h(wire.data(), d_wire.p, nb);
  d2h(data0, d_data0.p, n * 8ull);
  io_->send_data(wire.data(), nb);                             // :123 / :159
}

void SilentOT::recv(uint64_t *data, const bool *b, int64_t n, int l, uint64_t p) {
  if (n <= 0) return;
  std::vector<Block> rcm(n);
  ferret_->recv_ot_rcm_cc(rcm.data(), b, n);                   // :173 / :211
  uint64_t seed[2];
  io_->recv_data(seed, 16);                                    // :175-176
  const size_t nb = wire_bytes(n, l, p);
  std::vector<uint8_t> wire(nb);
  io_->recv_data(wire.data(), nb);                             // :194 / :226
  DevBuf d_rcm(n * 16ull), d_b(n), d_data(n * 8ull), d_wire((nb + 7) / 8 * 8);
  h2d(d_rcm.p, rcm.data(), n * 16ull);
  static_assert(sizeof(bool) == 1, "choice bits are bytes");
  h2d(d_b.p, b, (size_t)n);
  h2d(d_wire.p, wire.data(), nb);
  must(mmh_cot_recv((const uint64_t *)d_rcm.p, (const uint8_t *)d_b.p, seed, gid, (const uint8_t *)d_wire.p, (size_t)n,
                    l, p, p ? 1 : 0, (uint64_t *)d_data.p, nullptr), "mmh_cot_recv");
  gid += gids(n);
  d2h(data, d_data.p, n * 8ull);
}
}  // namespace mm
