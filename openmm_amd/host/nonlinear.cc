// mm::CotAuxProtocols and mm::ReLURingProtocol: the loops of AuxProtocols::multiplexer / B2A
// (SCI/src/BuildingBlocks/aux-protocols.cpp:62-126) and ReLURingProtocol::relu (SCI/src/NonLinear/relu-ring.h:216-313,
// USE_CHEETAH branch) on libmm_hip's kernels, the COTs and the comparison on the host.
#include "mm_gemini/nonlinear.h"

#include <stdexcept>
#include <string>
#include <vector>

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
  uint64_t *u64() const { return static_cast<uint64_t *>(p); }
  uint8_t *u8() const { return static_cast<uint8_t *>(p); }
};
void h2d(void *d, const void *h, size_t bytes) { must(mmh_memcpy_h2d(d, h, bytes, nullptr), "h2d"); }
void d2h(void *h, const void *d, size_t bytes) {
  must(mmh_memcpy_d2h(h, d, bytes, nullptr), "d2h");
  must(mmh_stream_sync(nullptr), "sync");
}
static_assert(sizeof(bool) == 1, "choice bits are passed as bool*");
}  // namespace

void CotAuxProtocols::multiplexer(const uint8_t *sel, const uint64_t *x, uint64_t *y, int32_t size, int32_t bw_x,
                                  int32_t bw_y) {
  if (bw_x > 64 || bw_y > bw_x) throw std::invalid_argument("multiplexer: bw_x <= 64 && bw_y <= bw_x");  // :64
  if (size <= 0) return;
  const size_t n = (size_t)size;
  DevBuf d_sel(n), d_x(n * 8), d_corr(n * 8), d_r(n * 8), d_s(n * 8), d_y(n * 8);
  h2d(d_sel.p, sel, n);
  h2d(d_x.p, x, n * 8);
  must(mmh_mux_pre(d_sel.u8(), d_x.u64(), n, bw_y, d_corr.u64(), nullptr), "multiplexer");   // :76-78
  std::vector<uint64_t> corr(n), data_s(n), data_r(n);
  d2h(corr.data(), d_corr.p, n * 8);
  const bool *b = reinterpret_cast<const bool *>(sel);
  if (party == ALICE) {  // :79-81
    straight_->send_cot(data_s.data(), corr.data(), size, bw_y);
    reversed_->recv_cot(data_r.data(), b, size, bw_y);
  } else {               // :82-85
    straight_->recv_cot(data_r.data(), b, size, bw_y);
    reversed_->send_cot(data_s.data(), corr.data(), size, bw_y);
  }
  h2d(d_r.p, data_r.data(), n * 8);
  h2d(d_s.p, data_s.data(), n * 8);
  must(mmh_mux_post(d_sel.u8(), d_x.u64(), d_r.u64(), d_s.u64(), n, bw_y, d_y.u64(), nullptr), "multiplexer");  // :86-88
  d2h(y, d_y.p, n * 8);
}

void CotAuxProtocols::B2A(const uint8_t *x, uint64_t *y, int32_t dim, int32_t bw) {
  if (dim <= 0) return;
  const size_t n = (size_t)dim;
  DevBuf d_x(n), d_corr(n * 8), d_y(n * 8), d_out(n * 8);
  h2d(d_x.p, x, n);
  if (bw == 1) {  // :97-102
    must(mmh_b2a_post(party, d_x.u8(), nullptr, n, bw, d_out.u64(), nullptr), "B2A");
    d2h(y, d_out.p, n * 8);
    return;
  }
  std::vector<uint64_t> cot_y(n);
  if (party == ALICE) {  // :104-115
    must(mmh_b2a_pre(d_x.u8(), n, bw, d_corr.u64(), nullptr), "B2A");
    std::vector<uint64_t> corr(n);
    d2h(corr.data(), d_corr.p, n * 8);
    straight_->send_cot(cot_y.data(), corr.data(), dim, bw);
  } else {               // :116-122
    straight_->recv_cot(cot_y.data(), reinterpret_cast<const bool *>(x), dim, bw);
  }
  h2d(d_y.p, cot_y.data(), n * 8);
  must(mmh_b2a_post(party, d_x.u8(), d_y.u64(), n, bw, d_out.u64(), nullptr), "B2A");
  d2h(y, d_out.p, n * 8);
}

void ReLURingProtocol::relu(uint64_t *result, uint64_t *share, int num_relu, uint8_t *drelu_res, bool skip_ot) {
  if (num_relu <= 0) return;
  if (l < 2 || l > 64) throw std::invalid_argument("ReLURingProtocol: 2 <= l <= 64");
  const size_t n = (size_t)num_relu;
  DevBuf d_share(n * 8), d_msb(n), d_cmp(n * 8), d_wrap(n), d_drelu(n), d_sel(n);
  h2d(d_share.p, share, n * 8);
  must(mmh_relu_pre(party, d_share.u64(), n, l, d_msb.u8(), d_cmp.u64(), nullptr), "relu");   // :216-238
  std::vector<uint64_t> cmp(n);
  std::vector<uint8_t> wrap(n), sel(n), drelu(n);
  d2h(cmp.data(), d_cmp.p, n * 8);
  aux_->compare(wrap.data(), cmp.data(), num_relu, l - 1);                                    // :243
  h2d(d_wrap.p, wrap.data(), n);
  must(mmh_relu_post(party, d_msb.u8(), d_wrap.u8(), n, d_drelu.u8(), d_sel.u8(), nullptr), "relu");   // :244-252, :308-311
  if (drelu_res) d2h(drelu_res, d_drelu.p, n);
  if (skip_ot) return;                                                                        // :254-259
  d2h(sel.data(), d_sel.p, n);
  aux_->multiplexer(sel.data(), share, result, num_relu, l, l);                               // :312
}

}  // namespace mm
