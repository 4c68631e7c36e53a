// The COT-based helpers and the ReLU of the shipped USE_CHEETAH build over libmm_hip's pre/post kernels:
//  * mm::CotAuxProtocols -- AuxProtocols::multiplexer and B2A (SCI/src/BuildingBlocks/aux-protocols.cpp:62-126) around
//    the COT calls they make on otpack->iknp_straight / iknp_reversed; the millionaire / wrap sub-protocols
//    (out of scope) are forwarded to an inner AuxProtocols, e.g. the ideal double;
//  * mm::ReLURingProtocol -- ReLURingProtocol::relu (SCI/src/NonLinear/relu-ring.h:216-313), same argument order.
// Host arrays in and out like the reference.
#pragma once
#include <cstdint>

#include "mm_gemini/io.h"
#include "mm_gemini/truncation.h"

namespace mm {

class CotAuxProtocols : public AuxProtocols {
 public:
  // straight: ALICE sends, BOB receives; reversed: BOB sends, ALICE receives (OTPack::iknp_straight / iknp_reversed)
  CotAuxProtocols(int party, COT *straight, COT *reversed, AuxProtocols *inner)
      : party(party), straight_(straight), reversed_(reversed), inner_(inner) {}
  int party;
  // aux-protocols.cpp:62-93: y = sel * x on shares, sel boolean-shared, x and y arithmetic mod 2^bw_x / 2^bw_y
  void multiplexer(const uint8_t *sel, const uint64_t *x, uint64_t *y, int32_t size, int32_t bw_x, int32_t bw_y);
  // aux-protocols.cpp:95-126
  void B2A(const uint8_t *x, uint64_t *y, int32_t dim, int32_t bw) override;
  void compare(uint8_t *res, const uint64_t *data, int32_t dim, int32_t bw) override {
    inner_->compare(res, data, dim, bw);
  }
  void MSB_to_Wrap(const uint64_t *x, const uint8_t *msb_x, uint8_t *wrap_x, int32_t dim, int32_t bw) override {
    inner_->MSB_to_Wrap(x, msb_x, wrap_x, dim, bw);
  }
  void msb0_to_wrap(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) override { inner_->msb0_to_wrap(x, w, dim, bw); }
  void msb1_to_wrap(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) override { inner_->msb1_to_wrap(x, w, dim, bw); }
  void wrap_computation(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) override {
    inner_->wrap_computation(x, w, dim, bw);
  }

 private:
  COT *straight_, *reversed_;
  AuxProtocols *inner_;
};

class ReLURingProtocol {
 public:
  ReLURingProtocol(int party, int l, CotAuxProtocols *aux) : party(party), l(l), aux_(aux) {}
  int party, l;
  // relu-ring.h:216: result = share of max(x, 0); drelu_res (optional) receives what the reference writes there,
  // shares of msb(x) (:244-252); skip_ot stops after it (:254-259)
  void relu(uint64_t *result, uint64_t *share, int num_relu, uint8_t *drelu_res = nullptr, bool skip_ot = false);

 private:
  CotAuxProtocols *aux_;
};

}  // namespace mm
