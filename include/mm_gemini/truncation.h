// The modulus-conversion methods of SCI's Truncation class (SCI/src/BuildingBlocks/truncation.h:29-210)
// over libmm_hip: same method names and argument order; host arrays in and out like the reference.
#pragma once
#include <cstdint>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <vector>

#include "mm_gemini/io.h"

namespace mm {
const int ALICE = 1, BOB = 2;  // SCI/src/utils/constants.h:40-41

// The OT-based sub-protocols Truncation::truncate* call (SCI/src/BuildingBlocks/aux-protocols.h:33-170 and
// Millionaire::compare): out of scope here, reached through this interface.  Each call returns this party's
// boolean or arithmetic share.
class AuxProtocols {
 public:
  virtual ~AuxProtocols() = default;
  // mill->compare: shares of 1{ALICE's data > BOB's data} on bw-bit values
  virtual void compare(uint8_t *res, const uint64_t *data, int32_t dim, int32_t bw) = 0;
  // shares of the wrap bit 1{x_A + x_B >= 2^bw} from shares of the most significant bit / under the msb = 0 /
  // msb = 1 promise / of shift-bit low parts
  virtual void MSB_to_Wrap(const uint64_t *x, const uint8_t *msb_x, uint8_t *wrap_x, int32_t dim, int32_t bw) = 0;
  virtual void msb0_to_wrap(const uint64_t *x, uint8_t *wrap_x, int32_t dim, int32_t bw) = 0;
  virtual void msb1_to_wrap(const uint64_t *x, uint8_t *wrap_x, int32_t dim, int32_t bw) = 0;
  virtual void wrap_computation(const uint64_t *x, uint8_t *wrap_x, int32_t dim, int32_t bw) = 0;
  // boolean shares -> arithmetic shares mod 2^bw
  virtual void B2A(const uint8_t *x, uint64_t *y, int32_t dim, int32_t bw) = 0;
};

// Ideal versions over a shared in-process mailbox; the two parties make the same calls in the same order.
class IdealAux : public AuxProtocols {
 public:
  struct Mailbox {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::vector<uint64_t>> to[3];   // indexed by the receiving party
  };
  IdealAux(int party, std::shared_ptr<Mailbox> box, uint64_t seed) : party_(party), box_(box), rng_(seed) {}
  void compare(uint8_t *res, const uint64_t *data, int32_t dim, int32_t bw) override {
    auto other = swap(std::vector<uint64_t>(data, data + dim));
    for (int i = 0; i < dim; ++i) {
      const uint64_t a = party_ == 1 ? data[i] : other[i], b = party_ == 1 ? other[i] : data[i];
      res[i] = share_bit(a > b);
    }
    (void)bw;
  }
  void MSB_to_Wrap(const uint64_t *x, const uint8_t *, uint8_t *w, int32_t dim, int32_t bw) override { wrap(x, w, dim, bw); }
  void msb0_to_wrap(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) override { wrap(x, w, dim, bw); }
  void msb1_to_wrap(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) override { wrap(x, w, dim, bw); }
  void wrap_computation(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) override { wrap(x, w, dim, bw); }
  void B2A(const uint8_t *x, uint64_t *y, int32_t dim, int32_t bw) override {
    auto other = swap(std::vector<uint64_t>(x, x + dim));
    const uint64_t mask = bw >= 64 ? ~uint64_t(0) : (uint64_t(1) << bw) - 1;
    for (int i = 0; i < dim; ++i) {
      const uint64_t bit = (x[i] ^ other[i]) & 1, r = rng_() & mask;   // both parties draw the same r
      y[i] = party_ == 1 ? r : (bit - r) & mask;
    }
  }

 private:
  void wrap(const uint64_t *x, uint8_t *w, int32_t dim, int32_t bw) {
    auto other = swap(std::vector<uint64_t>(x, x + dim));
    const uint64_t mask = bw >= 64 ? ~uint64_t(0) : (uint64_t(1) << bw) - 1;
    for (int i = 0; i < dim; ++i) {
      const unsigned __int128 sum = (unsigned __int128)(x[i] & mask) + (other[i] & mask);
      w[i] = share_bit(sum > mask);
    }
  }
  uint8_t share_bit(bool bit) {
    const uint8_t r = rng_() & 1;
    return party_ == 1 ? r : (uint8_t)(bit ^ r);
  }
  std::vector<uint64_t> swap(std::vector<uint64_t> mine) {
    const int other = party_ == 1 ? 2 : 1;
    std::unique_lock<std::mutex> g(box_->m);
    box_->to[other].push_back(std::move(mine));
    box_->cv.notify_all();
    box_->cv.wait(g, [&] { return !box_->to[party_].empty(); });
    auto got = std::move(box_->to[party_].front());
    box_->to[party_].pop_front();
    return got;
  }
  int party_;
  std::shared_ptr<Mailbox> box_;
  std::mt19937_64 rng_;
};

class Truncation {
 public:
  Truncation(int party, COT *cot, AuxProtocols *aux = nullptr) : party(party), cot_(cot), aux_(aux) {}
  int party;
  // truncation.cpp:319, :383, :518, :583, :639.  128-bit output as (lo, hi) word pairs.
  void ring_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t bw_A, int32_t bw_B);
  void ring_to_ring128(int32_t dim, uint64_t *inA, unsigned __int128 *outB, int32_t bw_A, int32_t bw_B);
  void ring_to_field(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t ring_bw, uint64_t field_mod);
  void field_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw);
  void field_to_ring_with_truncate(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw,
                                   int32_t shift_bw);
  // The truncate family of the shipped USE_CHEETAH build (truncation-cheetah.cpp:2-218; signatures and defaults of
  // truncation.h:45-67, :131-163) and truncate_and_reduce (truncation.cpp:289-317); they need `aux`.
  void truncate(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw, bool signed_arithmetic = true,
                uint8_t *msb_x = nullptr, bool apply_msb0_heuristic = true);
  void truncate_msb(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw, bool signed_arithmetic,
                    uint8_t *msb_x);
  void truncate_msb0(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw,
                     bool signed_arithmetic = true);
  void truncate_and_reduce(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw);

 private:
  enum class Wrap { kCompare, kMsb, kMsb0 };
  void truncate_common(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t shift, int32_t bw, bool signed_arithmetic,
                       bool heuristic, Wrap how, const uint8_t *msb_x);
  COT *cot_;
  AuxProtocols *aux_;
};
}  // namespace mm
