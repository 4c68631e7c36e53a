// sci_shim/BuildingBlocks/truncation.h -- what the reference's own acceptance programs
// (tests/{ring_to_field,field_to_ring,ring_extension,bole}_test.cpp) need from
// SCI/src/BuildingBlocks/truncation.h, so that they compile FROM WHERE THEY LIE, unchanged, with
//     -I include/sci_shim -I $REF/SCI/src ...      (this directory first: it shadows the reference's header)
// and run the five modulus conversions on the MI355X library:
//   * class Truncation with the reference's constructor (truncation.h:29-52) and the five conversion methods
//     (truncation.cpp:319-695) forwarding to mm::Truncation (libmm_gemini.so: pre / post kernels around one COT);
//   * sci::OTPack<sci::NetIO>(io, party) and AuxProtocols(party, io, otpack) with the constructors the programs call.
//     OTPack::iknp_straight is the object the reference's Truncation calls send/recv_cot[_prime] and io->flush() on
//     (truncation.cpp:351-352,:549-550): here an mm::SilentOT -- the reference's SilentOT post-processing
//     (SCI/src/OT/ferret/silent_ot.h:85-242: hash, mask / mod p, correction, wire packing) on the device -- over an
//     IDEAL random-correlated-OT source in the style of SCI/src/OT/ideal.h:106-156: the sender draws the blocks and
//     ships them in the clear.  Ferret itself (emp-ot, absent from the reference tree) is out of scope: the stub gives
//     the functionality, not the security, and its 16 bytes per element show up in the programs' byte counts.
// Everything else the reference header declares (truncate*, div_pow2, the Millionaire members) is OT-protocol code
// outside the path and is not offered here.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <random>
#include <stdexcept>

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <iostream>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "mm_gemini/silent_ot.h"
#include "mm_gemini/truncation.h"
#include "utils/emp-tool.h"  // the reference's own: sci::NetIO, sci::PRG128, sci::ALICE / BOB, default_prime_mod

namespace sci {
namespace shim {
// mm::IOChannel over the reference's sci::NetIO (the dozen lines include/mm_gemini/io.h speaks of)
class NetChannel : public mm::IOChannel {
 public:
  explicit NetChannel(sci::NetIO *io) : io_(io) {}
  void send_data(const void *data, size_t nbytes) override { io_->send_data(data, (int)nbytes), counter += nbytes; }
  void recv_data(void *data, size_t nbytes) override { io_->recv_data(data, (int)nbytes); }
  void flush() override { io_->flush(); }

 private:
  sci::NetIO *io_;
};
// Ideal random correlated OT between two processes (pattern SCI/src/OT/ideal.h:106-156): the sender draws n random
// blocks and sends them; the receiver takes block_j ^ b_j * Delta.  Delta is a constant both stubs know.
class IdealFerretOverNet : public mm::RandomCOT {
 public:
  IdealFerretOverNet(sci::NetIO *io, uint64_t seed) : io_(io), rng_(seed) {}
  mm::Block Delta() const override { return mm::Block{0x9E3779B97F4A7C15ull | 1, 0xD1B54A32D192ED03ull}; }
  void send_ot_rcm_cc(mm::Block *data0, int64_t n) override {
    for (int64_t i = 0; i < n; ++i) data0[i] = mm::Block{rng_(), rng_()};
    for (int64_t at = 0; at < n; at += kChunk) io_->send_data(data0 + at, (int)(std::min<int64_t>(kChunk, n - at) * 16));
    io_->flush();
  }
  void recv_ot_rcm_cc(mm::Block *data, const bool *b, int64_t n) override {
    for (int64_t at = 0; at < n; at += kChunk) io_->recv_data(data + at, (int)(std::min<int64_t>(kChunk, n - at) * 16));
    const mm::Block d = Delta();
    for (int64_t i = 0; i < n; ++i)
      if (b[i]) data[i].lo ^= d.lo, data[i].hi ^= d.hi;
  }

 private:
  static constexpr int64_t kChunk = 1 << 24;  // NetIO takes int lengths
  sci::NetIO *io_;
  std::mt19937_64 rng_;
};
}  // namespace shim

template <typename IO>
class IKNP;  // (the programs declare a pointer to it and never use it)

template <typename IO>
class OTPack {
 public:
  // cheetah-ot_pack.h: OTPack(io, party): silent OT set up over `io`; iknp_straight's sender is ALICE
  OTPack(IO *io, int party) : io(io), party(party), channel_(io), ferret_(io, 0x4D264D + party) {
    std::random_device rd;
    silent_.reset(new mm::SilentOT(&ferret_, &channel_));
    iknp_straight = silent_.get();
    (void)rd;
  }
  IO *io;
  int party;
  mm::SilentOT *iknp_straight = nullptr;

 private:
  shim::NetChannel channel_;
  shim::IdealFerretOverNet ferret_;
  std::unique_ptr<mm::SilentOT> silent_;
};
}  // namespace sci

// aux-protocols.h:33-60: constructed by the programs and handed to Truncation; the conversions do not use it
class AuxProtocols {
 public:
  AuxProtocols(int party, sci::NetIO *io, sci::OTPack<sci::NetIO> *otpack) : party(party), io(io), otpack(otpack) {}
  int party;
  sci::NetIO *io;
  sci::OTPack<sci::NetIO> *otpack;
};
template <typename IO>
class MillionaireWithEquality;

class Truncation {
 public:
  sci::NetIO *io = nullptr;
  sci::OTPack<sci::NetIO> *otpack;
  AuxProtocols *aux = nullptr;
  int party;
  // truncation.h:42-44 / truncation.cpp:28-52
  Truncation(int party, sci::NetIO *io, sci::OTPack<sci::NetIO> *otpack, AuxProtocols *auxp = nullptr,
             MillionaireWithEquality<sci::NetIO> * = nullptr)
      : io(io), otpack(otpack), aux(auxp), party(party), impl_(party, otpack->iknp_straight) {}
  // truncation.cpp:319-381
  void ring_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t bw_A, int32_t bw_B) {
    impl_.ring_to_ring(dim, inA, outB, bw_A, bw_B);
  }
  // :383-442
  void ring_to_ring128(int32_t dim, uint64_t *inA, __uint128_t *outB, int32_t bw_A, int32_t bw_B) {
    assert(bw_B - bw_A <= 64);  // :392
    impl_.ring_to_ring128(dim, inA, outB, bw_A, bw_B);
  }
  // :518-581
  void ring_to_field(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t ring_bw, uint64_t field_mod) {
    impl_.ring_to_field(dim, inA, outB, ring_bw, field_mod);
  }
  // :583-637
  void field_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw) {
    impl_.field_to_ring(dim, inA, outB, field_mod, ring_bw);
  }
  // :639-695
  void field_to_ring_with_truncate(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw,
                                   int32_t shift_bw) {
    impl_.field_to_ring_with_truncate(dim, inA, outB, field_mod, ring_bw, shift_bw);
  }

 private:
  mm::Truncation impl_;
};
