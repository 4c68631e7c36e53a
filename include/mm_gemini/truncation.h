// The modulus-conversion methods of SCI's Truncation class (SCI/src/BuildingBlocks/truncation.h:29-210)
// over libmm_hip: same method names and argument order; host arrays in and out like the reference.
#pragma once
#include <cstdint>

#include "mm_gemini/io.h"

namespace mm {
const int ALICE = 1, BOB = 2;  // SCI/src/utils/constants.h:40-41

class Truncation {
 public:
  Truncation(int party, COT *cot) : party(party), cot_(cot) {}
  int party;
  // truncation.cpp:319, :383, :518, :583, :639.  128-bit output as (lo, hi) word pairs.
  void ring_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t bw_A, int32_t bw_B);
  void ring_to_ring128(int32_t dim, uint64_t *inA, unsigned __int128 *outB, int32_t bw_A, int32_t bw_B);
  void ring_to_field(int32_t dim, uint64_t *inA, uint64_t *outB, int32_t ring_bw, uint64_t field_mod);
  void field_to_ring(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw);
  void field_to_ring_with_truncate(int32_t dim, uint64_t *inA, uint64_t *outB, uint64_t field_mod, int32_t ring_bw,
                                   int32_t shift_bw);

 private:
  COT *cot_;
};
}  // namespace mm
