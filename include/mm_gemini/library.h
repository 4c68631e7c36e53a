// The layer adapters of SCI/src/library_fixed_uniform_cheetah.cpp over the C++ host mirror: MatMul2D (:70),
// Conv2DWrapper (:216), BatchNorm (:429), Ring2Field (:492), Field2Ring (:517), Field2Ring_with_Truncate
// (:542), ElemWiseActModelVectorMult (:567), with the reference's argument lists and array layouts (NHWC
// activations, [FH][FW][CI][CO] filters, row-major matrices), so the linear-layer calls of
// networks/main_*.cpp drop in.  The reference keeps party, bit width, scale and the protocol objects in
// process-wide globals (SCI/src/globals.h); here they live in a Library object and the free functions below
// act on the object bound to the calling thread (mm::Library::Bind), which lets both parties run in one
// process for tests.  Quirks kept on purpose: SAME vs VALID from "any pad non-zero",
// only strideH is used, left/right pads are swapped into descending order, MatMul2D writes row r of the
// result at mat_C + r * input_cols.
#pragma once
#include <cstdint>
#include <vector>

#include "mm_gemini/cheetah_linear.h"
#include "mm_gemini/truncation.h"

namespace mm {

using intType = uint64_t;
using signedIntType = int64_t;

struct Library {
  int party = 0;                          // sci::ALICE = SERVER = 1, sci::BOB = CLIENT = 2
  int32_t bitlength = 37, kScale = 12;    // shipped ell / k (scripts/common.sh:48-56)
  uint64_t prime_mod = 0, field_mod = 0;  // 2^bitlength and the table prime (constants.h:45-50)
  bool kIsSharedInput = true;             // networks set it false around their first layer
  gemini::CheetahLinear *cheetah_linear = nullptr;
  Truncation *truncation = nullptr;       // truncationArr[0] of the reference (library_fixed_uniform.cpp:1556)
  // One conversion object per worker thread, each on its own channel, odd ones built with the roles swapped
  // (party 3 - party: library_fixed_uniform.cpp:1540-1554), so that both directions of the link carry COT
  // corrections.  Ring2Field / Field2Ring / Field2Ring_with_Truncate split their range over them exactly as
  // library_fixed_uniform_cheetah.cpp:492-565 does; empty = the single object above takes the whole range.
  std::vector<Truncation *> truncationArr;

  uint64_t getRingElt(int64_t x) const { return (uint64_t)x & (prime_mod - 1); }
  uint64_t getFieldElt(int64_t x) const {
    return x >= 0 ? (uint64_t)x % field_mod : field_mod - 1 - ((uint64_t)(-x - 1) % field_mod);
  }
  uint64_t SecretAdd(uint64_t x, uint64_t y) const { return (x + y) & (prime_mod - 1); }

  void MatMul2D(int32_t d0, int32_t d1, int32_t d2, const intType *mat_A, const intType *mat_B, intType *mat_C,
                bool is_A_weight_matrix) const;
  void Conv2DWrapper(signedIntType N, signedIntType H, signedIntType W, signedIntType CI, signedIntType FH,
                     signedIntType FW, signedIntType CO, signedIntType zPadHLeft, signedIntType zPadHRight,
                     signedIntType zPadWLeft, signedIntType zPadWRight, signedIntType strideH, signedIntType strideW,
                     intType *inputArr, intType *filterArr, intType *outArr) const;
  void BatchNorm(int32_t B, int32_t H, int32_t W, int32_t C, const intType *inputArr, const intType *scales,
                 const intType *bias, intType *outArr) const;
  void Ring2Field(int32_t size, intType *inArr, intType *outputArr) const;
  void Field2Ring(int32_t size, intType *inArr, intType *outputArr) const;
  void Field2Ring_with_Truncate(int32_t size, intType *inArr, intType *outputArr) const;
  void ElemWiseActModelVectorMult(int32_t size, intType *inArr, intType *multArrVec, intType *outputArr) const;

  // the object the free functions of this thread act on
  static void Bind(Library *lib);
  static Library &Current();
};

}  // namespace mm

// free functions with the reference's names and signatures, acting on mm::Library::Current()
void MatMul2D(int32_t d0, int32_t d1, int32_t d2, const mm::intType *mat_A, const mm::intType *mat_B,
              mm::intType *mat_C, bool is_A_weight_matrix);
void Conv2DWrapper(mm::signedIntType N, mm::signedIntType H, mm::signedIntType W, mm::signedIntType CI,
                   mm::signedIntType FH, mm::signedIntType FW, mm::signedIntType CO, mm::signedIntType zPadHLeft,
                   mm::signedIntType zPadHRight, mm::signedIntType zPadWLeft, mm::signedIntType zPadWRight,
                   mm::signedIntType strideH, mm::signedIntType strideW, mm::intType *inputArr,
                   mm::intType *filterArr, mm::intType *outArr);
void BatchNorm(int32_t B, int32_t H, int32_t W, int32_t C, const mm::intType *inputArr, const mm::intType *scales,
               const mm::intType *bias, mm::intType *outArr);
void Ring2Field(int32_t size, mm::intType *inArr, mm::intType *outputArr);
void Field2Ring(int32_t size, mm::intType *inArr, mm::intType *outputArr);
void Field2Ring_with_Truncate(int32_t size, mm::intType *inArr, mm::intType *outputArr);
void ElemWiseActModelVectorMult(int32_t size, mm::intType *inArr, mm::intType *multArrVec, mm::intType *outputArr);
