// Layer adapters between the networks' NHWC arrays and CheetahLinear / Truncation
// (SCI/src/library_fixed_uniform_cheetah.cpp:70-600); see include/mm_gemini/library.h.
#include "mm_gemini/library.h"

#include <exception>
#include <thread>

#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <vector>

namespace mm {
using gemini::CheetahLinear;
using gemini::Padding;
using gemini::Tensor;
using gemini::TensorShape;

namespace {
thread_local Library *g_current = nullptr;
const int kServer = 1;
}  // namespace

void Library::Bind(Library *lib) { g_current = lib; }
Library &Library::Current() {
  if (!g_current) throw std::logic_error("mm::Library: no library bound to this thread");
  return *g_current;
}

// :70-135.  One fc call per input row; the weight matrix is transposed into (outputs x inputs).
void Library::MatMul2D(int32_t d0, int32_t d1, int32_t d2, const intType *mat_A, const intType *mat_B, intType *mat_C,
                       bool is_A_weight_matrix) const {
  const int64_t in_rows = is_A_weight_matrix ? d1 : d0, in_cols = is_A_weight_matrix ? d2 : d1;
  const int64_t w_rows = is_A_weight_matrix ? d0 : d1, w_cols = is_A_weight_matrix ? d1 : d2;
  CheetahLinear::FCMeta meta;
  meta.input_shape = TensorShape({in_cols});
  meta.weight_shape = TensorShape({w_cols, w_rows});  // transposed
  meta.is_shared_input = kIsSharedInput;
  const intType *weight_mat = is_A_weight_matrix ? mat_A : mat_B, *input_mat = is_A_weight_matrix ? mat_B : mat_A;
  // weight_mat is w_rows x w_cols (input x output) row-major; the transposition into meta.weight_shape and
  // getRingElt happen on the device (fc_in_out)
  for (int64_t r = 0; r < in_rows; ++r) {
    const intType *input_row = input_mat + r * in_cols;
    Tensor<intType> input_vector;
    if (meta.is_shared_input) {
      input_vector = Tensor<intType>::Wrap(const_cast<intType *>(input_row), meta.input_shape);
    } else {
      input_vector.Reshape(meta.input_shape);
      std::transform(input_row, input_row + in_cols, input_vector.data(), [this](uint64_t v) { return getRingElt(v); });
    }
    Tensor<uint64_t> out_vec;
    cheetah_linear->fc_in_out(input_vector, weight_mat, meta, out_vec);
    std::copy_n(out_vec.data(), out_vec.NumElements(), mat_C + r * in_cols);  // :126-127, as written
  }
}

// :216-300
void Library::Conv2DWrapper(signedIntType N, signedIntType H, signedIntType W, signedIntType CI, signedIntType FH,
                            signedIntType FW, signedIntType CO, signedIntType zPadHLeft, signedIntType zPadHRight,
                            signedIntType zPadWLeft, signedIntType zPadWRight, signedIntType strideH,
                            signedIntType strideW, intType *inputArr, intType *filterArr, intType *outArr) const {
  if (zPadWLeft < zPadWRight) std::swap(zPadWLeft, zPadWRight);
  if (zPadHLeft < zPadHRight) std::swap(zPadHLeft, zPadHRight);
  const signedIntType newH = (H + zPadHLeft + zPadHRight - FH) / strideH + 1;
  const signedIntType newW = (W + zPadWLeft + zPadWRight - FW) / strideW + 1;  // :237 (meta.stride is strideH alone, :262)
  CheetahLinear::ConvMeta meta;
  meta.ishape = TensorShape({CI, H, W});
  meta.fshape = TensorShape({CI, FH, FW});
  meta.n_filters = (size_t)CO;
  meta.padding = (zPadHLeft + zPadHRight + zPadWLeft + zPadWRight) == 0 ? Padding::VALID : Padding::SAME;
  meta.stride = (size_t)strideH;
  meta.is_shared_input = kIsSharedInput;
  // The reference repacks on the host, element by element ([FH][FW][CI][CO] -> per filter [CI][FH][FW] with
  // getRingElt :236-262, [H][W][CI] -> [CI][H][W] :284-287, the result back :293-299); here the arrays cross the link
  // as they are and change layout on the device (CheetahLinear::conv2d_nhwc).  Same values either way.
  for (signedIntType b = 0; b < N; ++b)
    cheetah_linear->conv2d_nhwc(inputArr + b * H * W * CI, filterArr, meta, outArr + b * newH * newW * CO, newH, newW);
}

// :429-490: channel-wise scaling through bn_direct, then the bias share is added
void Library::BatchNorm(int32_t B, int32_t H, int32_t W, int32_t C, const intType *inputArr, const intType *scales,
                        const intType *bias, intType *outArr) const {
  CheetahLinear::BNMeta meta;
  meta.target_base_mod = prime_mod;
  meta.is_shared_input = kIsSharedInput;
  meta.ishape = TensorShape({C, H, W});
  Tensor<intType> scale_vec(TensorShape({C}));
  if (cheetah_linear->party() == kServer)
    std::transform(scales, scales + C, scale_vec.data(), [this](uint64_t v) { return getRingElt(v); });
  Tensor<intType> in_tensor(meta.ishape), out_tensor;
  for (int32_t b = 0; b < B; ++b) {
    for (int32_t h = 0; h < H; ++h)
      for (int32_t w = 0; w < W; ++w)
        for (int32_t c = 0; c < C; ++c) in_tensor(c, h, w) = getRingElt(inputArr[(((int64_t)b * H + h) * W + w) * C + c]);
    cheetah_linear->bn_direct(in_tensor, scale_vec, meta, out_tensor);
    for (int32_t h = 0; h < H; ++h)
      for (int32_t w = 0; w < W; ++w)
        for (int32_t c = 0; c < C; ++c)
          outArr[(((int64_t)b * H + h) * W + w) * C + c] = SecretAdd(out_tensor(c, h, w), bias[c]);
  }
  if (cheetah_linear->party() == kServer) cheetah_linear->safe_erase(scale_vec.data(), scale_vec.NumElements());
}

// :492-565: the range split over num_threads = truncationArr.size() conversion objects, one std::thread each, chunks
// of (size / (8 * num_threads)) * 8 elements with the remainder on the last thread (:494-502); every object runs
// its own batched device calls on its own channel.  With no array set the single object takes the whole range.
template <class Fn>
static void split_over_threads(const Library &lib, int32_t size, intType *in, intType *out, Fn &&one) {
  const int nt = (int)lib.truncationArr.size();
  if (nt == 0) {
    one(*lib.truncation, size, in, out);
    return;
  }
  const int32_t chunk = (size / (8 * nt)) * 8;
  std::vector<std::thread> workers;
  std::vector<std::exception_ptr> failed(nt);
  for (int i = 0; i < nt; ++i) {
    const int32_t offset = i * chunk, cur = i == nt - 1 ? size - offset : chunk;
    if (cur <= 0) continue;  // (both parties skip the same empty chunks)
    workers.emplace_back([&, i, offset, cur] {
      try {
        one(*lib.truncationArr[i], cur, in + offset, out + offset);
      } catch (...) {
        failed[i] = std::current_exception();
      }
    });
  }
  for (auto &w : workers) w.join();
  for (auto &e : failed)
    if (e) std::rethrow_exception(e);
}
void Library::Ring2Field(int32_t size, intType *inArr, intType *outputArr) const {
  split_over_threads(*this, size, inArr, outputArr, [this](Truncation &t, int32_t n, intType *in, intType *out) {
    t.ring_to_field(n, in, out, bitlength, field_mod);
  });
}
void Library::Field2Ring(int32_t size, intType *inArr, intType *outputArr) const {
  split_over_threads(*this, size, inArr, outputArr, [this](Truncation &t, int32_t n, intType *in, intType *out) {
    t.field_to_ring(n, in, out, field_mod, bitlength);
  });
}
void Library::Field2Ring_with_Truncate(int32_t size, intType *inArr, intType *outputArr) const {
  split_over_threads(*this, size, inArr, outputArr, [this](Truncation &t, int32_t n, intType *in, intType *out) {
    t.field_to_ring_with_truncate(n, in, out, field_mod, bitlength, kScale);
  });
}

// :567-600: ring -> field, SIMD product with the model's scale vector, field -> ring with the free truncation
void Library::ElemWiseActModelVectorMult(int32_t size, intType *inArr, intType *multArrVec, intType *outputArr) const {
  CheetahLinear::BNMeta meta;
  meta.target_base_mod = field_mod;
  meta.is_shared_input = kIsSharedInput;
  meta.vec_shape = TensorShape({size});
  Tensor<intType> field_in(meta.vec_shape), field_scale(meta.vec_shape), field_out;
  if (meta.is_shared_input) {
    Ring2Field(size, inArr, field_in.data());
  } else {
    std::transform(inArr, inArr + size, field_in.data(), [this](uint64_t v) { return getFieldElt((int64_t)v); });
  }
  if (cheetah_linear->party() == kServer)
    std::transform(multArrVec, multArrVec + size, field_scale.data(),
                   [this](uint64_t v) { return getFieldElt((int64_t)v); });
  cheetah_linear->bn(field_in, field_scale, meta, field_out);
  Field2Ring_with_Truncate(size, field_out.data(), outputArr);
  if (cheetah_linear->party() == kServer) cheetah_linear->safe_erase(field_scale.data(), field_scale.NumElements());
}

}  // namespace mm

void MatMul2D(int32_t d0, int32_t d1, int32_t d2, const mm::intType *mat_A, const mm::intType *mat_B,
              mm::intType *mat_C, bool is_A_weight_matrix) {
  mm::Library::Current().MatMul2D(d0, d1, d2, mat_A, mat_B, mat_C, is_A_weight_matrix);
}
void Conv2DWrapper(mm::signedIntType N, mm::signedIntType H, mm::signedIntType W, mm::signedIntType CI,
                   mm::signedIntType FH, mm::signedIntType FW, mm::signedIntType CO, mm::signedIntType zPadHLeft,
                   mm::signedIntType zPadHRight, mm::signedIntType zPadWLeft, mm::signedIntType zPadWRight,
                   mm::signedIntType strideH, mm::signedIntType strideW, mm::intType *inputArr,
                   mm::intType *filterArr, mm::intType *outArr) {
  mm::Library::Current().Conv2DWrapper(N, H, W, CI, FH, FW, CO, zPadHLeft, zPadHRight, zPadWLeft, zPadWRight, strideH,
                                       strideW, inputArr, filterArr, outArr);
}
void BatchNorm(int32_t B, int32_t H, int32_t W, int32_t C, const mm::intType *inputArr, const mm::intType *scales,
               const mm::intType *bias, mm::intType *outArr) {
  mm::Library::Current().BatchNorm(B, H, W, C, inputArr, scales, bias, outArr);
}
void Ring2Field(int32_t size, mm::intType *inArr, mm::intType *outputArr) {
  mm::Library::Current().Ring2Field(size, inArr, outputArr);
}
void Field2Ring(int32_t size, mm::intType *inArr, mm::intType *outputArr) {
  mm::Library::Current().Field2Ring(size, inArr, outputArr);
}
void Field2Ring_with_Truncate(int32_t size, mm::intType *inArr, mm::intType *outputArr) {
  mm::Library::Current().Field2Ring_with_Truncate(size, inArr, outputArr);
}
void ElemWiseActModelVectorMult(int32_t size, mm::intType *inArr, mm::intType *multArrVec, mm::intType *outputArr) {
  mm::Library::Current().ElemWiseActModelVectorMult(size, inArr, multArrVec, outputArr);
}
