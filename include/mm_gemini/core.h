// Host-side C++ mirror of the reference's tensor and error types for the HE-linear path:
// gemini::Code (include/gemini/core/types.h:30-40), gemini::Padding / TensorShape
// (include/gemini/cheetah/tensor_shape.h), gemini::Tensor<T> (include/gemini/cheetah/tensor.h),
// without the SEAL and Eigen dependencies.  Only what the callers of CheetahLinear touch
// (SCI/src/library_fixed_uniform_cheetah.cpp, tests/*.cpp) is provided.
#pragma once
#include <array>
#include <cstdint>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <vector>

namespace gemini {

enum class Code {
  OK, ERR_CONFIG, ERR_NULL_POINTER, ERR_DIM_MISMATCH, ERR_SEAL_MEMORY, ERR_KEY_MISSING, ERR_OUT_BOUND,
  ERR_INVALID_ARG, ERR_INTERNAL,
};
std::string CodeMessage(Code code);

enum class Padding { VALID, SAME };

class TensorShape {
 public:
  TensorShape() = default;
  TensorShape(std::initializer_list<int64_t> dims) {
    if (dims.size() > 3) throw std::invalid_argument("TensorShape: at most 3 dimensions");
    for (int64_t d : dims) dims_[n_++] = d;
    for (int i = 0; i < n_; ++i)
      if (dims_[i] < 0) throw std::invalid_argument("TensorShape: Update invalid arguments");
  }
  int dims() const { return n_; }
  int64_t dim_size(int d) const { return d < 0 || d >= n_ ? -1 : dims_[d]; }
  void Update(int d, int64_t v) {
    if (d < 0 || d >= n_ || v < 0) throw std::invalid_argument("TensorShape: Update invalid arguments");
    dims_[d] = v;
  }
  int64_t num_elements() const {
    if (n_ == 0) return 0;
    int64_t p = 1;
    for (int i = 0; i < n_; ++i) p *= dims_[i];
    return p;
  }
  bool IsValid() const { return num_elements() > 0; }
  bool IsSameSize(const TensorShape &b) const {
    if (!IsValid() || !b.IsValid() || n_ != b.n_) return false;
    for (int i = 0; i < n_; ++i)
      if (dims_[i] != b.dims_[i]) return false;
    return true;
  }
  int64_t channels() const { return dim_size(0); }
  int64_t height() const { return dim_size(1); }
  int64_t width() const { return dim_size(2); }
  int64_t rows() const { return dim_size(0); }
  int64_t cols() const { return dim_size(1); }
  int64_t length() const { return dim_size(0); }

 private:
  std::array<int64_t, 3> dims_{{0, 0, 0}};
  int n_ = 0;
};

// Row-major tensor that owns its storage or wraps caller memory (tensor.h:55-83: Reshape on a
// wrapped tensor is an error, Reshape zero-fills).
template <typename T>
class Tensor {
 public:
  using ScalarType = T;
  Tensor() = default;
  explicit Tensor(const TensorShape &shape) { Reshape(shape); }
  static Tensor<T> Wrap(T *ptr, const TensorShape &shape) {
    Tensor<T> t;
    t.wrapped_ = ptr;
    t.shape_ = shape;
    return t;
  }
  void Reshape(const TensorShape &shape) {
    if (wrapped_) throw std::logic_error("Tensor: Reshape on a wrapped tensor");
    shape_ = shape;
    own_.assign(static_cast<size_t>(shape.num_elements()), T(0));
  }
  const TensorShape &shape() const { return shape_; }
  int dims() const { return shape_.dims(); }
  int64_t dim_size(int d) const { return shape_.dim_size(d); }
  int64_t NumElements() const { return shape_.num_elements(); }
  int64_t channels() const { return shape_.channels(); }
  int64_t height() const { return shape_.height(); }
  int64_t width() const { return shape_.width(); }
  int64_t rows() const { return shape_.rows(); }
  int64_t cols() const { return shape_.cols(); }
  int64_t length() const { return shape_.length(); }
  T *data() { return wrapped_ ? wrapped_ : own_.data(); }
  const T *data() const { return wrapped_ ? wrapped_ : own_.data(); }
  T &operator()(int64_t i) { return data()[check(i, 0, 0, 1)]; }
  T operator()(int64_t i) const { return data()[check(i, 0, 0, 1)]; }
  T &operator()(int64_t r, int64_t c) { return data()[check(r, c, 0, 2)]; }
  T operator()(int64_t r, int64_t c) const { return data()[check(r, c, 0, 2)]; }
  T &operator()(int64_t c, int64_t h, int64_t w) { return data()[check(c, h, w, 3)]; }
  T operator()(int64_t c, int64_t h, int64_t w) const { return data()[check(c, h, w, 3)]; }

 private:
  size_t check(int64_t a, int64_t b, int64_t c, int nd) const {
    if (nd != shape_.dims()) throw std::invalid_argument("Tensor: wrong number of indices");
    const int64_t d0 = shape_.dim_size(0), d1 = nd > 1 ? shape_.dim_size(1) : 1, d2 = nd > 2 ? shape_.dim_size(2) : 1;
    if (a < 0 || a >= d0 || b < 0 || b >= d1 || c < 0 || c >= d2) throw std::invalid_argument("Tensor index out-of-bound");
    return static_cast<size_t>((a * d1 + b) * d2 + c);
  }
  TensorShape shape_;
  std::vector<T> own_;
  T *wrapped_ = nullptr;
};

}  // namespace gemini
