// gemini::HomFCSS with the reference's public surface (include/gemini/cheetah/hom_fc_ss.h:27-83): out = W * v for a
// rows x cols weight matrix, same method names, argument order and gemini::Code results; mm::he:: containers stand
// where the seal:: types do.  Every method is one libmm_hip layer call between an upload and a download.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <vector>

#include "mm_gemini/core.h"
#include "mm_gemini/he_types.h"

struct mmh_fc_plan;

namespace gemini {

class HomFCSS {
 public:
  static constexpr size_t kMaxThreads = 16;
  struct Meta {  // hom_fc_ss.h:34-38
    TensorShape input_shape;
    TensorShape weight_shape;
    bool is_shared_input;
  };

  HomFCSS();
  ~HomFCSS();
  HomFCSS(const HomFCSS &) = delete;

  Code setUp(const mm::he::Context &context, std::optional<mm::he::SecretKey> sk = std::nullopt,
             std::shared_ptr<mm::he::PublicKey> pk = nullptr);
  size_t poly_degree() const;
  uint64_t plain_modulus() const;

  Code encryptInputVector(const Tensor<uint64_t> &vector, const Meta &meta,
                          std::vector<mm::he::Ciphertext> &encrypted_share, size_t nthreads = 1) const;
  Code encodeInputVector(const Tensor<uint64_t> &vector, const Meta &meta, std::vector<mm::he::Plaintext> &encoded_share,
                         size_t nthreads = 1) const;
  Code encodeWeightMatrix(const Tensor<uint64_t> &weight_matrix, const Meta &meta,
                          std::vector<std::vector<mm::he::Plaintext>> &encoded_share, size_t nthreads = 1) const;
  Code matVecMul(const std::vector<std::vector<mm::he::Plaintext>> &matrix,
                 const std::vector<mm::he::Ciphertext> &vec_share0, const std::vector<mm::he::Plaintext> &vec_share1,
                 const Meta &meta, std::vector<mm::he::Ciphertext> &out_vec_share0, Tensor<uint64_t> &out_vec_share1,
                 size_t nthreads = 1) const;
  Code decryptToVector(const std::vector<mm::he::Ciphertext> &enc_vector, const Meta &meta, Tensor<uint64_t> &out,
                       size_t nthreads = 1) const;
  // The reference's body is empty (hom_fc_ss.cc:604-610: returns OK, `out` untouched); here `out` receives
  // W * v mod the plain modulus, which a caller of the reference cannot have relied on either way.
  Code idealFunctionality(const Tensor<uint64_t> &weight_matrix, const Tensor<uint64_t> &vector, const Meta &meta,
                          Tensor<uint64_t> &out) const;

 private:
  mmh_fc_plan *plan(const Meta &meta) const;
  const mm::he::Context *context_ = nullptr;
  struct Keys;
  std::unique_ptr<Keys> keys_;
  uint32_t prng_key_[8] = {0};
  mutable uint64_t stream_id_ = 1;
  mutable std::mutex mu_;
  mutable std::map<std::vector<int64_t>, mmh_fc_plan *> plans_;
};

}  // namespace gemini
