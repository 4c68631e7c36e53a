// gemini::HomConv2DSS with the reference's public surface (include/gemini/cheetah/hom_conv2d_ss.h:27-84): same
// method names, argument order and gemini::Code results; mm::he:: containers stand where the seal:: types do
// (SEAL is not part of this build) and every method is one libmm_hip layer call between an upload and a
// download.  The reference's worker pool (nthreads) has no work left to split and is ignored.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <vector>

#include "mm_gemini/core.h"
#include "mm_gemini/he_types.h"

struct mmh_conv_plan;

namespace gemini {

class HomConv2DSS {
 public:
  static constexpr size_t kMaxThreads = 16;

  struct Meta {  // hom_conv2d_ss.h:35-42
    TensorShape ishape;
    TensorShape fshape;
    size_t n_filters;
    Padding padding;
    size_t stride;
    bool is_shared_input;
  };

  explicit HomConv2DSS() = default;
  ~HomConv2DSS();
  HomConv2DSS(const HomConv2DSS &) = delete;

  // :467-495.  sk: the key owner (client); pk: the evaluating party (server)
  Code setUp(const mm::he::Context &context, std::optional<mm::he::SecretKey> sk = std::nullopt,
             std::shared_ptr<mm::he::PublicKey> pk = nullptr);
  size_t poly_degree() const;
  uint64_t plain_modulus() const;

  Code encryptImage(const Tensor<uint64_t> &in_tensor_share, const Meta &meta,
                    std::vector<mm::he::Ciphertext> &encrypted_share, size_t nthreads = 1) const;
  Code encodeImage(const Tensor<uint64_t> &in_tensor_share, const Meta &meta,
                   std::vector<mm::he::Plaintext> &encoded_share, size_t nthreads = 1) const;
  Code encodeFilters(const std::vector<Tensor<uint64_t>> &filters, const Meta &meta,
                     std::vector<std::vector<mm::he::Plaintext>> &encoded_filters, size_t nthreads = 1) const;
  Code conv2DSS(const std::vector<mm::he::Ciphertext> &img_share0, const std::vector<mm::he::Plaintext> &img_share1,
                const std::vector<std::vector<mm::he::Plaintext>> &filters, const Meta &meta,
                std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1, size_t nthreads = 1) const;
  Code decryptToTensor(const std::vector<mm::he::Ciphertext> &enc_tensor, const Meta &meta, Tensor<uint64_t> &out,
                       size_t nthreads = 1) const;
  // cleartext convolution mod the plain modulus (:975-1012; Tensor::Conv2D, tensor.h:274-362)
  Code idealFunctionality(const Tensor<uint64_t> &in_tensor, const std::vector<Tensor<uint64_t>> &filters,
                          const Meta &meta, Tensor<uint64_t> &out_tensor) const;

 private:
  mmh_conv_plan *plan(const Meta &meta) const;
  const mm::he::Context *context_ = nullptr;
  std::optional<mm::he::SecretKey> sk_;
  std::shared_ptr<mm::he::PublicKey> pk_;
  uint64_t *d_sk_ = nullptr, *d_sk_last_ = nullptr, *d_pk_last_ = nullptr;  // device copies of the keys
  uint32_t prng_key_[8] = {0};
  mutable uint64_t stream_id_ = 1;
  mutable std::mutex mu_;
  mutable std::map<std::vector<int64_t>, mmh_conv_plan *> plans_;
};

}  // namespace gemini
