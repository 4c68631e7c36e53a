// gemini::HomBNSS with the reference's public surface (include/gemini/cheetah/hom_bn_ss.h:25-112): the SIMD path
// (setUp_Field, encryptVector, encodeVector, encodeScales, bn, bole, decryptToVector) over a batching prime and the
// coefficient-packed direct path (setUp_Base, encryptTensor, encodeTensor, bn_direct, decryptToTensor) over the
// ring; same method names, argument order and gemini::Code results, mm::he:: containers where the seal:: types
// stand.  sendEncryptVector / recvEncryptVector serialise with SEAL's format and are not mirrored.
#pragma once
#include <memory>
#include <mutex>
#include <optional>
#include <vector>

#include "mm_gemini/core.h"
#include "mm_gemini/he_types.h"

namespace gemini {

class HomBNSS {
 public:
  static constexpr size_t kMaxThreads = 16;
  struct Meta {  // hom_bn_ss.h:33-38
    TensorShape ishape;
    TensorShape vec_shape;
    uint64_t target_base_mod;
    bool is_shared_input;
  };

  HomBNSS();
  ~HomBNSS();
  HomBNSS(const HomBNSS &) = delete;

  Code setUp_Field(uint64_t target_field_mod, const mm::he::Context &simd_context,
                   std::optional<mm::he::SecretKey> simd_sk = std::nullopt,
                   std::shared_ptr<mm::he::PublicKey> simd_pk = nullptr);
  Code setUp_Base(uint64_t target_base_mod, const mm::he::Context &direct_context,
                  std::optional<mm::he::SecretKey> direct_sk = std::nullopt,
                  std::shared_ptr<mm::he::PublicKey> direct_pk = nullptr);
  size_t poly_degree() const { return poly_degree_; }
  uint64_t base_plain_modulus() const;
  uint64_t field_plain_modulus() const;

  Code encryptVector(const Tensor<uint64_t> &in_vec, const Meta &meta, std::vector<mm::he::Ciphertext> &out,
                     size_t nthreads = 1) const;
  Code encodeVector(const Tensor<uint64_t> &in_vec, const Meta &meta, std::vector<mm::he::Plaintext> &out,
                    size_t nthreads = 1) const;
  Code encodeScales(const Tensor<uint64_t> &scales, const Meta &meta, std::vector<mm::he::Plaintext> &out,
                    size_t nthreads = 1) const;
  Code bn(const std::vector<mm::he::Ciphertext> &vec_share0, const std::vector<mm::he::Plaintext> &vec_share1,
          const std::vector<mm::he::Plaintext> &scales, const Meta &meta, std::vector<mm::he::Ciphertext> &out_share0,
          Tensor<uint64_t> &out_share1, size_t nthreads = 1) const;
  Code bole(std::vector<mm::he::Ciphertext> &encrypted_x, std::vector<mm::he::Ciphertext> &encrypted_y,
            const std::vector<mm::he::Plaintext> &encoded_x, const std::vector<mm::he::Plaintext> &encoded_y,
            const Meta &meta, std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1,
            size_t nthreads = 1) const;
  Code decryptToVector(const std::vector<mm::he::Ciphertext> &in_vec, const Meta &meta, Tensor<uint64_t> &out_vec,
                       size_t nthreads = 1) const;

  Code encryptTensor(const Tensor<uint64_t> &in_tensor, const Meta &meta, std::vector<mm::he::Ciphertext> &out,
                     size_t nthreads = 1) const;
  Code encodeTensor(const Tensor<uint64_t> &in_tensor, const Meta &meta, std::vector<mm::he::Plaintext> &out,
                    size_t nthreads = 1) const;
  Code bn_direct(const std::vector<mm::he::Ciphertext> &tensor_share0,
                 const std::vector<mm::he::Plaintext> &tensor_share1, const Tensor<uint64_t> &scales, const Meta &meta,
                 std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1, size_t nthreads = 1) const;
  Code decryptToTensor(const std::vector<mm::he::Ciphertext> &in_vec, const Meta &meta, Tensor<uint64_t> &out_tensor,
                       size_t nthreads = 1) const;

 private:
  struct Keys;
  Code server_simd(const uint64_t *h_cts, size_t n_cts_in, const std::vector<const mm::he::Plaintext *> &pts,
                   bool bole, const Meta &meta, std::vector<mm::he::Ciphertext> &out_share0,
                   Tensor<uint64_t> &out_share1) const;
  uint64_t target_base_mod_ = 0, target_field_mod_ = 0;
  size_t poly_degree_ = 0;
  const mm::he::Context *direct_context_ = nullptr, *simd_context_ = nullptr;
  std::unique_ptr<Keys> direct_keys_, simd_keys_;
  uint32_t prng_key_[8] = {0};
  mutable uint64_t stream_id_ = 1;
  mutable std::mutex mu_;
};

}  // namespace gemini
