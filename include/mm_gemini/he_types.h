// Host-side value types that stand where SEAL's do in the signatures of the reference's HE classes
// (seal::SEALContext, SecretKey, PublicKey, Plaintext, Ciphertext; include/gemini/cheetah/hom_conv2d_ss.h:48-80).
// SEAL is not part of this build: these are plain containers with SEAL's memory layout --
// Ciphertext::data(k)[l * N + i] (hom_conv2d_ss.cc:96-103), keys in NTT form -- whose arithmetic is libmm_hip's.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

struct mmh_ctx;

namespace mm {
namespace he {

// BFV parameters + device tables: N, CoeffModulus::Create(N, bits), plain modulus t (cheetah-api.cpp:95-105)
class Context {
 public:
  Context(size_t poly_degree, const std::vector<int> &coeff_modulus_bits, uint64_t plain_modulus, int device = 0);
  ~Context();
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  const mmh_ctx *raw() const { return ctx_; }
  size_t poly_degree() const { return N_; }
  size_t coeff_modulus_size() const { return L_; }
  uint64_t plain_modulus() const { return t_; }
  const std::vector<uint64_t> &coeff_modulus() const { return q_; }

 private:
  mmh_ctx *ctx_ = nullptr;
  size_t N_ = 0, L_ = 0;
  uint64_t t_ = 0;
  std::vector<uint64_t> q_;
};

struct SecretKey {
  std::vector<uint64_t> data;  // [L][N], NTT form
};
struct PublicKey {
  std::vector<uint64_t> data;  // [2][L][N], NTT form
};
// KeyGenerator + create_public_key (cheetah-api.cpp:107-114), randomness from the device ChaCha20 sampler
void GenerateKeys(const Context &context, SecretKey &sk, PublicKey &pk);

struct Plaintext {
  std::vector<uint64_t> coeffs;
  size_t coeff_count() const { return coeffs.size(); }
  uint64_t *data() { return coeffs.data(); }
  const uint64_t *data() const { return coeffs.data(); }
  bool is_zero() const {
    for (uint64_t c : coeffs)
      if (c) return false;
    return true;
  }
};

struct Ciphertext {
  std::vector<uint64_t> words;  // [size][coeff_modulus_size][poly_modulus_degree]
  size_t n_limbs = 0, degree = 0;
  bool ntt_form = false;
  void resize(size_t size, size_t limbs, size_t N) {
    n_limbs = limbs, degree = N;
    words.assign(size * limbs * N, 0);
  }
  size_t size() const { return n_limbs && degree ? words.size() / (n_limbs * degree) : 0; }
  size_t coeff_modulus_size() const { return n_limbs; }
  size_t poly_modulus_degree() const { return degree; }
  bool is_ntt_form() const { return ntt_form; }
  uint64_t *data(size_t k = 0) { return words.data() + k * n_limbs * degree; }
  const uint64_t *data(size_t k = 0) const { return words.data() + k * n_limbs * degree; }
};

}  // namespace he
}  // namespace mm
