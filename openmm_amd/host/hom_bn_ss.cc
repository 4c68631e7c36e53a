// gemini::HomBNSS (include/gemini/cheetah/hom_bn_ss.cc) over libmm_hip: the reference's argument checks, then
// one layer call of the C ABI between an upload and a download.
#include "mm_gemini/hom_bn_ss.h"

#include "host_util.h"

namespace gemini {

struct HomBNSS::Keys : DeviceKeys {};

HomBNSS::HomBNSS() = default;
HomBNSS::~HomBNSS() {
  if (direct_keys_) direct_keys_->clear();
  if (simd_keys_) simd_keys_->clear();
}

namespace {
Code check_keys(const mm::he::Context &c, const std::optional<mm::he::SecretKey> &sk,
                const std::shared_ptr<mm::he::PublicKey> &pk) {
  const size_t N = c.poly_degree(), L = c.coeff_modulus_size();
  if (sk) MM_ENSURE(sk->data.size() == L * N, Code::ERR_INVALID_ARG);
  if (pk) MM_ENSURE(pk->data.size() == 2 * L * N, Code::ERR_INVALID_ARG);
  return Code::OK;
}
bool is_batching_prime(uint64_t p, size_t N) {  // qualifiers().using_batching: p prime, p = 1 mod 2N
  if (p < 2 || (p - 1) % (2 * N) != 0) return false;
  for (uint64_t d = 2; d * d <= p && d < (1ull << 22); ++d)
    if (p % d == 0) return false;
  return true;
}
size_t n_simd_cts(const HomBNSS::Meta &meta, size_t N) {
  const size_t n = (size_t)meta.vec_shape.length(), sub = std::min(n, N);
  return sub ? (n + sub - 1) / sub : 0;
}
}  // namespace

// hom_bn_ss.cc:107-149
Code HomBNSS::setUp_Field(uint64_t target_field_mod, const mm::he::Context &simd_context,
                          std::optional<mm::he::SecretKey> simd_sk, std::shared_ptr<mm::he::PublicKey> simd_pk) {
  MM_ENSURE(simd_context.raw() != nullptr, Code::ERR_NULL_POINTER);
  target_field_mod_ = target_field_mod;
  MM_ENSURE(target_field_mod == simd_context.plain_modulus(), Code::ERR_INVALID_ARG);
  MM_ENSURE(is_batching_prime(target_field_mod, simd_context.poly_degree()), Code::ERR_INVALID_ARG);
  const Code kc = check_keys(simd_context, simd_sk, simd_pk);
  if (kc != Code::OK) return kc;
  MM_ENSURE(seed_from_os(prng_key_), Code::ERR_INTERNAL);
  simd_context_ = &simd_context;
  poly_degree_ = simd_context.poly_degree();
  if (!simd_keys_) simd_keys_.reset(new Keys());
  MM_TRY(simd_keys_->load(simd_context, simd_sk ? &*simd_sk : nullptr, simd_pk.get()));
  return Code::OK;
}

// :69-105
Code HomBNSS::setUp_Base(uint64_t target_base_mod, const mm::he::Context &direct_context,
                         std::optional<mm::he::SecretKey> direct_sk, std::shared_ptr<mm::he::PublicKey> direct_pk) {
  MM_ENSURE(direct_context.raw() != nullptr, Code::ERR_NULL_POINTER);
  target_base_mod_ = target_base_mod;
  MM_ENSURE(target_base_mod == direct_context.plain_modulus(), Code::ERR_INVALID_ARG);
  const Code kc = check_keys(direct_context, direct_sk, direct_pk);
  if (kc != Code::OK) return kc;
  MM_ENSURE(seed_from_os(prng_key_), Code::ERR_INTERNAL);
  direct_context_ = &direct_context;
  poly_degree_ = direct_context.poly_degree();
  if (!direct_keys_) direct_keys_.reset(new Keys());
  MM_TRY(direct_keys_->load(direct_context, direct_sk ? &*direct_sk : nullptr, direct_pk.get()));
  return Code::OK;
}

uint64_t HomBNSS::base_plain_modulus() const { return direct_context_ ? direct_context_->plain_modulus() : -1; }
uint64_t HomBNSS::field_plain_modulus() const { return simd_context_ ? simd_context_->plain_modulus() : -1; }

// :151-201
Code HomBNSS::encryptVector(const Tensor<uint64_t> &in_vec, const Meta &meta, std::vector<mm::he::Ciphertext> &out,
                            size_t) const {
  MM_ENSURE(simd_context_ && simd_keys_ && simd_keys_->sk, Code::ERR_CONFIG);
  MM_ENSURE(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  MM_ENSURE(in_vec.shape().IsSameSize(meta.vec_shape), Code::ERR_DIM_MISMATCH);
  const size_t N = poly_degree_, L = simd_context_->coeff_modulus_size(), ct = 2 * L * N;
  const size_t n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  Dev vec(n), a(n_ct * L * N), e(n_ct * N), cts(n_ct * ct);
  MM_TRY(first_rc({vec.rc, a.rc, e.rc, cts.rc}));
  MM_TRY(vec.up(in_vec.data(), n));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 2;
  }
  MM_TRY(mmh_sample_uniform(simd_context_->raw(), 0, prng_key_, sid, a.p, n_ct, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e.p, n_ct * N, nullptr));
  MM_TRY(mmh_bn_encrypt_vector(simd_context_->raw(), simd_keys_->sk, vec.p, n, a.p, (const int64_t *)e.p, cts.p, nullptr));
  out.resize(n_ct);
  for (size_t i = 0; i < n_ct; ++i) {
    out[i].resize(2, L, N);
    MM_TRY(cts.down(out[i].data(), ct, i * ct));
  }
  return Code::OK;
}

// :203-233
Code HomBNSS::encodeVector(const Tensor<uint64_t> &in_vec, const Meta &meta, std::vector<mm::he::Plaintext> &out,
                           size_t) const {
  MM_ENSURE(simd_context_, Code::ERR_CONFIG);
  MM_ENSURE(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  MM_ENSURE(in_vec.shape().IsSameSize(meta.vec_shape), Code::ERR_DIM_MISMATCH);
  const size_t N = poly_degree_, n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  Dev vec(n), pts(n_ct * N);
  MM_TRY(first_rc({vec.rc, pts.rc}));
  MM_TRY(vec.up(in_vec.data(), n));
  MM_TRY(mmh_bn_encode_vector(simd_context_->raw(), vec.p, n, pts.p, nullptr));
  out.resize(n_ct);
  for (size_t i = 0; i < n_ct; ++i) {
    out[i].coeffs.resize(N);
    MM_TRY(pts.down(out[i].data(), N, i * N));
  }
  return Code::OK;
}

// :235-239
Code HomBNSS::encodeScales(const Tensor<uint64_t> &scales, const Meta &meta, std::vector<mm::he::Plaintext> &out,
                           size_t nthreads) const {
  return encodeVector(scales, meta, out, nthreads);
}

// bn and bole share the upload / mask / download frame
Code HomBNSS::server_simd(const uint64_t *h_cts, size_t n_cts_in, const std::vector<const mm::he::Plaintext *> &pts,
                          bool bole, const Meta &meta, std::vector<mm::he::Ciphertext> &out_share0,
                          Tensor<uint64_t> &out_share1) const {
  const size_t N = poly_degree_, L = simd_context_->coeff_modulus_size(), ct = 2 * L * N;
  const size_t n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  Dev cts(n_cts_in * ct), d_pts(std::max<size_t>(1, pts.size()) * N), rnd(n), out_cts(n_ct * 2 * N), share(n);
  MM_TRY(first_rc({cts.rc, d_pts.rc, rnd.rc, out_cts.rc, share.rc}));
  MM_TRY(cts.up(h_cts, n_cts_in * ct));
  for (size_t i = 0; i < pts.size(); ++i) {
    MM_ENSURE(pts[i]->coeff_count() == N, Code::ERR_INTERNAL);
    MM_TRY(d_pts.up(pts[i]->data(), N, i * N));
  }
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_++;
  }
  MM_TRY(mmh_chacha20_words(prng_key_, sid, 0, rnd.p, n, nullptr));
  if (bole) {
    MM_TRY(mmh_bole_server(simd_context_->raw(), cts.p, d_pts.p, n, rnd.p, out_cts.p, share.p, nullptr));
  } else {
    const bool shared = pts.size() == 2 * n_ct;  // [share ; scales] or [scales]
    MM_TRY(mmh_bn_server(simd_context_->raw(), cts.p, shared ? d_pts.p : nullptr, d_pts.p + (shared ? n_ct * N : 0), n,
                         rnd.p, out_cts.p, share.p, nullptr));
  }
  out_share0.resize(n_ct);
  for (size_t k = 0; k < n_ct; ++k) {
    out_share0[k].resize(2, 1, N);
    MM_TRY(out_cts.down(out_share0[k].data(), 2 * N, k * 2 * N));
  }
  if (!out_share1.shape().IsSameSize(meta.vec_shape)) out_share1.Reshape(meta.vec_shape);
  MM_TRY(share.down(out_share1.data(), n));
  return Code::OK;
}

namespace {
Code gather_cts(const std::vector<mm::he::Ciphertext> &v, size_t L, size_t N, std::vector<uint64_t> &host, size_t at) {
  for (size_t i = 0; i < v.size(); ++i) {
    MM_ENSURE(v[i].size() == 2 && v[i].coeff_modulus_size() == L && v[i].poly_modulus_degree() == N, Code::ERR_INTERNAL);
    std::copy(v[i].data(), v[i].data() + 2 * L * N, host.begin() + (at + i) * 2 * L * N);
  }
  return Code::OK;
}
}  // namespace

// :241-278 + addMask :318-362
Code HomBNSS::bn(const std::vector<mm::he::Ciphertext> &vec_share0, const std::vector<mm::he::Plaintext> &vec_share1,
                 const std::vector<mm::he::Plaintext> &scales, const Meta &meta,
                 std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1, size_t) const {
  MM_ENSURE(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  MM_ENSURE(simd_context_, Code::ERR_CONFIG);
  const size_t N = poly_degree_, L = simd_context_->coeff_modulus_size(), n_ct = n_simd_cts(meta, N);
  MM_ENSURE(vec_share0.size() == n_ct, Code::ERR_INVALID_ARG);
  MM_ENSURE(scales.size() == n_ct, Code::ERR_INVALID_ARG);
  if (meta.is_shared_input) MM_ENSURE(vec_share1.size() == n_ct, Code::ERR_INVALID_ARG);
  std::vector<uint64_t> host(n_ct * 2 * L * N);
  const Code gc = gather_cts(vec_share0, L, N, host, 0);
  if (gc != Code::OK) return gc;
  std::vector<const mm::he::Plaintext *> pts;
  if (meta.is_shared_input)
    for (const auto &p : vec_share1) pts.push_back(&p);
  for (const auto &p : scales) pts.push_back(&p);
  return server_simd(host.data(), n_ct, pts, false, meta, out_share0, out_share1);
}

// :280-316 + addMask
Code HomBNSS::bole(std::vector<mm::he::Ciphertext> &encrypted_x, std::vector<mm::he::Ciphertext> &encrypted_y,
                   const std::vector<mm::he::Plaintext> &encoded_x, const std::vector<mm::he::Plaintext> &encoded_y,
                   const Meta &meta, std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1,
                   size_t) const {
  MM_ENSURE(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  MM_ENSURE(simd_context_, Code::ERR_CONFIG);
  const size_t N = poly_degree_, L = simd_context_->coeff_modulus_size(), n_ct = n_simd_cts(meta, N);
  MM_ENSURE(encrypted_x.size() == n_ct, Code::ERR_INVALID_ARG);
  MM_ENSURE(encrypted_y.size() == n_ct, Code::ERR_INVALID_ARG);
  MM_ENSURE(encoded_x.size() == n_ct, Code::ERR_INVALID_ARG);
  MM_ENSURE(encoded_y.size() == n_ct, Code::ERR_INVALID_ARG);
  std::vector<uint64_t> host(2 * n_ct * 2 * L * N);  // [ex ; ey]
  Code gc = gather_cts(encrypted_x, L, N, host, 0);
  if (gc == Code::OK) gc = gather_cts(encrypted_y, L, N, host, n_ct);
  if (gc != Code::OK) return gc;
  std::vector<const mm::he::Plaintext *> pts;  // [py ; px]: out = ex * py + ey * px
  for (const auto &p : encoded_y) pts.push_back(&p);
  for (const auto &p : encoded_x) pts.push_back(&p);
  return server_simd(host.data(), 2 * n_ct, pts, true, meta, out_share0, out_share1);
}

// :364-406
Code HomBNSS::decryptToVector(const std::vector<mm::he::Ciphertext> &in_vec, const Meta &meta,
                              Tensor<uint64_t> &out_vec, size_t) const {
  MM_ENSURE(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  MM_ENSURE(simd_context_ && simd_keys_ && simd_keys_->sk_last, Code::ERR_CONFIG);
  const size_t N = poly_degree_, n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  MM_ENSURE(in_vec.size() == n_ct, Code::ERR_INVALID_ARG);
  Dev cts(n_ct * 2 * N), res(n);
  MM_TRY(first_rc({cts.rc, res.rc}));
  for (size_t k = 0; k < n_ct; ++k) {
    MM_ENSURE(in_vec[k].size() == 2 && in_vec[k].coeff_modulus_size() == 1 && in_vec[k].poly_modulus_degree() == N,
              Code::ERR_INTERNAL);
    MM_TRY(cts.up(in_vec[k].data(), 2 * N, k * 2 * N));
  }
  MM_TRY(mmh_bn_decrypt_vector(simd_context_->raw(), simd_keys_->sk_last, cts.p, n, res.p, nullptr));
  if (!out_vec.shape().IsSameSize(meta.vec_shape)) out_vec.Reshape(meta.vec_shape);
  MM_TRY(res.down(out_vec.data(), n));
  return Code::OK;
}

namespace {
mmh_bn_tensor_meta tensor_meta(const HomBNSS::Meta &meta) {
  return mmh_bn_tensor_meta{meta.ishape.channels(), meta.ishape.height(), meta.ishape.width(), meta.is_shared_input};
}
}  // namespace

// :471-520
Code HomBNSS::encodeTensor(const Tensor<uint64_t> &in_tensor, const Meta &meta, std::vector<mm::he::Plaintext> &out,
                           size_t) const {
  MM_ENSURE(direct_context_, Code::ERR_CONFIG);
  MM_ENSURE(in_tensor.shape().IsSameSize(meta.ishape), Code::ERR_CONFIG);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  MM_TRY(mmh_bn_tensor_count(direct_context_->raw(), &m, &n_ct, nullptr));
  const size_t N = poly_degree_, words = (size_t)in_tensor.NumElements();
  Dev x(words), pts(n_ct * N);
  MM_TRY(first_rc({x.rc, pts.rc}));
  MM_TRY(x.up(in_tensor.data(), words));
  MM_TRY(mmh_bn_encode_tensor(direct_context_->raw(), &m, x.p, pts.p, nullptr));
  out.resize(n_ct);
  for (size_t i = 0; i < n_ct; ++i) {
    out[i].coeffs.resize(N);
    MM_TRY(pts.down(out[i].data(), N, i * N));
  }
  return Code::OK;
}

// :522-576
Code HomBNSS::encryptTensor(const Tensor<uint64_t> &in_tensor, const Meta &meta, std::vector<mm::he::Ciphertext> &out,
                            size_t) const {
  MM_ENSURE(direct_context_ && direct_keys_ && direct_keys_->sk, Code::ERR_CONFIG);
  MM_ENSURE(in_tensor.shape().IsSameSize(meta.ishape), Code::ERR_CONFIG);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  MM_TRY(mmh_bn_tensor_count(direct_context_->raw(), &m, &n_ct, nullptr));
  const size_t N = poly_degree_, L = direct_context_->coeff_modulus_size(), ct = 2 * L * N;
  const size_t words = (size_t)in_tensor.NumElements();
  Dev x(words), a(n_ct * L * N), e(n_ct * N), cts(n_ct * ct);
  MM_TRY(first_rc({x.rc, a.rc, e.rc, cts.rc}));
  MM_TRY(x.up(in_tensor.data(), words));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 2;
  }
  MM_TRY(mmh_sample_uniform(direct_context_->raw(), 0, prng_key_, sid, a.p, n_ct, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e.p, n_ct * N, nullptr));
  MM_TRY(mmh_bn_encrypt_tensor(direct_context_->raw(), &m, direct_keys_->sk, x.p, a.p, (const int64_t *)e.p, cts.p,
                               nullptr));
  out.resize(n_ct);
  for (size_t i = 0; i < n_ct; ++i) {
    out[i].resize(2, L, N);
    MM_TRY(cts.down(out[i].data(), ct, i * ct));
  }
  return Code::OK;
}

// :578-738
Code HomBNSS::bn_direct(const std::vector<mm::he::Ciphertext> &tensor_share0,
                        const std::vector<mm::he::Plaintext> &tensor_share1, const Tensor<uint64_t> &scales,
                        const Meta &meta, std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1,
                        size_t) const {
  MM_ENSURE(direct_context_ && direct_keys_ && direct_keys_->pk, Code::ERR_CONFIG);
  MM_ENSURE(scales.dims() == 1 && scales.length() == meta.ishape.channels(), Code::ERR_CONFIG);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  MM_TRY(mmh_bn_tensor_count(direct_context_->raw(), &m, &n_ct, nullptr));
  MM_ENSURE(tensor_share0.size() == n_ct, Code::ERR_DIM_MISMATCH);
  if (meta.is_shared_input) MM_ENSURE(tensor_share1.size() == n_ct, Code::ERR_DIM_MISMATCH);
  const size_t N = poly_degree_, L = direct_context_->coeff_modulus_size(), ct = 2 * L * N;
  const size_t words = (size_t)meta.ishape.num_elements(), C = (size_t)scales.length();
  Dev cts(n_ct * ct), pts(meta.is_shared_input ? n_ct * N : 1), d_scales(C);
  Dev u(n_ct * N), e0(n_ct * N), e1(n_ct * N), share1(words), out_cts(n_ct * 2 * N);
  MM_TRY(first_rc({cts.rc, pts.rc, d_scales.rc, u.rc, e0.rc, e1.rc, share1.rc, out_cts.rc}));
  for (size_t i = 0; i < n_ct; ++i) {
    MM_ENSURE(tensor_share0[i].size() == 2 && tensor_share0[i].coeff_modulus_size() == L &&
                  tensor_share0[i].poly_modulus_degree() == N,
              Code::ERR_INTERNAL);
    MM_TRY(cts.up(tensor_share0[i].data(), ct, i * ct));
    if (meta.is_shared_input) {
      MM_ENSURE(tensor_share1[i].coeff_count() == N, Code::ERR_INTERNAL);
      MM_TRY(pts.up(tensor_share1[i].data(), N, i * N));
    }
  }
  MM_TRY(d_scales.up(scales.data(), C));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 4;
  }
  MM_TRY(mmh_sample_ternary(prng_key_, sid, (int64_t *)u.p, n_ct * N, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e0.p, n_ct * N, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 2, (int64_t *)e1.p, n_ct * N, nullptr));
  MM_TRY(mmh_chacha20_words(prng_key_, sid + 3, 0, share1.p, words, nullptr));
  MM_TRY(mmh_bn_direct_server(direct_context_->raw(), &m, cts.p, meta.is_shared_input ? pts.p : nullptr, d_scales.p,
                              direct_keys_->pk, (const int64_t *)u.p, (const int64_t *)e0.p, (const int64_t *)e1.p,
                              share1.p, out_cts.p, nullptr));
  out_share0.resize(n_ct);
  for (size_t k = 0; k < n_ct; ++k) {
    out_share0[k].resize(2, 1, N);
    MM_TRY(out_cts.down(out_share0[k].data(), 2 * N, k * 2 * N));
  }
  if (!out_share1.shape().IsSameSize(meta.ishape)) out_share1.Reshape(meta.ishape);
  MM_TRY(share1.down(out_share1.data(), words));  // :712-723: the random tensor, reduced mod the ring
  const uint64_t t = direct_context_->plain_modulus();
  for (size_t i = 0; i < words; ++i) out_share1.data()[i] = (t & (t - 1)) ? out_share1.data()[i] % t : out_share1.data()[i] & (t - 1);
  return Code::OK;
}

// :740-790
Code HomBNSS::decryptToTensor(const std::vector<mm::he::Ciphertext> &in_vec, const Meta &meta,
                              Tensor<uint64_t> &out_tensor, size_t) const {
  MM_ENSURE(direct_context_ && direct_keys_ && direct_keys_->sk_last, Code::ERR_CONFIG);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  MM_TRY(mmh_bn_tensor_count(direct_context_->raw(), &m, &n_ct, nullptr));
  MM_ENSURE(in_vec.size() == n_ct, Code::ERR_DIM_MISMATCH);
  const size_t N = poly_degree_, words = (size_t)meta.ishape.num_elements();
  Dev cts(n_ct * 2 * N), res(words);
  MM_TRY(first_rc({cts.rc, res.rc}));
  for (size_t k = 0; k < n_ct; ++k) {
    MM_ENSURE(in_vec[k].size() == 2 && in_vec[k].coeff_modulus_size() == 1 && in_vec[k].poly_modulus_degree() == N,
              Code::ERR_INTERNAL);
    MM_TRY(cts.up(in_vec[k].data(), 2 * N, k * 2 * N));
  }
  MM_TRY(mmh_bn_decrypt_tensor(direct_context_->raw(), &m, direct_keys_->sk_last, cts.p, res.p, nullptr));
  if (!out_tensor.shape().IsSameSize(meta.ishape)) out_tensor.Reshape(meta.ishape);
  MM_TRY(res.down(out_tensor.data(), words));
  return Code::OK;
}

}  // namespace gemini
