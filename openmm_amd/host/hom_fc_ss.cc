// gemini::HomFCSS (include/gemini/cheetah/hom_fc_ss.cc) over libmm_hip: the reference's argument checks, then
// one layer call of the C ABI between an upload and a download.
#include "mm_gemini/hom_fc_ss.h"

#include "host_util.h"

namespace gemini {

struct HomFCSS::Keys : DeviceKeys {};

HomFCSS::HomFCSS() = default;
HomFCSS::~HomFCSS() {
  for (auto &kv : plans_) mmh_fc_plan_destroy(kv.second);
  if (keys_) keys_->clear();
}

// hom_fc_ss.cc:129-155
Code HomFCSS::setUp(const mm::he::Context &context, std::optional<mm::he::SecretKey> sk,
                    std::shared_ptr<mm::he::PublicKey> pk) {
  MM_ENSURE(context.raw() != nullptr, Code::ERR_NULL_POINTER);
  const size_t N = context.poly_degree(), L = context.coeff_modulus_size();
  if (sk) MM_ENSURE(sk->data.size() == L * N, Code::ERR_INVALID_ARG);
  if (pk) MM_ENSURE(pk->data.size() == 2 * L * N, Code::ERR_INVALID_ARG);
  MM_ENSURE(seed_from_os(prng_key_), Code::ERR_INTERNAL);
  context_ = &context;
  if (!keys_) keys_.reset(new Keys());
  MM_TRY(keys_->load(context, sk ? &*sk : nullptr, pk.get()));
  return Code::OK;
}

size_t HomFCSS::poly_degree() const { return context_ ? context_->poly_degree() : 0; }
uint64_t HomFCSS::plain_modulus() const { return context_ ? context_->plain_modulus() : -1; }

mmh_fc_plan *HomFCSS::plan(const Meta &meta) const {
  const mmh_fc_meta m{meta.weight_shape.rows(), meta.weight_shape.cols(), meta.is_shared_input};
  const std::vector<int64_t> key{m.rows, m.cols, m.is_shared_input};
  std::lock_guard<std::mutex> g(mu_);
  auto it = plans_.find(key);
  if (it != plans_.end()) return it->second;
  mmh_fc_plan *p = nullptr;
  if (mmh_fc_plan_create(context_->raw(), &m, &p) != MMH_OK) return nullptr;
  return plans_[key] = p;
}

// :214-273
Code HomFCSS::encryptInputVector(const Tensor<uint64_t> &vec, const Meta &meta,
                                 std::vector<mm::he::Ciphertext> &encrypted_share, size_t) const {
  MM_ENSURE(context_ && keys_ && keys_->sk, Code::ERR_CONFIG);
  MM_ENSURE(vec.shape().IsSameSize(meta.input_shape), Code::ERR_DIM_MISMATCH);
  MM_ENSURE(meta.input_shape.num_elements() > 0, Code::ERR_INVALID_ARG);
  mmh_fc_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INTERNAL);
  size_t n_in = 0, n_out = 0;
  MM_TRY(mmh_fc_plan_info(p, nullptr, &n_in, &n_out));
  const size_t N = poly_degree(), L = context_->coeff_modulus_size(), ct = 2 * L * N, len = (size_t)vec.NumElements();
  Dev d_vec(len), a(n_in * L * N), e(n_in * N), cts(n_in * ct);
  MM_TRY(first_rc({d_vec.rc, a.rc, e.rc, cts.rc}));
  MM_TRY(d_vec.up(vec.data(), len));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 2;
  }
  MM_TRY(mmh_sample_uniform(context_->raw(), 0, prng_key_, sid, a.p, n_in, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e.p, n_in * N, nullptr));
  MM_TRY(mmh_fc_encrypt_input(p, keys_->sk, d_vec.p, a.p, (const int64_t *)e.p, cts.p, nullptr));
  encrypted_share.resize(n_in);
  for (size_t i = 0; i < n_in; ++i) {
    encrypted_share[i].resize(2, L, N);
    MM_TRY(cts.down(encrypted_share[i].data(), ct, i * ct));
  }
  return Code::OK;
}

// :275-325
Code HomFCSS::encodeInputVector(const Tensor<uint64_t> &vec, const Meta &meta,
                                std::vector<mm::he::Plaintext> &encoded_share, size_t) const {
  MM_ENSURE(context_, Code::ERR_CONFIG);
  MM_ENSURE(vec.shape().IsSameSize(meta.input_shape), Code::ERR_DIM_MISMATCH);
  MM_ENSURE(meta.input_shape.num_elements() > 0, Code::ERR_INVALID_ARG);
  mmh_fc_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INTERNAL);
  size_t n_in = 0;
  MM_TRY(mmh_fc_plan_info(p, nullptr, &n_in, nullptr));
  const size_t N = poly_degree(), len = (size_t)vec.NumElements();
  Dev d_vec(len), pts(n_in * N);
  MM_TRY(first_rc({d_vec.rc, pts.rc}));
  MM_TRY(d_vec.up(vec.data(), len));
  MM_TRY(mmh_fc_encode_input(p, d_vec.p, pts.p, nullptr));
  encoded_share.resize(n_in);
  for (size_t i = 0; i < n_in; ++i) {
    encoded_share[i].coeffs.resize(N);
    MM_TRY(pts.down(encoded_share[i].data(), N, i * N));
  }
  return Code::OK;
}

// :327-392
Code HomFCSS::encodeWeightMatrix(const Tensor<uint64_t> &w, const Meta &meta,
                                 std::vector<std::vector<mm::he::Plaintext>> &encoded_share, size_t) const {
  MM_ENSURE(context_, Code::ERR_CONFIG);
  MM_ENSURE(w.shape().IsSameSize(meta.weight_shape), Code::ERR_DIM_MISMATCH);
  mmh_fc_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INTERNAL);
  size_t n_in = 0, n_out = 0;
  MM_TRY(mmh_fc_plan_info(p, nullptr, &n_in, &n_out));
  const size_t N = poly_degree(), len = (size_t)w.NumElements();
  Dev d_w(len), pts(n_out * n_in * N);
  MM_TRY(first_rc({d_w.rc, pts.rc}));
  MM_TRY(d_w.up(w.data(), len));
  MM_TRY(mmh_fc_encode_weights(p, d_w.p, pts.p, nullptr));
  encoded_share.assign(n_out, std::vector<mm::he::Plaintext>(n_in));
  for (size_t j = 0; j < n_out; ++j)
    for (size_t i = 0; i < n_in; ++i) {
      encoded_share[j][i].coeffs.resize(N);
      MM_TRY(pts.down(encoded_share[j][i].data(), N, (j * n_in + i) * N));
    }
  return Code::OK;
}

// :394-471 (shipped USE_APPROX_RESHARE build)
Code HomFCSS::matVecMul(const std::vector<std::vector<mm::he::Plaintext>> &matrix,
                        const std::vector<mm::he::Ciphertext> &vec_share0,
                        const std::vector<mm::he::Plaintext> &vec_share1, const Meta &meta,
                        std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1, size_t) const {
  MM_ENSURE(context_ && keys_ && keys_->pk_last, Code::ERR_CONFIG);
  mmh_fc_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INVALID_ARG);
  size_t n_in = 0, n_out = 0;
  MM_TRY(mmh_fc_plan_info(p, nullptr, &n_in, &n_out));
  MM_ENSURE(vec_share0.size() == n_in, Code::ERR_INVALID_ARG);
  MM_ENSURE(matrix.size() == n_out, Code::ERR_INVALID_ARG);
  for (const auto &c : matrix) MM_ENSURE(c.size() == n_in, Code::ERR_INVALID_ARG);
  const size_t N = poly_degree(), L = context_->coeff_modulus_size(), ct = 2 * L * N;
  for (const auto &rows : matrix)
    for (const auto &sub : rows) MM_ENSURE(sub.coeff_count() == N && !sub.is_zero(), Code::ERR_INVALID_ARG);  // :414-421
  if (meta.is_shared_input) MM_ENSURE(vec_share1.size() == n_in, Code::ERR_DIM_MISMATCH);
  const size_t rows = (size_t)meta.weight_shape.rows();
  Dev cts(n_in * ct), share(meta.is_shared_input ? n_in * N : 1), wpts(n_out * n_in * N);
  Dev u(n_out * N), e1(n_out * N), mask(n_out * N), out_cts(n_out * 2 * N), out_share(rows);
  MM_TRY(first_rc({cts.rc, share.rc, wpts.rc, u.rc, e1.rc, mask.rc, out_cts.rc, out_share.rc}));
  for (size_t i = 0; i < n_in; ++i) {
    MM_ENSURE(vec_share0[i].size() == 2 && vec_share0[i].coeff_modulus_size() == L &&
                  vec_share0[i].poly_modulus_degree() == N,
              Code::ERR_INTERNAL);
    MM_TRY(cts.up(vec_share0[i].data(), ct, i * ct));
    if (meta.is_shared_input) {
      MM_ENSURE(vec_share1[i].coeff_count() == N, Code::ERR_INTERNAL);
      MM_TRY(share.up(vec_share1[i].data(), N, i * N));
    }
  }
  for (size_t j = 0; j < n_out; ++j)
    for (size_t i = 0; i < n_in; ++i) MM_TRY(wpts.up(matrix[j][i].data(), N, (j * n_in + i) * N));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 3;
  }
  MM_TRY(mmh_sample_ternary(prng_key_, sid, (int64_t *)u.p, n_out * N, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e1.p, n_out * N, nullptr));
  MM_TRY(mmh_sample_uniform(context_->raw(), 1, prng_key_, sid + 2, mask.p, n_out, nullptr));
  MM_TRY(mmh_fc_server(p, cts.p, meta.is_shared_input ? share.p : nullptr, wpts.p, keys_->pk_last,
                       (const int64_t *)u.p, (const int64_t *)e1.p, mask.p, out_cts.p, out_share.p, nullptr));
  out_share0.resize(n_out);
  for (size_t k = 0; k < n_out; ++k) {
    out_share0[k].resize(2, 1, N);
    MM_TRY(out_cts.down(out_share0[k].data(), 2 * N, k * 2 * N));
  }
  const TensorShape oshape({(int64_t)rows});
  if (!out_share1.shape().IsSameSize(oshape)) out_share1.Reshape(oshape);
  MM_TRY(out_share.down(out_share1.data(), rows));
  return Code::OK;
}

// :572-602
Code HomFCSS::decryptToVector(const std::vector<mm::he::Ciphertext> &enc_vector, const Meta &meta,
                              Tensor<uint64_t> &out, size_t) const {
  MM_ENSURE(context_ && keys_ && keys_->sk_last, Code::ERR_CONFIG);
  mmh_fc_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INVALID_ARG);
  size_t n_out = 0;
  MM_TRY(mmh_fc_plan_info(p, nullptr, nullptr, &n_out));
  MM_ENSURE(enc_vector.size() == n_out, Code::ERR_INVALID_ARG);
  const size_t N = poly_degree(), rows = (size_t)meta.weight_shape.rows();
  Dev cts(n_out * 2 * N), res(rows);
  MM_TRY(first_rc({cts.rc, res.rc}));
  for (size_t k = 0; k < n_out; ++k) {
    MM_ENSURE(enc_vector[k].size() == 2 && enc_vector[k].coeff_modulus_size() == 1 &&
                  enc_vector[k].poly_modulus_degree() == N,
              Code::ERR_DIM_MISMATCH);
    MM_TRY(cts.up(enc_vector[k].data(), 2 * N, k * 2 * N));
  }
  MM_TRY(mmh_fc_decrypt(p, keys_->sk_last, cts.p, res.p, nullptr));
  const TensorShape oshape({(int64_t)rows});
  if (!out.shape().IsSameSize(oshape)) out.Reshape(oshape);
  MM_TRY(res.down(out.data(), rows));
  return Code::OK;
}

Code HomFCSS::idealFunctionality(const Tensor<uint64_t> &w, const Tensor<uint64_t> &vec, const Meta &meta,
                                 Tensor<uint64_t> &out) const {
  MM_ENSURE(w.shape().IsSameSize(meta.weight_shape) && vec.shape().IsSameSize(meta.input_shape),
            Code::ERR_DIM_MISMATCH);
  MM_ENSURE(w.cols() == vec.length(), Code::ERR_DIM_MISMATCH);
  const uint64_t t = plain_modulus();
  MM_ENSURE(t != (uint64_t)-1 && t > 1, Code::ERR_INTERNAL);
  out.Reshape(TensorShape({w.rows()}));
  for (int64_t r = 0; r < w.rows(); ++r) {
    unsigned __int128 sum = 0;
    for (int64_t c = 0; c < w.cols(); ++c)
      sum += (uint64_t)((unsigned __int128)(w(r, c) % t) * (vec(c) % t) % t);
    out(r) = (uint64_t)(sum % t);
  }
  return Code::OK;
}

}  // namespace gemini
