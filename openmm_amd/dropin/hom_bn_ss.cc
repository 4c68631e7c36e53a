// gemini::HomBNSS -- the member functions the REFERENCE'S header declares (include/gemini/cheetah/hom_bn_ss.h:25-198,
// compiled from /root/reference, unchanged) on the seal:: facade types, one libmm_hip layer call per method: the SIMD
// path (BOLE / bn over BatchEncoder, hom_bn_ss.cc:151-406) and the coefficient-packed direct path (:471-790).  Linked in
// place of include/gemini/cheetah/hom_bn_ss.cc; checks and return codes follow that file, cited per method.
#include "gemini/cheetah/hom_bn_ss.h"

#include <algorithm>

#include "dropin_common.h"
#include "gemini/core/common.h"
#include "gemini/core/logging.h"

namespace gemini {
using namespace mm_dropin;

namespace {
Fast *fast_of(const std::shared_ptr<seal::Evaluator> &ev) { return ev ? static_cast<Fast *>(ev->facade_ext().get()) : nullptr; }
std::shared_ptr<Fast> fast_ptr(const std::shared_ptr<seal::Evaluator> &ev) {
  return std::static_pointer_cast<Fast>(ev->facade_ext());
}
size_t n_simd_cts(const HomBNSS::Meta &meta, size_t N) {
  const size_t n = (size_t)meta.vec_shape.length(), sub = std::min(n, N);
  return sub ? (n + sub - 1) / sub : 0;
}
std::array<std::int64_t, 10> vec_tag(const HomBNSS::Meta &m) { return {m.vec_shape.length(), 0, 0, 0, 0, 0, 0, 0, 0, 1}; }
std::array<std::int64_t, 10> tensor_tag(const HomBNSS::Meta &m) {
  return {m.ishape.channels(), m.ishape.height(), m.ishape.width(), 0, 0, 0, 0, 0, 0, 2};
}
mmh_bn_tensor_meta tensor_meta(const HomBNSS::Meta &meta) {
  return mmh_bn_tensor_meta{meta.ishape.channels(), meta.ishape.height(), meta.ishape.width(), meta.is_shared_input};
}
// plaintexts of one device batch, or uploaded from the host objects: -> device pointer of [n][N]
Code plain_operand(Fast &f, const std::vector<seal::Plaintext> &v, size_t n, int kind,
                   const std::array<std::int64_t, 10> &tag, Dev &tmp, const u64 **out) {
  DI_ENSURE(v.size() == n, Code::ERR_INVALID_ARG);
  const PlainBatch *b = whole_batch(v.data(), n, kind, tag);
  if (b && b->fast.get() == &f && b->pts) {
    *out = b->pts.p;
    return Code::OK;
  }
  tmp = f.mk(n * f.N);
  DI_ENSURE(tmp, Code::ERR_SEAL_MEMORY);
  DI_TRY(mmh_memset(tmp.p, 0, n * f.N * 8, f.stream));
  for (size_t i = 0; i < n; ++i) {
    DI_ENSURE(v[i].coeff_count() <= f.N && !v[i].is_ntt_form(), Code::ERR_INTERNAL);
    DI_TRY(f.up(tmp, v[i].data(), v[i].coeff_count(), i * f.N));
  }
  *out = tmp.p;
  return Code::OK;
}
}  // namespace

// hom_bn_ss.cc:69-105
Code HomBNSS::setUp_Base(uint64_t target_base_mod, const seal::SEALContext &direct_context,
                         std::optional<seal::SecretKey> direct_sk, std::shared_ptr<seal::PublicKey> direct_pk) {
  direct_context_ = std::make_shared<seal::SEALContext>(direct_context);
  ENSURE_OR_RETURN(direct_context_, Code::ERR_NULL_POINTER);
  auto parms = direct_context_->first_context_data()->parms();
  scheme_ = parms.scheme();
  target_base_mod_ = target_base_mod;
  poly_degree_ = parms.poly_modulus_degree();
  if (target_base_mod_ != parms.plain_modulus().value()) {
    LOG(WARNING) << "HomBNSS: invalid target base mod this SEALContext";
    return Code::ERR_INVALID_ARG;
  }
  if (direct_sk) {
    if (!seal::is_metadata_valid_for(*direct_sk, *direct_context_)) {
      LOG(WARNING) << "HomBNSS: invalid secret key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    direct_sk_ = seal::SecretKey(*direct_sk);
    direct_encryptor_ = std::make_shared<seal::Encryptor>(*direct_context_, *direct_sk);
  }
  if (direct_pk) {
    if (!seal::is_metadata_valid_for(*direct_pk, *direct_context_)) {
      LOG(WARNING) << "HomBNSS: invalid public key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    direct_pk_encryptor_ = std::make_shared<seal::Encryptor>(*direct_context_, *direct_pk);
  }
  direct_evaluator_ = std::make_shared<seal::Evaluator>(*direct_context_);
  auto f = std::make_shared<Fast>();
  CHECK_ERR(f->init(*direct_context_, direct_sk ? &*direct_sk : nullptr, direct_pk.get()), "HomBNSS::setUp_Base");
  direct_evaluator_->facade_ext() = f;
  return Code::OK;
}

// :107-149
Code HomBNSS::setUp_Field(uint64_t target_field_mod, const seal::SEALContext &simd_context,
                          std::optional<seal::SecretKey> simd_sk, std::shared_ptr<seal::PublicKey> simd_pk) {
  simd_context_ = std::make_shared<seal::SEALContext>(simd_context);
  ENSURE_OR_RETURN(simd_context_, Code::ERR_NULL_POINTER);
  auto parms = simd_context_->first_context_data()->parms();
  target_field_mod_ = target_field_mod;
  if (target_field_mod_ != parms.plain_modulus().value()) {
    LOG(WARNING) << "HomBNSS: invalid target field mod this SEALContext";
    return Code::ERR_INVALID_ARG;
  }
  if (!simd_context_->first_context_data()->qualifiers().using_batching) {
    LOG(WARNING) << "HomBNSS: invalid SIMD modulus";
    return Code::ERR_INVALID_ARG;
  }
  simd_encoder_ = std::make_shared<seal::BatchEncoder>(*simd_context_);
  if (simd_sk) {
    if (!seal::is_metadata_valid_for(*simd_sk, *simd_context_)) {
      LOG(WARNING) << "HomBNSS: invalid secret key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    simd_sk_ = seal::SecretKey(*simd_sk);
    simd_encryptor_ = std::make_shared<seal::Encryptor>(*simd_context_, *simd_sk);
  }
  if (simd_pk) {
    if (!seal::is_metadata_valid_for(*simd_pk, *simd_context_)) {
      LOG(WARNING) << "HomBNSS: invalid public key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    simd_encryptor_ = std::make_shared<seal::Encryptor>(*simd_context_, *simd_pk);
  }
  simd_evaluator_ = std::make_shared<seal::Evaluator>(*simd_context_);
  if (!poly_degree_) poly_degree_ = parms.poly_modulus_degree();  // (setUp_Base sets it when it runs; same N)
  auto f = std::make_shared<Fast>();
  CHECK_ERR(f->init(*simd_context_, simd_sk ? &*simd_sk : nullptr, simd_pk.get()), "HomBNSS::setUp_Field");
  simd_evaluator_->facade_ext() = f;
  return Code::OK;
}

uint64_t HomBNSS::base_plain_modulus() const {
  return direct_context_ ? direct_context_->first_context_data()->parms().plain_modulus().value() : -1;
}
uint64_t HomBNSS::field_plain_modulus() const {
  return simd_context_ ? simd_context_->first_context_data()->parms().plain_modulus().value() : -1;
}

// :151-201
Code HomBNSS::encryptVector(const Tensor<uint64_t> &in_vec, const Meta &meta,
                            std::vector<seal::Serializable<seal::Ciphertext>> &out, size_t nthreads) const {
  Fast *f = fast_of(simd_evaluator_);
  ENSURE_OR_RETURN(simd_context_ && simd_sk_ && f && f->sk, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  ENSURE_OR_RETURN(in_vec.shape().IsSameSize(meta.vec_shape), Code::ERR_DIM_MISMATCH);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const size_t N = f->N, L = f->L, n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  Dev vec = f->mk(n), e = f->mk(n_ct * N), cts = f->mk(n_ct * 2 * L * N), a;
  ENSURE_OR_RETURN(vec && e && cts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(vec, in_vec.data(), n));
  std::vector<std::array<std::uint8_t, seal::wire::kSeedBytes>> seeds;
  CHECK_ERR(fresh_a(*f, *simd_context_, n_ct, nthreads, a, seeds), "encryptVector");
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 1);
  DI_TRY(mmh_sample_noise(key, sid, (std::int64_t *)e.p, n_ct * N, f->stream));
  DI_TRY(mmh_bn_encrypt_vector(f->ctx, f->sk.p, vec.p, n, a.p, (const std::int64_t *)e.p, cts.p, f->stream));
  return download_fresh(*f, *simd_context_, cts, n_ct, seeds, out);
}

// :203-233: BatchEncoder::encode per chunk -> one call; the plaintexts stay on the device
Code HomBNSS::encodeVector(const Tensor<uint64_t> &in_vec, const Meta &meta, std::vector<seal::Plaintext> &out,
                           size_t) const {
  Fast *f = fast_of(simd_evaluator_);
  ENSURE_OR_RETURN(simd_context_ && f, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  ENSURE_OR_RETURN(in_vec.shape().IsSameSize(meta.vec_shape), Code::ERR_DIM_MISMATCH);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const size_t N = f->N, n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  auto batch = std::make_shared<PlainBatch>();
  batch->fast = fast_ptr(simd_evaluator_);
  batch->kind = kKindBnVector, batch->tag = vec_tag(meta), batch->count = n_ct, batch->N = N;
  Dev vec = f->mk(n);
  batch->pts = f->mk(n_ct * N);
  ENSURE_OR_RETURN(vec && batch->pts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(vec, in_vec.data(), n));
  DI_TRY(mmh_bn_encode_vector(f->ctx, vec.p, n, batch->pts.p, f->stream));
  DI_TRY(f->sync());
  out.assign(n_ct, seal::Plaintext());
  for (size_t i = 0; i < n_ct; ++i) out[i].facade_set_lazy(batch, i);
  return Code::OK;
}

// :235-239
Code HomBNSS::encodeScales(const Tensor<uint64_t> &scales, const Meta &meta, std::vector<seal::Plaintext> &out,
                           size_t nthreads) const {
  return encodeVector(scales, meta, out, nthreads);
}

// :241-278 + addMask :318-362
Code HomBNSS::bn(const std::vector<seal::Ciphertext> &vec_share0, const std::vector<seal::Plaintext> &vec_share1,
                 const std::vector<seal::Plaintext> &scales, const Meta &meta, std::vector<seal::Ciphertext> &out_share0,
                 Tensor<uint64_t> &out_share1, size_t) const {
  ENSURE_OR_RETURN(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  Fast *f = fast_of(simd_evaluator_);
  ENSURE_OR_RETURN(simd_context_ && f, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const size_t N = f->N, L = f->L, n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  ENSURE_OR_RETURN(vec_share0.size() == n_ct, Code::ERR_INVALID_ARG);
  ENSURE_OR_RETURN(scales.size() == n_ct, Code::ERR_INVALID_ARG);
  if (meta.is_shared_input) ENSURE_OR_RETURN(vec_share1.size() == n_ct, Code::ERR_INVALID_ARG);
  Dev cts, t_in, t_sc;
  const u64 *d_in = nullptr, *d_sc = nullptr;
  {
    const Code c = upload_cts(*f, vec_share0, L, cts);
    if (c != Code::OK) return c == Code::ERR_INVALID_ARG ? Code::ERR_INTERNAL : c;
  }
  if (meta.is_shared_input) CHECK_ERR(plain_operand(*f, vec_share1, n_ct, kKindBnVector, vec_tag(meta), t_in, &d_in), "bn");
  CHECK_ERR(plain_operand(*f, scales, n_ct, kKindBnVector, vec_tag(meta), t_sc, &d_sc), "bn");
  Dev rnd = f->mk(n), out_cts = f->mk(n_ct * 2 * N), share = f->mk(n);
  ENSURE_OR_RETURN(rnd && out_cts && share, Code::ERR_SEAL_MEMORY);
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 1);   // addMask's r (:329-334)
  DI_TRY(mmh_chacha20_words(key, sid, 0, rnd.p, n, f->stream));
  DI_TRY(mmh_bn_server(f->ctx, cts.p, d_in, d_sc, n, rnd.p, out_cts.p, share.p, f->stream));
  out_share1.Reshape(meta.vec_shape);
  DI_TRY(f->down(out_share1.data(), share, n));
  return download_last_level(*f, *simd_context_, out_cts, n_ct, out_share0);
}

// :280-316 + addMask.  The reference multiplies into encrypted_x / encrypted_y in place (:304-305); here they are read
// only (a caller cannot tell: the reference leaves them in an unspecified, consumed state).
Code HomBNSS::bole(std::vector<seal::Ciphertext> &encrypted_x, std::vector<seal::Ciphertext> &encrypted_y,
                   const std::vector<seal::Plaintext> &encoded_x, const std::vector<seal::Plaintext> &encoded_y,
                   const Meta &meta, std::vector<seal::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1,
                   size_t) const {
  ENSURE_OR_RETURN(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  Fast *f = fast_of(simd_evaluator_);
  ENSURE_OR_RETURN(simd_context_ && f, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const size_t N = f->N, L = f->L, n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, N);
  ENSURE_OR_RETURN(encrypted_x.size() == n_ct && encrypted_y.size() == n_ct, Code::ERR_INVALID_ARG);
  ENSURE_OR_RETURN(encoded_x.size() == n_ct && encoded_y.size() == n_ct, Code::ERR_INVALID_ARG);
  // [ex ; ey] and [py ; px]: out = ex * py + ey * px
  Dev cts, pts = f->mk(2 * n_ct * N), tx, ty;
  ENSURE_OR_RETURN(pts, Code::ERR_SEAL_MEMORY);
  {
    const Code c = upload_cts(*f, {&encrypted_x, &encrypted_y}, L, cts);
    if (c != Code::OK) return c == Code::ERR_INVALID_ARG ? Code::ERR_INTERNAL : c;
  }
  const u64 *px = nullptr, *py = nullptr;
  CHECK_ERR(plain_operand(*f, encoded_x, n_ct, kKindBnVector, vec_tag(meta), tx, &px), "bole");
  CHECK_ERR(plain_operand(*f, encoded_y, n_ct, kKindBnVector, vec_tag(meta), ty, &py), "bole");
  DI_TRY(mmh_memcpy_d2d(pts.p, py, n_ct * N * 8, f->stream));
  DI_TRY(mmh_memcpy_d2d(pts.p + n_ct * N, px, n_ct * N * 8, f->stream));
  Dev rnd = f->mk(n), out_cts = f->mk(n_ct * 2 * N), share = f->mk(n);
  ENSURE_OR_RETURN(rnd && out_cts && share, Code::ERR_SEAL_MEMORY);
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 1);
  DI_TRY(mmh_chacha20_words(key, sid, 0, rnd.p, n, f->stream));
  DI_TRY(mmh_bole_server(f->ctx, cts.p, pts.p, n, rnd.p, out_cts.p, share.p, f->stream));
  out_share1.Reshape(meta.vec_shape);
  DI_TRY(f->down(out_share1.data(), share, n));
  return download_last_level(*f, *simd_context_, out_cts, n_ct, out_share0);
}

// :364-406
Code HomBNSS::decryptToVector(const std::vector<seal::Ciphertext> &in_vec, const Meta &meta, Tensor<uint64_t> &out_vec,
                              size_t) const {
  ENSURE_OR_RETURN(meta.target_base_mod == target_field_mod_, Code::ERR_OUT_BOUND);
  Fast *f = fast_of(simd_evaluator_);
  ENSURE_OR_RETURN(simd_context_ && simd_sk_ && f && f->sk_last, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const size_t n = (size_t)meta.vec_shape.length(), n_ct = n_simd_cts(meta, f->N);
  ENSURE_OR_RETURN(in_vec.size() == n_ct, Code::ERR_INVALID_ARG);
  for (const auto &c : in_vec) ENSURE_OR_RETURN(c.parms_id() == simd_context_->last_parms_id(), Code::ERR_INTERNAL);
  Dev cts;
  CHECK_ERR(upload_cts(*f, in_vec, 1, cts), "decryptToVector");
  Dev res = f->mk(n);
  ENSURE_OR_RETURN(res, Code::ERR_SEAL_MEMORY);
  DI_TRY(mmh_bn_decrypt_vector(f->ctx, f->sk_last.p, cts.p, n, res.p, f->stream));
  out_vec.Reshape(meta.vec_shape);
  DI_TRY(f->down(out_vec.data(), res, n));
  DI_TRY(f->sync());
  return Code::OK;
}

// :471-520
Code HomBNSS::encodeTensor(const Tensor<uint64_t> &in_tensor, const Meta &meta, std::vector<seal::Plaintext> &out,
                           size_t) const {
  Fast *f = fast_of(direct_evaluator_);
  ENSURE_OR_RETURN(direct_context_ && f, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(in_tensor.shape().IsSameSize(meta.ishape) && meta.ishape.dims() == 3, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  DI_TRY(mmh_bn_tensor_count(f->ctx, &m, &n_ct, nullptr));
  const size_t N = f->N, words = (size_t)in_tensor.NumElements();
  auto batch = std::make_shared<PlainBatch>();
  batch->fast = fast_ptr(direct_evaluator_);
  batch->kind = kKindBnTensor, batch->tag = tensor_tag(meta), batch->count = n_ct, batch->N = N;
  Dev x = f->mk(words);
  batch->pts = f->mk(n_ct * N);
  ENSURE_OR_RETURN(x && batch->pts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(x, in_tensor.data(), words));
  DI_TRY(mmh_bn_encode_tensor(f->ctx, &m, x.p, batch->pts.p, f->stream));
  DI_TRY(f->sync());
  out.assign(n_ct, seal::Plaintext());
  for (size_t i = 0; i < n_ct; ++i) out[i].facade_set_lazy(batch, i);
  return Code::OK;
}

// :522-576
Code HomBNSS::encryptTensor(const Tensor<uint64_t> &in_tensor, const Meta &meta,
                            std::vector<seal::Serializable<seal::Ciphertext>> &out, size_t nthreads) const {
  Fast *f = fast_of(direct_evaluator_);
  ENSURE_OR_RETURN(direct_context_ && direct_sk_ && f && f->sk, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(in_tensor.shape().IsSameSize(meta.ishape) && meta.ishape.dims() == 3, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  DI_TRY(mmh_bn_tensor_count(f->ctx, &m, &n_ct, nullptr));
  const size_t N = f->N, L = f->L, words = (size_t)in_tensor.NumElements();
  Dev x = f->mk(words), e = f->mk(n_ct * N), cts = f->mk(n_ct * 2 * L * N), a;
  ENSURE_OR_RETURN(x && e && cts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(x, in_tensor.data(), words));
  std::vector<std::array<std::uint8_t, seal::wire::kSeedBytes>> seeds;
  CHECK_ERR(fresh_a(*f, *direct_context_, n_ct, nthreads, a, seeds), "encryptTensor");
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 1);
  DI_TRY(mmh_sample_noise(key, sid, (std::int64_t *)e.p, n_ct * N, f->stream));
  DI_TRY(mmh_bn_encrypt_tensor(f->ctx, &m, f->sk.p, x.p, a.p, (const std::int64_t *)e.p, cts.p, f->stream));
  return download_fresh(*f, *direct_context_, cts, n_ct, seeds, out);
}

// :578-738
Code HomBNSS::bn_direct(const std::vector<seal::Ciphertext> &tensor_share0, const std::vector<seal::Plaintext> &tensor_share1,
                        const Tensor<uint64_t> &scales, const Meta &meta, std::vector<seal::Ciphertext> &out_share0,
                        Tensor<uint64_t> &out_share1, size_t) const {
  Fast *f = fast_of(direct_evaluator_);
  ENSURE_OR_RETURN(direct_context_ && direct_pk_encryptor_ && f && f->pk, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(meta.ishape.dims() == 3, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(scales.dims() == 1 && scales.length() == meta.ishape.channels(), Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  DI_TRY(mmh_bn_tensor_count(f->ctx, &m, &n_ct, nullptr));
  ENSURE_OR_RETURN(tensor_share0.size() == n_ct, Code::ERR_DIM_MISMATCH);
  if (meta.is_shared_input) ENSURE_OR_RETURN(tensor_share1.size() == n_ct, Code::ERR_DIM_MISMATCH);
  const size_t N = f->N, L = f->L, words = (size_t)meta.ishape.num_elements(), C = (size_t)scales.length();
  Dev cts, t_in;
  const u64 *d_in = nullptr;
  {
    const Code c = upload_cts(*f, tensor_share0, L, cts);
    if (c != Code::OK) return c == Code::ERR_INVALID_ARG ? Code::ERR_INTERNAL : c;
  }
  if (meta.is_shared_input)
    CHECK_ERR(plain_operand(*f, tensor_share1, n_ct, kKindBnTensor, tensor_tag(meta), t_in, &d_in), "bn_direct");
  Dev d_scales = f->mk(C), u = f->mk(n_ct * N), e0 = f->mk(n_ct * N), e1 = f->mk(n_ct * N), share1 = f->mk(words);
  Dev out_cts = f->mk(n_ct * 2 * N);
  ENSURE_OR_RETURN(d_scales && u && e0 && e1 && share1 && out_cts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(d_scales, scales.data(), C));
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 4);
  DI_TRY(mmh_sample_ternary(key, sid, (std::int64_t *)u.p, n_ct * N, f->stream));
  DI_TRY(mmh_sample_noise(key, sid + 1, (std::int64_t *)e0.p, n_ct * N, f->stream));
  DI_TRY(mmh_sample_noise(key, sid + 2, (std::int64_t *)e1.p, n_ct * N, f->stream));
  DI_TRY(mmh_chacha20_words(key, sid + 3, 0, share1.p, words, f->stream));
  DI_TRY(mmh_bn_direct_server(f->ctx, &m, cts.p, d_in, d_scales.p, f->pk.p, (const std::int64_t *)u.p,
                              (const std::int64_t *)e0.p, (const std::int64_t *)e1.p, share1.p, out_cts.p, f->stream));
  out_share1.Reshape(meta.ishape);
  DI_TRY(f->down(out_share1.data(), share1, words));  // :712-723: the random tensor, reduced mod the ring below
  const Code c = download_last_level(*f, *direct_context_, out_cts, n_ct, out_share0);
  const uint64_t t = base_plain_modulus();
  for (size_t i = 0; i < words; ++i)
    out_share1.data()[i] = (t & (t - 1)) ? out_share1.data()[i] % t : out_share1.data()[i] & (t - 1);
  return c;
}

// :740-790
Code HomBNSS::decryptToTensor(const std::vector<seal::Ciphertext> &in_vec, const Meta &meta, Tensor<uint64_t> &out_tensor,
                              size_t) const {
  Fast *f = fast_of(direct_evaluator_);
  ENSURE_OR_RETURN(direct_context_ && direct_sk_ && f && f->sk_last, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(meta.ishape.dims() == 3, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  const mmh_bn_tensor_meta m = tensor_meta(meta);
  size_t n_ct = 0;
  DI_TRY(mmh_bn_tensor_count(f->ctx, &m, &n_ct, nullptr));
  ENSURE_OR_RETURN(in_vec.size() == n_ct, Code::ERR_DIM_MISMATCH);
  for (const auto &c : in_vec) ENSURE_OR_RETURN(c.parms_id() == direct_context_->last_parms_id(), Code::ERR_INTERNAL);
  const size_t words = (size_t)meta.ishape.num_elements();
  Dev cts;
  CHECK_ERR(upload_cts(*f, in_vec, 1, cts), "decryptToTensor");
  Dev res = f->mk(words);
  ENSURE_OR_RETURN(res, Code::ERR_SEAL_MEMORY);
  DI_TRY(mmh_bn_decrypt_tensor(f->ctx, &m, f->sk_last.p, cts.p, res.p, f->stream));
  out_tensor.Reshape(meta.ishape);
  DI_TRY(f->down(out_tensor.data(), res, words));
  DI_TRY(f->sync());
  return Code::OK;
}

}  // namespace gemini
