// gemini::HomFCSS -- the member functions the REFERENCE'S header declares (include/gemini/cheetah/hom_fc_ss.h:27-121,
// compiled from /root/reference, unchanged) on the seal:: facade types, one libmm_hip layer call per method.  Linked in
// place of include/gemini/cheetah/hom_fc_ss.cc; checks and return codes follow that file, cited per method.
#include "gemini/cheetah/hom_fc_ss.h"

#include <algorithm>

#include "dropin_common.h"
#include "gemini/core/common.h"
#include "gemini/core/logging.h"

#ifndef USE_APPROX_RESHARE
#define USE_APPROX_RESHARE 0
#endif

namespace gemini {
using namespace mm_dropin;

namespace {
struct FcFast : Fast {
  std::map<std::array<std::int64_t, 10>, mmh_fc_plan *> plans;
  ~FcFast() {
    if (stream) (void)mmh_stream_sync(stream);
    for (auto &kv : plans) (void)mmh_fc_plan_destroy(kv.second);
  }
};
struct PlanInfo {
  mmh_fc_plan *plan = nullptr;
  std::int64_t split[2] = {0, 0};
  std::size_t n_in = 0, n_out = 0;
};
std::array<std::int64_t, 10> tag_of(const HomFCSS::Meta &m) {
  return {m.weight_shape.rows(), m.weight_shape.cols(), m.is_shared_input, 0, 0, 0, 0, 0, 0, 0};
}
Code plan_for(FcFast &f, const HomFCSS::Meta &meta, PlanInfo &pi) {
  DI_ENSURE(meta.weight_shape.dims() == 2 && meta.input_shape.dims() == 1 &&
                meta.weight_shape.cols() == meta.input_shape.length(),
            Code::ERR_INVALID_ARG);
  const auto tag = tag_of(meta);
  auto it = f.plans.find(tag);
  if (it == f.plans.end()) {
    mmh_fc_meta m{tag[0], tag[1], (std::int32_t)tag[2]};
    mmh_fc_plan *p = nullptr;
    DI_TRY(mmh_fc_plan_create(f.ctx, &m, &p));
    it = f.plans.emplace(tag, p).first;
  }
  pi.plan = it->second;
  DI_TRY(mmh_fc_plan_info(pi.plan, pi.split, &pi.n_in, &pi.n_out));
  return Code::OK;
}
FcFast *fast_of(const std::shared_ptr<seal::Evaluator> &ev) {
  return ev ? static_cast<FcFast *>(ev->facade_ext().get()) : nullptr;
}
}  // namespace

// hom_fc_ss.cc:105-127
seal::scheme_type HomFCSS::scheme() const {
  return context_ ? context_->first_context_data()->parms().scheme() : seal::scheme_type::none;
}
size_t HomFCSS::poly_degree() const {
  return context_ ? context_->first_context_data()->parms().poly_modulus_degree() : 0;
}
uint64_t HomFCSS::plain_modulus() const {
  return context_ ? context_->first_context_data()->parms().plain_modulus().value() : -1;
}

// :129-155
Code HomFCSS::setUp(const seal::SEALContext &context, std::optional<seal::SecretKey> sk,
                    std::shared_ptr<seal::PublicKey> pk) {
  context_ = std::make_shared<seal::SEALContext>(context);
  ENSURE_OR_RETURN(context_, Code::ERR_NULL_POINTER);
  if (sk) {
    if (!seal::is_metadata_valid_for(*sk, *context_)) {
      LOG(WARNING) << "HomFCSS: invalid secret key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    sk_ = seal::SecretKey(*sk);
    encryptor_ = std::make_shared<seal::Encryptor>(*context_, *sk);
  }
  if (pk) {
    if (!seal::is_metadata_valid_for(*pk, *context_)) {
      LOG(WARNING) << "HomFCSS: invalid public key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    pk_ = std::make_shared<seal::PublicKey>(*pk);
  }
  evaluator_ = std::make_shared<seal::Evaluator>(*context_);
  auto f = std::make_shared<FcFast>();
  CHECK_ERR(f->init(*context_, sk ? &*sk : nullptr, pk.get()), "HomFCSS::setUp");
  evaluator_->facade_ext() = f;
  return Code::OK;
}

// :214-273
Code HomFCSS::encryptInputVector(const Tensor<uint64_t> &vec, const Meta &meta,
                                 std::vector<seal::Serializable<seal::Ciphertext>> &encrypted_share,
                                 size_t nthreads) const {
  FcFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && encryptor_ && f && f->sk, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(vec.shape().IsSameSize(meta.input_shape), Code::ERR_DIM_MISMATCH);
  ENSURE_OR_RETURN(meta.input_shape.num_elements() > 0, Code::ERR_INVALID_ARG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  PlanInfo pi;
  CHECK_ERR(plan_for(*f, meta, pi), "encryptInputVector");
  const size_t N = f->N, L = f->L, len = (size_t)vec.NumElements();
  Dev d_vec = f->mk(len), e = f->mk(pi.n_in * N), cts = f->mk(pi.n_in * 2 * L * N), a;
  ENSURE_OR_RETURN(d_vec && e && cts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(d_vec, vec.data(), len));
  std::vector<std::array<std::uint8_t, seal::wire::kSeedBytes>> seeds;
  CHECK_ERR(fresh_a(*f, *context_, pi.n_in, nthreads, a, seeds), "encryptInputVector");
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 1);
  DI_TRY(mmh_sample_noise(key, sid, (std::int64_t *)e.p, pi.n_in * N, f->stream));
  DI_TRY(mmh_fc_encrypt_input(pi.plan, f->sk.p, d_vec.p, a.p, (const std::int64_t *)e.p, cts.p, f->stream));
  return download_fresh(*f, *context_, cts, pi.n_in, seeds, encrypted_share);
}

// :275-325: the encoded share stays on the device
Code HomFCSS::encodeInputVector(const Tensor<uint64_t> &vec, const Meta &meta, std::vector<seal::Plaintext> &encoded_share,
                                size_t) const {
  FcFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && f, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(vec.shape().IsSameSize(meta.input_shape), Code::ERR_DIM_MISMATCH);
  ENSURE_OR_RETURN(meta.input_shape.num_elements() > 0, Code::ERR_INVALID_ARG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  PlanInfo pi;
  CHECK_ERR(plan_for(*f, meta, pi), "encodeInputVector");
  const size_t N = f->N, len = (size_t)vec.NumElements();
  auto batch = std::make_shared<PlainBatch>();
  batch->fast = std::static_pointer_cast<Fast>(evaluator_->facade_ext());
  batch->kind = kKindFcInput, batch->tag = tag_of(meta), batch->count = pi.n_in, batch->N = N;
  Dev d_vec = f->mk(len);
  batch->pts = f->mk(pi.n_in * N);
  ENSURE_OR_RETURN(d_vec && batch->pts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(d_vec, vec.data(), len));
  DI_TRY(mmh_fc_encode_input(pi.plan, d_vec.p, batch->pts.p, f->stream));
  DI_TRY(f->sync());
  encoded_share.assign(pi.n_in, seal::Plaintext());
  for (size_t i = 0; i < pi.n_in; ++i) encoded_share[i].facade_set_lazy(batch, i);
  return Code::OK;
}

// :327-392: [n_out][n_in] block plaintexts, encoded on the device and left there
Code HomFCSS::encodeWeightMatrix(const Tensor<uint64_t> &w, const Meta &meta,
                                 std::vector<std::vector<seal::Plaintext>> &encoded_share, size_t) const {
  FcFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && f, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(w.shape().IsSameSize(meta.weight_shape), Code::ERR_DIM_MISMATCH);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  PlanInfo pi;
  CHECK_ERR(plan_for(*f, meta, pi), "encodeWeightMatrix");
  const size_t N = f->N, len = (size_t)w.NumElements(), rows = (size_t)w.rows(), cols = (size_t)w.cols();
  auto batch = std::make_shared<PlainBatch>();
  batch->fast = std::static_pointer_cast<Fast>(evaluator_->facade_ext());
  batch->kind = kKindFcWeights, batch->tag = tag_of(meta), batch->count = pi.n_out * pi.n_in, batch->N = N;
  Dev d_w = f->mk(len);
  batch->pts = f->mk(batch->count * N);
  ENSURE_OR_RETURN(d_w && batch->pts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(d_w, w.data(), len));
  DI_TRY(mmh_fc_encode_weights(pi.plan, d_w.p, batch->pts.p, f->stream));
  // a block's plaintext is zero exactly when every weight of the block is zero mod t: what matVecMul's is_zero test
  // (:414-421) asks; settled here while the copy runs
  const u64 t = plain_modulus();
  batch->zero.assign(batch->count, 1);
  const u64 *wd = w.data();
  for (size_t r = 0; r < rows; ++r) {
    std::uint8_t *row = batch->zero.data() + (r / (size_t)pi.split[0]) * pi.n_in;
    for (size_t cb = 0; cb < pi.n_in; ++cb) {
      if (!row[cb]) continue;
      const size_t c1 = std::min<size_t>((cb + 1) * (size_t)pi.split[1], cols);
      for (size_t c = cb * (size_t)pi.split[1]; c < c1; ++c)
        if (wd[r * cols + c] % t) {
          row[cb] = 0;
          break;
        }
    }
  }
  DI_TRY(f->sync());
  encoded_share.assign(pi.n_out, std::vector<seal::Plaintext>(pi.n_in));
  for (size_t j = 0; j < pi.n_out; ++j)
    for (size_t i = 0; i < pi.n_in; ++i) encoded_share[j][i].facade_set_lazy(batch, j * pi.n_in + i);
  return Code::OK;
}

// :394-471 + addRandomMask :473-520 + removeUnusedCoeffs :525-561
Code HomFCSS::matVecMul(const std::vector<std::vector<seal::Plaintext>> &matrix,
                        const std::vector<seal::Ciphertext> &vec_share0, const std::vector<seal::Plaintext> &vec_share1,
                        const Meta &meta, std::vector<seal::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1,
                        size_t) const {
  FcFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && evaluator_ && f && f->pk_last, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  PlanInfo pi;
  ENSURE_OR_RETURN(plan_for(*f, meta, pi) == Code::OK, Code::ERR_INVALID_ARG);
  ENSURE_OR_RETURN(vec_share0.size() == pi.n_in, Code::ERR_INVALID_ARG);
  ENSURE_OR_RETURN(matrix.size() == pi.n_out, Code::ERR_INVALID_ARG);
  for (const auto &c : matrix) ENSURE_OR_RETURN(c.size() == pi.n_in, Code::ERR_INVALID_ARG);
  for (const auto &row : matrix)
    for (const auto &sub : row) {
      if (sub.is_zero()) {  // :414-421
        LOG(WARNING) << "matVecMul: sub-matrix with all zero is not supported";
        return Code::ERR_INVALID_ARG;
      }
    }
  if (meta.is_shared_input) ENSURE_OR_RETURN(vec_share1.size() == pi.n_in, Code::ERR_DIM_MISMATCH);
  const size_t N = f->N, L = f->L, rows = (size_t)meta.weight_shape.rows();
  const auto tag = tag_of(meta);
  const PlainBatch *wb = whole_batch(matrix[0].data(), pi.n_in, kKindFcWeights, tag);
  for (size_t j = 1; wb && j < pi.n_out; ++j)
    if (whole_batch(matrix[j].data(), pi.n_in, kKindFcWeights, tag, j * pi.n_in) != wb) wb = nullptr;
  const PlainBatch *ib = meta.is_shared_input ? whole_batch(vec_share1.data(), pi.n_in, kKindFcInput, tag) : nullptr;
  if (wb && wb->fast.get() != f) wb = nullptr;
  if (ib && ib->fast.get() != f) ib = nullptr;
  Dev cts;
  {
    const Code c = upload_cts(*f, vec_share0, L, cts);
    if (c != Code::OK) return c == Code::ERR_INVALID_ARG ? Code::ERR_INTERNAL : c;
  }
  Dev spts, wpts;
  const u64 *d_spts = nullptr, *d_wpts = nullptr;
  if (meta.is_shared_input) {
    if (ib) {
      d_spts = ib->pts.p;
    } else {
      spts = f->mk(pi.n_in * N);
      ENSURE_OR_RETURN(spts, Code::ERR_SEAL_MEMORY);
      DI_TRY(mmh_memset(spts.p, 0, pi.n_in * N * 8, f->stream));
      for (size_t i = 0; i < pi.n_in; ++i) {
        ENSURE_OR_RETURN(vec_share1[i].coeff_count() <= N && !vec_share1[i].is_ntt_form(), Code::ERR_INTERNAL);
        DI_TRY(f->up(spts, vec_share1[i].data(), vec_share1[i].coeff_count(), i * N));
      }
      d_spts = spts.p;
    }
  }
  if (wb) {
    d_wpts = wb->pts.p;
  } else {
    wpts = f->mk(pi.n_out * pi.n_in * N);
    ENSURE_OR_RETURN(wpts, Code::ERR_SEAL_MEMORY);
    DI_TRY(mmh_memset(wpts.p, 0, pi.n_out * pi.n_in * N * 8, f->stream));
    for (size_t j = 0; j < pi.n_out; ++j)
      for (size_t i = 0; i < pi.n_in; ++i) {
        const auto &pt = matrix[j][i];
        ENSURE_OR_RETURN(pt.coeff_count() <= N && !pt.is_ntt_form(), Code::ERR_INTERNAL);
        DI_TRY(f->up(wpts, pt.data(), pt.coeff_count(), (j * pi.n_in + i) * N));
      }
    d_wpts = wpts.p;
  }
  Dev u = f->mk(pi.n_out * N), e1 = f->mk(pi.n_out * N), mask = f->mk(pi.n_out * N), out_cts = f->mk(pi.n_out * 2 * N);
  Dev share = f->mk(rows);
  ENSURE_OR_RETURN(u && e1 && mask && out_cts && share, Code::ERR_SEAL_MEMORY);
  std::uint32_t key[8];
#if USE_APPROX_RESHARE
  const std::uint64_t sid = f->streams(key, 3);
  DI_TRY(mmh_sample_reshare(f->ctx, key, sid, pi.n_out, (std::int64_t *)u.p, (std::int64_t *)e1.p, mask.p, f->stream));
  DI_TRY(mmh_fc_server(pi.plan, cts.p, d_spts, d_wpts, f->pk_last.p, (const std::int64_t *)u.p, (const std::int64_t *)e1.p,
                       mask.p, out_cts.p, share.p, f->stream));
#else
  Dev flood = f->mk(pi.n_out * N), e0 = f->mk(pi.n_out * N);
  ENSURE_OR_RETURN(flood && e0, Code::ERR_SEAL_MEMORY);
  const std::uint64_t sid = f->streams(key, 5);
  DI_TRY(mmh_sample_ternary(key, sid, (std::int64_t *)u.p, pi.n_out * N, f->stream));
  DI_TRY(mmh_sample_noise(key, sid + 1, (std::int64_t *)e1.p, pi.n_out * N, f->stream));
  DI_TRY(mmh_chacha20_words(key, sid + 2, 0, mask.p, pi.n_out * N, f->stream));
  DI_TRY(mmh_chacha20_words(key, sid + 3, 0, flood.p, pi.n_out * N, f->stream));
  DI_TRY(mmh_sample_noise(key, sid + 4, (std::int64_t *)e0.p, pi.n_out * N, f->stream));
  DI_TRY(mmh_fc_server_exact(pi.plan, cts.p, d_spts, d_wpts, f->pk_last.p, flood.p, (const std::int64_t *)u.p,
                             (const std::int64_t *)e0.p, (const std::int64_t *)e1.p, mask.p, out_cts.p, share.p, f->stream));
#endif
  const TensorShape oshape({(int64_t)rows});
  out_share1.Reshape(oshape);  // (the reference reshapes once here too; its in-worker Reshape race is not reproduced)
  DI_TRY(f->down(out_share1.data(), share, rows));
  return download_last_level(*f, *context_, out_cts, pi.n_out, out_share0);
}

// :572-602
Code HomFCSS::decryptToVector(const std::vector<seal::Ciphertext> &enc_vector, const Meta &meta, Tensor<uint64_t> &out,
                              size_t) const {
  FcFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && sk_ && f && f->sk_last, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  PlanInfo pi;
  ENSURE_OR_RETURN(plan_for(*f, meta, pi) == Code::OK, Code::ERR_INVALID_ARG);
  ENSURE_OR_RETURN(enc_vector.size() == pi.n_out, Code::ERR_INVALID_ARG);
  for (const auto &c : enc_vector) ENSURE_OR_RETURN(c.parms_id() == context_->last_parms_id(), Code::ERR_INVALID_ARG);
  const size_t rows = (size_t)meta.weight_shape.rows();
  Dev cts;
  CHECK_ERR(upload_cts(*f, enc_vector, 1, cts), "decryptToVector");
  Dev res = f->mk(rows);
  ENSURE_OR_RETURN(res, Code::ERR_SEAL_MEMORY);
  DI_TRY(mmh_fc_decrypt(pi.plan, f->sk_last.p, cts.p, res.p, f->stream));
  out.Reshape(TensorShape({(int64_t)rows}));
  DI_TRY(f->down(out.data(), res, rows));
  DI_TRY(f->sync());
  return Code::OK;
}

// :604-610
Code HomFCSS::idealFunctionality(const Tensor<uint64_t> &w, const Tensor<uint64_t> &vec, const Meta &meta,
                                 Tensor<uint64_t> &out) const {
  ENSURE_OR_RETURN(w.shape().IsSameSize(meta.weight_shape) && vec.shape().IsSameSize(meta.input_shape),
                   Code::ERR_DIM_MISMATCH);
  ENSURE_OR_RETURN(w.cols() == vec.length(), Code::ERR_DIM_MISMATCH);
  const uint64_t t = plain_modulus();
  ENSURE_OR_RETURN(t != (uint64_t)-1 && t > 1, Code::ERR_INTERNAL);
  out.Reshape(TensorShape({w.rows()}));
  for (int64_t r = 0; r < w.rows(); ++r) {
    unsigned __int128 sum = 0;
    for (int64_t c = 0; c < w.cols(); ++c) sum += (uint64_t)((unsigned __int128)(w(r, c) % t) * (vec(c) % t) % t);
    out(r) = (uint64_t)(sum % t);
  }
  return Code::OK;
}

}  // namespace gemini
