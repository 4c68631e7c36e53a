// gemini::HomConv2DSS (include/gemini/cheetah/hom_conv2d_ss.cc) over libmm_hip: every public method is the
// reference's argument checks followed by one layer call of the C ABI between an upload and a download.
#include "mm_gemini/hom_conv2d_ss.h"

#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "host_util.h"
#include "mm_hip.h"

namespace mm {
namespace he {
namespace {
void must(int rc, const char *what) {
  if (rc != MMH_OK) throw std::runtime_error(std::string(what) + ": " + mmh_last_error());
}
}  // namespace

Context::Context(size_t poly_degree, const std::vector<int> &bits, uint64_t plain_modulus, int device)
    : N_(poly_degree), L_(bits.size()), t_(plain_modulus), q_(bits.size()) {
  must(mmh_ctx_create_bits(&ctx_, poly_degree, bits.data(), bits.size(), plain_modulus, device), "SEALContext");
  must(mmh_ctx_info(ctx_, nullptr, nullptr, q_.data(), nullptr, nullptr, nullptr), "ctx_info");
}
Context::~Context() { mmh_ctx_destroy(ctx_); }
}  // namespace he
}  // namespace mm

namespace gemini {
namespace {
mmh_conv_meta to_abi(const HomConv2DSS::Meta &meta) {
  return mmh_conv_meta{meta.ishape.channels(), meta.ishape.height(),  meta.ishape.width(),
                       meta.fshape.height(),   meta.fshape.width(),   (int64_t)meta.n_filters,
                       meta.padding == Padding::SAME, (int32_t)meta.stride, meta.is_shared_input};
}
struct PlanInfo {
  size_t n_in = 0, n_cs = 0, n_sp = 0, n_out = 0;
  int64_t oshape[3] = {0, 0, 0};
};
int plan_info(const mmh_conv_plan *p, PlanInfo &i) {
  return mmh_conv_plan_info(p, &i.n_in, &i.n_cs, &i.n_sp, &i.n_out, i.oshape, nullptr);
}
}  // namespace

}  // namespace gemini

// KeyGenerator + create_public_key
void mm::he::GenerateKeys(const Context &context, SecretKey &sk, PublicKey &pk) {
  using gemini::Dev;
  const size_t N = context.poly_degree(), L = context.coeff_modulus_size();
  uint32_t key[8];
  if (!gemini::seed_from_os(key)) throw std::runtime_error("GenerateKeys: cannot seed the sampler");
  Dev s(N), a(L * N), e(N), d_sk(L * N), d_pk(2 * L * N);
  must(gemini::first_rc({s.rc, a.rc, e.rc, d_sk.rc, d_pk.rc}), "mmh_malloc");
  must(mmh_sample_ternary(key, 1, (int64_t *)s.p, N, nullptr), "sample_ternary");
  must(mmh_sample_uniform(context.raw(), 0, key, 2, a.p, 1, nullptr), "sample_uniform");
  must(mmh_sample_noise(key, 3, (int64_t *)e.p, N, nullptr), "sample_noise");
  must(mmh_signed_to_ntt(context.raw(), 0, (const int64_t *)s.p, 1, d_sk.p, nullptr), "keygen");
  must(mmh_make_public_key(context.raw(), d_sk.p, a.p, (const int64_t *)e.p, d_pk.p, nullptr), "create_public_key");
  sk.data.resize(L * N), pk.data.resize(2 * L * N);
  must(d_sk.down(sk.data.data(), L * N), "d2h");
  must(d_pk.down(pk.data.data(), 2 * L * N), "d2h");
}

namespace gemini {

HomConv2DSS::~HomConv2DSS() {
  for (auto &kv : plans_) mmh_conv_plan_destroy(kv.second);
  mmh_free(d_sk_), mmh_free(d_sk_last_), mmh_free(d_pk_last_);
}

// hom_conv2d_ss.cc:467-495: keys must belong to this context (here: have its size)
Code HomConv2DSS::setUp(const mm::he::Context &context, std::optional<mm::he::SecretKey> sk,
                        std::shared_ptr<mm::he::PublicKey> pk) {
  MM_ENSURE(context.raw() != nullptr, Code::ERR_NULL_POINTER);
  const size_t N = context.poly_degree(), L = context.coeff_modulus_size();
  if (sk) MM_ENSURE(sk->data.size() == L * N, Code::ERR_INVALID_ARG);
  if (pk) MM_ENSURE(pk->data.size() == 2 * L * N, Code::ERR_INVALID_ARG);
  MM_ENSURE(seed_from_os(prng_key_), Code::ERR_INTERNAL);
  context_ = &context;
  mmh_free(d_sk_), mmh_free(d_sk_last_), mmh_free(d_pk_last_);
  d_sk_ = d_sk_last_ = d_pk_last_ = nullptr;
  auto put = [&](uint64_t **dst, const uint64_t *src, size_t words) {
    void *q = nullptr;
    int rc = mmh_malloc(&q, words * 8);
    if (!rc) rc = mmh_memcpy_h2d(q, src, words * 8, nullptr);
    if (!rc) rc = mmh_stream_sync(nullptr);
    *dst = static_cast<uint64_t *>(q);
    return rc;
  };
  if (sk) {
    MM_TRY(put(&d_sk_, sk->data.data(), L * N));
    MM_TRY(put(&d_sk_last_, sk->data.data(), N));  // the last level keeps limb 0
  }
  if (pk) {
    std::vector<uint64_t> last(2 * N);
    std::copy(pk->data.begin(), pk->data.begin() + N, last.begin());
    std::copy(pk->data.begin() + L * N, pk->data.begin() + L * N + N, last.begin() + N);
    MM_TRY(put(&d_pk_last_, last.data(), 2 * N));
  }
  sk_ = std::move(sk);
  pk_ = std::move(pk);
  return Code::OK;
}

size_t HomConv2DSS::poly_degree() const { return context_ ? context_->poly_degree() : 0; }       // :451-456
uint64_t HomConv2DSS::plain_modulus() const { return context_ ? context_->plain_modulus() : -1; }  // :458-465

mmh_conv_plan *HomConv2DSS::plan(const Meta &meta) const {
  const mmh_conv_meta m = to_abi(meta);
  const std::vector<int64_t> key{m.in_c, m.in_h, m.in_w, m.f_h, m.f_w, m.n_filters, m.padding_same, m.stride,
                                 m.is_shared_input};
  std::lock_guard<std::mutex> g(mu_);
  auto it = plans_.find(key);
  if (it != plans_.end()) return it->second;
  mmh_conv_plan *p = nullptr;
  if (mmh_conv_plan_create(context_->raw(), &m, &p) != MMH_OK) return nullptr;
  return plans_[key] = p;
}

// :497-522
Code HomConv2DSS::encryptImage(const Tensor<uint64_t> &img, const Meta &meta,
                               std::vector<mm::he::Ciphertext> &encrypted_share, size_t) const {
  MM_ENSURE(context_ && d_sk_, Code::ERR_CONFIG);
  MM_ENSURE(img.shape().IsSameSize(meta.ishape), Code::ERR_DIM_MISMATCH);
  mmh_conv_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INTERNAL);
  PlanInfo pi;
  MM_TRY(plan_info(p, pi));
  const size_t N = poly_degree(), L = context_->coeff_modulus_size(), ct = 2 * L * N;
  Dev d_img((size_t)img.NumElements()), a(pi.n_in * L * N), e(pi.n_in * N), cts(pi.n_in * ct);
  MM_TRY(first_rc({d_img.rc, a.rc, e.rc, cts.rc}));
  MM_TRY(d_img.up(img.data(), (size_t)img.NumElements()));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 2;
  }
  MM_TRY(mmh_sample_uniform(context_->raw(), 0, prng_key_, sid, a.p, pi.n_in, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e.p, pi.n_in * N, nullptr));
  MM_TRY(mmh_conv_encrypt_image(p, d_sk_, d_img.p, a.p, (const int64_t *)e.p, cts.p, nullptr));
  encrypted_share.resize(pi.n_in);
  for (size_t i = 0; i < pi.n_in; ++i) {
    encrypted_share[i].resize(2, L, N);
    MM_TRY(cts.down(encrypted_share[i].data(), ct, i * ct));
  }
  return Code::OK;
}

// :524-535
Code HomConv2DSS::encodeImage(const Tensor<uint64_t> &img, const Meta &meta,
                              std::vector<mm::he::Plaintext> &encoded_share, size_t) const {
  MM_ENSURE(context_, Code::ERR_CONFIG);
  MM_ENSURE(img.shape().IsSameSize(meta.ishape), Code::ERR_DIM_MISMATCH);
  mmh_conv_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INTERNAL);
  PlanInfo pi;
  MM_TRY(plan_info(p, pi));
  const size_t N = poly_degree();
  Dev d_img((size_t)img.NumElements()), pts(pi.n_in * N);
  MM_TRY(first_rc({d_img.rc, pts.rc}));
  MM_TRY(d_img.up(img.data(), (size_t)img.NumElements()));
  MM_TRY(mmh_conv_encode_image(p, d_img.p, pts.p, nullptr));
  encoded_share.resize(pi.n_in);
  for (size_t i = 0; i < pi.n_in; ++i) {
    encoded_share[i].coeffs.resize(N);
    MM_TRY(pts.down(encoded_share[i].data(), N, i * N));
  }
  return Code::OK;
}

// :537-561
Code HomConv2DSS::encodeFilters(const std::vector<Tensor<uint64_t>> &filters, const Meta &meta,
                                std::vector<std::vector<mm::he::Plaintext>> &encoded_filters, size_t) const {
  MM_ENSURE(context_, Code::ERR_CONFIG);
  const size_t M = filters.size();
  MM_ENSURE(M > 0 && M == meta.n_filters, Code::ERR_INVALID_ARG);
  for (const auto &f : filters) MM_ENSURE(f.shape().IsSameSize(meta.fshape), Code::ERR_DIM_MISMATCH);
  mmh_conv_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INTERNAL);
  PlanInfo pi;
  MM_TRY(plan_info(p, pi));
  const size_t N = poly_degree(), fw = (size_t)meta.fshape.num_elements();
  Dev d_f(M * fw), pts(M * pi.n_cs * N);
  MM_TRY(first_rc({d_f.rc, pts.rc}));
  for (size_t m = 0; m < M; ++m) MM_TRY(d_f.up(filters[m].data(), fw, m * fw));
  MM_TRY(mmh_conv_encode_filters(p, d_f.p, pts.p, nullptr));
  encoded_filters.assign(M, std::vector<mm::he::Plaintext>(pi.n_cs));
  for (size_t m = 0; m < M; ++m)
    for (size_t c = 0; c < pi.n_cs; ++c) {
      encoded_filters[m][c].coeffs.resize(N);
      MM_TRY(pts.down(encoded_filters[m][c].data(), N, (m * pi.n_cs + c) * N));
    }
  return Code::OK;
}

// :621-712 (shipped USE_APPROX_RESHARE build)
Code HomConv2DSS::conv2DSS(const std::vector<mm::he::Ciphertext> &img_share0,
                           const std::vector<mm::he::Plaintext> &img_share1,
                           const std::vector<std::vector<mm::he::Plaintext>> &filters, const Meta &meta,
                           std::vector<mm::he::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1, size_t) const {
  MM_ENSURE(context_ && d_pk_last_, Code::ERR_CONFIG);
  MM_ENSURE(filters.size() == meta.n_filters, Code::ERR_DIM_MISMATCH);
  if (meta.is_shared_input) MM_ENSURE(img_share0.size() == img_share1.size(), Code::ERR_DIM_MISMATCH);
  mmh_conv_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_CONFIG);  // "empty out_shape"
  PlanInfo pi;
  MM_TRY(plan_info(p, pi));
  MM_ENSURE(img_share0.size() == pi.n_in, Code::ERR_DIM_MISMATCH);
  const size_t N = poly_degree(), L = context_->coeff_modulus_size(), ct = 2 * L * N, M = meta.n_filters;
  for (const auto &c : img_share0)
    MM_ENSURE(c.size() == 2 && c.coeff_modulus_size() == L && c.poly_modulus_degree() == N, Code::ERR_INTERNAL);
  // conv2DOneFilter :573-580, :685: a filter whose plaintexts are all zero is an error
  for (const auto &f : filters) {
    MM_ENSURE(f.size() == pi.n_cs, Code::ERR_INTERNAL);
    bool nz = false;
    for (const auto &pt : f) {
      MM_ENSURE(pt.coeff_count() == N, Code::ERR_INTERNAL);
      nz = nz || !pt.is_zero();
    }
    MM_ENSURE(nz, Code::ERR_INTERNAL);
  }
  Dev cts(pi.n_in * ct), share(meta.is_shared_input ? pi.n_in * N : 1), fpts(M * pi.n_cs * N);
  Dev u(pi.n_out * N), e1(pi.n_out * N), mask(pi.n_out * N), out_cts(pi.n_out * 2 * N);
  const size_t out_words = (size_t)(pi.oshape[0] * pi.oshape[1] * pi.oshape[2]);
  Dev out_share(out_words);
  MM_TRY(first_rc({cts.rc, share.rc, fpts.rc, u.rc, e1.rc, mask.rc, out_cts.rc, out_share.rc}));
  for (size_t i = 0; i < pi.n_in; ++i) MM_TRY(cts.up(img_share0[i].data(), ct, i * ct));
  if (meta.is_shared_input)
    for (size_t i = 0; i < pi.n_in; ++i) {
      MM_ENSURE(img_share1[i].coeff_count() == N, Code::ERR_INTERNAL);
      MM_TRY(share.up(img_share1[i].data(), N, i * N));
    }
  for (size_t m = 0; m < M; ++m)
    for (size_t c = 0; c < pi.n_cs; ++c) MM_TRY(fpts.up(filters[m][c].data(), N, (m * pi.n_cs + c) * N));
  uint64_t sid;
  {
    std::lock_guard<std::mutex> g(mu_);
    sid = stream_id_, stream_id_ += 3;
  }
  MM_TRY(mmh_sample_ternary(prng_key_, sid, (int64_t *)u.p, pi.n_out * N, nullptr));
  MM_TRY(mmh_sample_noise(prng_key_, sid + 1, (int64_t *)e1.p, pi.n_out * N, nullptr));
  MM_TRY(mmh_sample_uniform(context_->raw(), 1, prng_key_, sid + 2, mask.p, pi.n_out, nullptr));
  MM_TRY(mmh_conv_server(p, cts.p, meta.is_shared_input ? share.p : nullptr, fpts.p, d_pk_last_, (const int64_t *)u.p,
                         (const int64_t *)e1.p, mask.p, out_cts.p, out_share.p, nullptr));
  out_share0.resize(pi.n_out);
  for (size_t k = 0; k < pi.n_out; ++k) {
    out_share0[k].resize(2, 1, N);
    MM_TRY(out_cts.down(out_share0[k].data(), 2 * N, k * 2 * N));
  }
  const TensorShape oshape({pi.oshape[0], pi.oshape[1], pi.oshape[2]});
  if (!out_share1.shape().IsSameSize(oshape)) out_share1.Reshape(oshape);
  MM_TRY(out_share.down(out_share1.data(), out_words));
  return Code::OK;
}

// :853-941
Code HomConv2DSS::decryptToTensor(const std::vector<mm::he::Ciphertext> &enc_tensor, const Meta &meta,
                                  Tensor<uint64_t> &out, size_t) const {
  MM_ENSURE(context_ && d_sk_last_, Code::ERR_CONFIG);
  mmh_conv_plan *p = plan(meta);
  MM_ENSURE(p != nullptr, Code::ERR_INVALID_ARG);
  PlanInfo pi;
  MM_TRY(plan_info(p, pi));
  MM_ENSURE(enc_tensor.size() == pi.n_out, Code::ERR_INTERNAL);
  const size_t N = poly_degree();
  Dev cts(pi.n_out * 2 * N);
  const size_t out_words = (size_t)(pi.oshape[0] * pi.oshape[1] * pi.oshape[2]);
  Dev res(out_words);
  MM_TRY(first_rc({cts.rc, res.rc}));
  for (size_t k = 0; k < pi.n_out; ++k) {
    MM_ENSURE(enc_tensor[k].size() == 2 && enc_tensor[k].coeff_modulus_size() == 1 &&
                  enc_tensor[k].poly_modulus_degree() == N,
              Code::ERR_DIM_MISMATCH);
    MM_TRY(cts.up(enc_tensor[k].data(), 2 * N, k * 2 * N));
  }
  MM_TRY(mmh_conv_decrypt(p, d_sk_last_, cts.p, res.p, nullptr));
  const TensorShape oshape({pi.oshape[0], pi.oshape[1], pi.oshape[2]});
  if (!out.shape().IsSameSize(oshape)) out.Reshape(oshape);
  MM_TRY(res.down(out.data(), out_words));
  return Code::OK;
}

// :975-1012
Code HomConv2DSS::idealFunctionality(const Tensor<uint64_t> &in_tensor, const std::vector<Tensor<uint64_t>> &filters,
                                     const Meta &meta, Tensor<uint64_t> &out_tensor) const {
  MM_ENSURE(meta.ishape.IsSameSize(in_tensor.shape()), Code::ERR_DIM_MISMATCH);
  MM_ENSURE(meta.n_filters == filters.size(), Code::ERR_DIM_MISMATCH);
  for (const auto &f : filters) MM_ENSURE(meta.fshape.IsSameSize(f.shape()), Code::ERR_DIM_MISMATCH);
  const long stride = (long)meta.stride, C = meta.ishape.channels(), H = meta.ishape.height(), W = meta.ishape.width();
  const long fh = meta.fshape.height(), fw = meta.fshape.width();
  const long ph = meta.padding == Padding::SAME ? fh - 1 : 0, pw = meta.padding == Padding::SAME ? fw - 1 : 0;
  MM_ENSURE(stride >= 1 && H + ph >= fh && W + pw >= fw, Code::ERR_DIM_MISMATCH);
  const long OH = (H + ph - fh + stride) / stride, OW = (W + pw - fw + stride) / stride;
  out_tensor.Reshape(TensorShape({(int64_t)meta.n_filters, OH, OW}));
  const uint64_t base_mod = plain_modulus();
  MM_ENSURE(base_mod != (uint64_t)-1, Code::ERR_INTERNAL);
  const long padh = (ph + 1) / 2, padw = (pw + 1) / 2;
  for (size_t m = 0; m < meta.n_filters; ++m)
    for (long oh = 0, ih = -padh; oh < OH; ++oh, ih += stride)
      for (long ow = 0, iw = -padw; ow < OW; ++ow, iw += stride) {
        unsigned __int128 sum = 0;
        uint64_t wrap = 0;
        for (long c = 0; c < C; ++c)
          for (long r = 0; r < fh; ++r)
            for (long q = 0; q < fw; ++q) {
              const long y = ih + r, x = iw + q;
              if (y < 0 || y >= H || x < 0 || x >= W) continue;
              if (base_mod > 1)
                sum += (uint64_t)((unsigned __int128)(filters[m](c, r, q) % base_mod) * (in_tensor(c, y, x) % base_mod) %
                                  base_mod);
              else
                wrap += filters[m](c, r, q) * in_tensor(c, y, x);  // mod 2^64
            }
        out_tensor((int64_t)m, oh, ow) = base_mod > 1 ? (uint64_t)(sum % base_mod) : wrap;
      }
  return Code::OK;
}

}  // namespace gemini
