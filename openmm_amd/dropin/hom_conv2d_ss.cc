// gemini::HomConv2DSS -- the member functions the REFERENCE'S header declares
// (include/gemini/cheetah/hom_conv2d_ss.h:27-115, compiled from /root/reference, unchanged), implemented on the
// seal:: facade types with one libmm_hip layer call per method.  Linked in place of the reference's
// include/gemini/cheetah/hom_conv2d_ss.cc (which reaches SEAL once per ciphertext from a thread pool); the argument
// checks, return codes and the order of effects follow that file, cited per method.
//
// Build variant: USE_APPROX_RESHARE as the reference's CMake sets it (scripts/build.sh:22 ships ON).
#include "gemini/cheetah/hom_conv2d_ss.h"

#include <algorithm>

#include "dropin_common.h"
#include "gemini/core/common.h"
#include "gemini/core/logging.h"

#ifndef USE_APPROX_RESHARE
#define USE_APPROX_RESHARE 0  // hom_conv2d_ss.cc:9-13: the macro decides, default off
#endif

namespace gemini {
using namespace mm_dropin;

namespace {
struct ConvFast : Fast {
  std::map<std::array<std::int64_t, 10>, mmh_conv_plan *> plans;
  ~ConvFast() {
    if (stream) (void)mmh_stream_sync(stream);
    for (auto &kv : plans) (void)mmh_conv_plan_destroy(kv.second);
  }
};
struct PlanInfo {
  mmh_conv_plan *plan = nullptr;
  std::size_t n_in = 0, n_cs = 0, n_sp = 0, n_out = 0;
  std::int64_t oshape[3] = {0, 0, 0}, strides[3] = {0, 0, 0};
};
std::array<std::int64_t, 10> tag_of(const HomConv2DSS::Meta &m) {
  return {m.ishape.channels(), m.ishape.height(), m.ishape.width(), m.fshape.height(), m.fshape.width(),
          (std::int64_t)m.n_filters, m.padding == Padding::SAME, (std::int64_t)m.stride, m.is_shared_input, 0};
}
Code plan_for(ConvFast &f, const HomConv2DSS::Meta &meta, PlanInfo &pi) {
  DI_ENSURE(meta.ishape.dims() == 3 && meta.fshape.dims() == 3 && meta.ishape.channels() == meta.fshape.channels(),
            Code::ERR_INVALID_ARG);
  const auto tag = tag_of(meta);
  auto it = f.plans.find(tag);
  if (it == f.plans.end()) {
    mmh_conv_meta m{tag[0], tag[1], tag[2], tag[3], tag[4], tag[5], (std::int32_t)tag[6], (std::int32_t)tag[7],
                    (std::int32_t)tag[8]};
    mmh_conv_plan *p = nullptr;
    DI_TRY(mmh_conv_plan_create(f.ctx, &m, &p));
    it = f.plans.emplace(tag, p).first;
  }
  pi.plan = it->second;
  DI_TRY(mmh_conv_plan_info(pi.plan, &pi.n_in, &pi.n_cs, &pi.n_sp, &pi.n_out, pi.oshape, pi.strides));
  return Code::OK;
}
ConvFast *fast_of(const std::shared_ptr<seal::Evaluator> &ev) {
  return ev ? static_cast<ConvFast *>(ev->facade_ext().get()) : nullptr;
}
std::shared_ptr<Fast> fast_ptr(const std::shared_ptr<seal::Evaluator> &ev) {
  return std::static_pointer_cast<Fast>(ev->facade_ext());
}
}  // namespace

// hom_conv2d_ss.cc:443-465
seal::scheme_type HomConv2DSS::scheme() const {
  return context_ ? context_->first_context_data()->parms().scheme() : seal::scheme_type::none;
}
size_t HomConv2DSS::poly_degree() const {
  return context_ ? context_->first_context_data()->parms().poly_modulus_degree() : 0;
}
uint64_t HomConv2DSS::plain_modulus() const {
  return context_ ? context_->first_context_data()->parms().plain_modulus().value() : -1;
}

// :467-495.  The device state (stream, key copies, layer plans) hangs on the evaluator this object owns; the tensor
// encoder of the reference is not needed: encoding happens inside the layer calls.
Code HomConv2DSS::setUp(const seal::SEALContext &context, std::optional<seal::SecretKey> sk,
                        std::shared_ptr<seal::PublicKey> pk) {
  context_ = std::make_shared<seal::SEALContext>(context);
  ENSURE_OR_RETURN(context_, Code::ERR_NULL_POINTER);
  if (sk) {
    if (!seal::is_metadata_valid_for(*sk, *context_)) {
      LOG(WARNING) << "HomConv2DSS: invalid secret key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    sk_ = seal::SecretKey(*sk);
    encryptor_ = std::make_shared<seal::Encryptor>(*context_, *sk);
  }
  if (pk) {
    if (!seal::is_metadata_valid_for(*pk, *context_)) {
      LOG(WARNING) << "HomConv2DSS: invalid public key for this SEALContext";
      return Code::ERR_INVALID_ARG;
    }
    pk_ = std::make_shared<seal::PublicKey>(*pk);
  }
  evaluator_ = std::make_shared<seal::Evaluator>(*context_);
  auto f = std::make_shared<ConvFast>();
  CHECK_ERR(f->init(*context_, sk ? &*sk : nullptr, pk.get()), "HomConv2DSS::setUp");
  evaluator_->facade_ext() = f;
  return Code::OK;
}

// :497-522: EncodeImageShare + encrypt_symmetric per slice -> one mmh_conv_encrypt_image
Code HomConv2DSS::encryptImage(const Tensor<uint64_t> &img, const Meta &meta,
                               std::vector<seal::Serializable<seal::Ciphertext>> &encrypted_img, size_t nthreads) const {
  ConvFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && encryptor_ && f && f->sk, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(img.shape().IsSameSize(meta.ishape), Code::ERR_DIM_MISMATCH);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  Trace tr("encryptImage");
  PlanInfo pi;
  CHECK_ERR(plan_for(*f, meta, pi), "encryptImage");
  const size_t N = f->N, L = f->L, words = (size_t)img.NumElements();
  Dev d_img = f->mk(words), e = f->mk(pi.n_in * N), cts = f->mk(pi.n_in * 2 * L * N), a;
  ENSURE_OR_RETURN(d_img && e && cts, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(d_img, img.data(), words));
  std::vector<std::array<std::uint8_t, seal::wire::kSeedBytes>> seeds;
  tr.lap("plan + upload");
  CHECK_ERR(fresh_a(*f, *context_, pi.n_in, nthreads, a, seeds), "encryptImage");
  tr.lap("seed expansion");
  std::uint32_t key[8];
  const std::uint64_t sid = f->streams(key, 1);
  DI_TRY(mmh_sample_noise(key, sid, (std::int64_t *)e.p, pi.n_in * N, f->stream));
  DI_TRY(mmh_conv_encrypt_image(pi.plan, f->sk.p, d_img.p, a.p, (const std::int64_t *)e.p, cts.p, f->stream));
  const Code c = download_fresh(*f, *context_, cts, pi.n_in, seeds, encrypted_img);
  tr.lap("encrypt + download");
  return c;
}

// :524-535: the server's share stays on the device as the raw tensor; the plaintexts the reference builds here exist
// only if somebody reads them (facade::LazyPlain)
Code HomConv2DSS::encodeImage(const Tensor<uint64_t> &img, const Meta &meta, std::vector<seal::Plaintext> &encoded_img,
                              size_t) const {
  ConvFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && f, Code::ERR_CONFIG);
  ENSURE_OR_RETURN(img.shape().IsSameSize(meta.ishape), Code::ERR_DIM_MISMATCH);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  PlanInfo pi;
  CHECK_ERR(plan_for(*f, meta, pi), "HomConv2DSS::encryptImage: encode failed");
  const size_t words = (size_t)img.NumElements();
  auto batch = std::make_shared<PlainBatch>();
  batch->fast = fast_ptr(evaluator_);
  batch->kind = kKindConvImage, batch->tag = tag_of(meta), batch->count = pi.n_in, batch->N = f->N;
  batch->raw = f->mk(words);
  ENSURE_OR_RETURN(batch->raw, Code::ERR_SEAL_MEMORY);
  DI_TRY(f->up(batch->raw, img.data(), words));
  DI_TRY(f->sync());  // the caller's tensor may go away
  mmh_conv_plan *plan = pi.plan;
  batch->encode = [plan](const PlainBatch &b) {
    b.pts = b.fast->mk(b.count * b.N);
    if (!b.pts) return (int)MMH_ERR_MEMORY;
    return mmh_conv_encode_image(plan, b.raw.p, b.pts.p, b.fast->stream);
  };
  encoded_img.assign(pi.n_in, seal::Plaintext());
  for (size_t i = 0; i < pi.n_in; ++i) encoded_img[i].facade_set_lazy(batch, i);
  return Code::OK;
}

// :537-561: likewise for the filters: M x Cs sparse plaintexts in the reference, one [M][C][fh][fw] tensor here
Code HomConv2DSS::encodeFilters(const std::vector<Tensor<uint64_t>> &filters, const Meta &meta,
                                std::vector<std::vector<seal::Plaintext>> &encoded_filters, size_t) const {
  const size_t M = filters.size();
  ENSURE_OR_RETURN(M > 0 && M == meta.n_filters, Code::ERR_INVALID_ARG);
  for (const auto &flt : filters) ENSURE_OR_RETURN(flt.shape().IsSameSize(meta.fshape), Code::ERR_DIM_MISMATCH);
  ConvFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && f, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  Trace tr("encodeFilters");
  PlanInfo pi;
  CHECK_ERR(plan_for(*f, meta, pi), "EncodeFilter");
  const size_t fw = (size_t)meta.fshape.num_elements(), per_ch = (size_t)(meta.fshape.height() * meta.fshape.width());
  auto batch = std::make_shared<PlainBatch>();
  batch->fast = fast_ptr(evaluator_);
  batch->kind = kKindConvFilters, batch->tag = tag_of(meta), batch->count = M * pi.n_cs, batch->N = f->N;
  batch->raw = f->mk(M * fw);
  ENSURE_OR_RETURN(batch->raw, Code::ERR_SEAL_MEMORY);
  u64 *host = f->stage(M * fw);
  ENSURE_OR_RETURN(host, Code::ERR_SEAL_MEMORY);
  // a slice's plaintext is zero exactly when every tap of its channels is zero mod t (EncodeFilter writes one
  // coefficient per tap, tensor_encoder.cc:124-176): what conv2DOneFilter's nnz test (:573-580) asks
  const u64 t = plain_modulus(), wc = (u64)pi.strides[0], C = (u64)meta.fshape.channels();
  batch->zero.assign(M * pi.n_cs, 1);
  for (size_t m = 0; m < M; ++m) {
    const u64 *src = filters[m].data();
    std::memcpy(host + m * fw, src, fw * 8);
    for (size_t cs = 0; cs < pi.n_cs; ++cs) {
      const size_t c0 = cs * wc, c1 = std::min<size_t>(C, c0 + wc);
      bool nz = false;
      for (size_t k = c0 * per_ch; k < c1 * per_ch && !nz; ++k) nz = (src[k] % t) != 0;
      batch->zero[m * pi.n_cs + cs] = nz ? 0 : 1;
    }
  }
  DI_TRY(f->up(batch->raw, host, M * fw));
  DI_TRY(f->sync());
  mmh_conv_plan *plan = pi.plan;
  batch->encode = [plan](const PlainBatch &b) {
    b.pts = b.fast->mk(b.count * b.N);
    if (!b.pts) return (int)MMH_ERR_MEMORY;
    return mmh_conv_encode_filters(plan, b.raw.p, b.pts.p, b.fast->stream);
  };
  tr.lap("upload + zero scan");
  encoded_filters.assign(M, std::vector<seal::Plaintext>(pi.n_cs));
  for (size_t m = 0; m < M; ++m)
    for (size_t c = 0; c < pi.n_cs; ++c) encoded_filters[m][c].facade_set_lazy(batch, m * pi.n_cs + c);
  tr.lap("plaintext objects");
  return Code::OK;
}

// :621-712: add_plain, conv2DOneFilter per filter, addRandomMask, truncate_for_decryption, removeUnusedCoeffs -> one
// layer call.  Plaintexts that came from encodeImage / encodeFilters of this object are consumed as the tensors they
// were made from; any others (a caller that built them itself) are uploaded and take the entry point that reads
// encoded plaintexts.  Same results either way.
Code HomConv2DSS::conv2DSS(const std::vector<seal::Ciphertext> &img_share0, const std::vector<seal::Plaintext> &img_share1,
                           const std::vector<std::vector<seal::Plaintext>> &filters, const Meta &meta,
                           std::vector<seal::Ciphertext> &out_share0, Tensor<uint64_t> &out_share1, size_t) const {
  if (filters.size() != meta.n_filters) {
    LOG(WARNING) << "conv2DSS: #filters " << filters.size() << " != " << meta.n_filters << "\n";
    return Code::ERR_DIM_MISMATCH;
  }
  if (meta.is_shared_input && img_share0.size() != img_share1.size()) {
    LOG(WARNING) << "conv2DSS: #shares " << img_share0.size() << " != " << img_share1.size() << "\n";
    return Code::ERR_DIM_MISMATCH;
  }
  ConvFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && f && f->pk_last, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  Trace tr("conv2DSS");
  PlanInfo pi;
  if (plan_for(*f, meta, pi) != Code::OK) {
    LOG(WARNING) << "conv2DSS: empty out_shape";
    return Code::ERR_CONFIG;
  }
  const size_t N = f->N, L = f->L, M = meta.n_filters;
  // conv2DOneFilter (:573-590) per filter: the slices must fit the image, a filter that is all zero is an error
  ENSURE_OR_RETURN(img_share0.size() == pi.n_in, Code::ERR_INTERNAL);
  for (const auto &flt : filters) {
    ENSURE_OR_RETURN(flt.size() == pi.n_cs, Code::ERR_INTERNAL);
    size_t nnz = 0;
    for (const auto &pt : flt) nnz += pt.is_zero() ? 0 : 1;
    if (nnz == 0) {
      LOG(WARNING) << "conv2DOneFilter: filter with all zero is not supported";
      return Code::ERR_INTERNAL;
    }
  }
  const auto tag = tag_of(meta);
  const PlainBatch *fb = M ? whole_batch(filters[0].data(), pi.n_cs, kKindConvFilters, tag) : nullptr;
  for (size_t m = 1; fb && m < M; ++m)
    if (whole_batch(filters[m].data(), pi.n_cs, kKindConvFilters, tag, m * pi.n_cs) != fb) fb = nullptr;
  const PlainBatch *ib = meta.is_shared_input ? whole_batch(img_share1.data(), pi.n_in, kKindConvImage, tag) : nullptr;
  if (fb && fb->fast.get() != f) fb = nullptr;
  if (ib && ib->fast.get() != f) ib = nullptr;

  Dev cts;
  {
    const Code c = upload_cts(*f, img_share0, L, cts);
    if (c != Code::OK) {
      LOG(WARNING) << "SEAL ERROR: encrypted is not valid for encryption parameters";
      return c == Code::ERR_INVALID_ARG ? Code::ERR_INTERNAL : c;  // add_plain / multiply_plain would throw (:659-662)
    }
  }
  tr.lap("checks + upload");
  const size_t out_words = (size_t)(pi.oshape[0] * pi.oshape[1] * pi.oshape[2]);
  Dev u = f->mk(pi.n_out * N), e1 = f->mk(pi.n_out * N), mask = f->mk(pi.n_out * N), out_cts = f->mk(pi.n_out * 2 * N);
  Dev share = f->mk(out_words);
  ENSURE_OR_RETURN(u && e1 && mask && out_cts && share, Code::ERR_SEAL_MEMORY);
  std::uint32_t key[8];
#if USE_APPROX_RESHARE
  const std::uint64_t sid = f->streams(key, 3);
  const bool raw_share = !meta.is_shared_input || ib;
  if (fb && raw_share) {
    // encodeImage + encodeFilters + conv2DSS from the raw tensors, the re-sharing's u, e1 and mask drawn inside the call
    // from streams sid .. sid + 2 (mmh_conv_server_tensors_sampled = mmh_sample_reshare + mmh_conv_server_tensors)
    DI_TRY(mmh_conv_server_tensors_sampled(pi.plan, cts.p, ib ? ib->raw.p : nullptr, fb->raw.p, f->pk_last.p, key, sid, 0,
                                           out_cts.p, share.p, nullptr, nullptr, f->stream));
  } else
#endif
  {
    // encoded plaintexts on the device: a batch's own encoding, or the caller's plaintexts uploaded
    Dev spts, fpts;
    const u64 *d_spts = nullptr, *d_fpts = nullptr;
    if (meta.is_shared_input) {
      if (ib) {
        if (!ib->pts) DI_TRY(ib->encode(*ib));
        d_spts = ib->pts.p;
      } else {
        spts = f->mk(pi.n_in * N);
        ENSURE_OR_RETURN(spts, Code::ERR_SEAL_MEMORY);
        DI_TRY(mmh_memset(spts.p, 0, pi.n_in * N * 8, f->stream));
        for (size_t i = 0; i < pi.n_in; ++i) {
          ENSURE_OR_RETURN(img_share1[i].coeff_count() <= N && !img_share1[i].is_ntt_form(), Code::ERR_INTERNAL);
          DI_TRY(f->up(spts, img_share1[i].data(), img_share1[i].coeff_count(), i * N));
        }
        d_spts = spts.p;
      }
    }
    if (fb) {
      if (!fb->pts) DI_TRY(fb->encode(*fb));
      d_fpts = fb->pts.p;
    } else {
      fpts = f->mk(M * pi.n_cs * N);
      ENSURE_OR_RETURN(fpts, Code::ERR_SEAL_MEMORY);
      DI_TRY(mmh_memset(fpts.p, 0, M * pi.n_cs * N * 8, f->stream));
      for (size_t m = 0; m < M; ++m)
        for (size_t c = 0; c < pi.n_cs; ++c) {
          const auto &pt = filters[m][c];
          ENSURE_OR_RETURN(pt.coeff_count() <= N && !pt.is_ntt_form(), Code::ERR_INTERNAL);
          DI_TRY(f->up(fpts, pt.data(), pt.coeff_count(), (m * pi.n_cs + c) * N));
        }
      d_fpts = fpts.p;
    }
#if USE_APPROX_RESHARE
    DI_TRY(mmh_sample_reshare(f->ctx, key, sid, pi.n_out, (std::int64_t *)u.p, (std::int64_t *)e1.p, mask.p, f->stream));
    DI_TRY(mmh_conv_server(pi.plan, cts.p, d_spts, d_fpts, f->pk_last.p, (const std::int64_t *)u.p,
                           (const std::int64_t *)e1.p, mask.p, out_cts.p, share.p, f->stream));
#else
    // USE_APPROX_RESHARE off: flood noise, e0 kept, mask mod t through sub_plain (:225-272, :323-328, :358-380)
    Dev flood = f->mk(pi.n_out * N), e0 = f->mk(pi.n_out * N);
    ENSURE_OR_RETURN(flood && e0, Code::ERR_SEAL_MEMORY);
    const std::uint64_t sid = f->streams(key, 5);
    DI_TRY(mmh_sample_ternary(key, sid, (std::int64_t *)u.p, pi.n_out * N, f->stream));
    DI_TRY(mmh_sample_noise(key, sid + 1, (std::int64_t *)e1.p, pi.n_out * N, f->stream));
    DI_TRY(mmh_chacha20_words(key, sid + 2, 0, mask.p, pi.n_out * N, f->stream));
    DI_TRY(mmh_chacha20_words(key, sid + 3, 0, flood.p, pi.n_out * N, f->stream));
    DI_TRY(mmh_sample_noise(key, sid + 4, (std::int64_t *)e0.p, pi.n_out * N, f->stream));
    DI_TRY(mmh_conv_server_exact(pi.plan, cts.p, d_spts, d_fpts, f->pk_last.p, flood.p, (const std::int64_t *)u.p,
                                 (const std::int64_t *)e0.p, (const std::int64_t *)e1.p, mask.p, out_cts.p, share.p,
                                 f->stream));
#endif
    DI_TRY(f->sync());  // spts / fpts return to the pool behind the call
  }
  const TensorShape oshape({pi.oshape[0], pi.oshape[1], pi.oshape[2]});
  out_share1.Reshape(oshape);  // :694
  DI_TRY(f->down(out_share1.data(), share, out_words));
  tr.lap("enqueue");
  const Code c = download_last_level(*f, *context_, out_cts, pi.n_out, out_share0);
  tr.lap("device + download");
  return c;
}

// :853-941: decrypt + gather (the ciphertexts arrive through unsafe_load: truncated, so "invalid" to SEAL)
Code HomConv2DSS::decryptToTensor(const std::vector<seal::Ciphertext> &enc_tensor, const Meta &meta,
                                  Tensor<uint64_t> &out, size_t) const {
  ConvFast *f = fast_of(evaluator_);
  ENSURE_OR_RETURN(context_ && sk_ && f && f->sk_last, Code::ERR_CONFIG);
  std::lock_guard<std::recursive_mutex> g(f->mu);
  Trace tr("decryptToTensor");
  PlanInfo pi;
  ENSURE_OR_RETURN(plan_for(*f, meta, pi) == Code::OK, Code::ERR_INVALID_ARG);
  ENSURE_OR_RETURN(enc_tensor.size() == pi.n_out, Code::ERR_INVALID_ARG);
  for (const auto &c : enc_tensor) ENSURE_OR_RETURN(c.parms_id() == context_->last_parms_id(), Code::ERR_INVALID_ARG);
  Dev cts;
  CHECK_ERR(upload_cts(*f, enc_tensor, 1, cts), "decryptToTensor");
  const size_t out_words = (size_t)(pi.oshape[0] * pi.oshape[1] * pi.oshape[2]);
  Dev res = f->mk(out_words);
  ENSURE_OR_RETURN(res, Code::ERR_SEAL_MEMORY);
  DI_TRY(mmh_conv_decrypt(pi.plan, f->sk_last.p, cts.p, res.p, f->stream));
  out.Reshape(TensorShape({pi.oshape[0], pi.oshape[1], pi.oshape[2]}));
  DI_TRY(f->down(out.data(), res, out_words));
  DI_TRY(f->sync());
  tr.lap("all");
  return Code::OK;
}

// :975-1012: the cleartext convolution mod t (Tensor::Conv2D, tensor.h:274-362: SAME pads f - 1 in total, the larger
// half first)
Code HomConv2DSS::idealFunctionality(const Tensor<uint64_t> &in_tensor, const std::vector<Tensor<uint64_t>> &filters,
                                     const Meta &meta, Tensor<uint64_t> &out_tensor) const {
  ENSURE_OR_RETURN(meta.ishape.IsSameSize(in_tensor.shape()), Code::ERR_DIM_MISMATCH);
  ENSURE_OR_RETURN(meta.n_filters == filters.size(), Code::ERR_DIM_MISMATCH);
  for (const auto &flt : filters) ENSURE_OR_RETURN(meta.fshape.IsSameSize(flt.shape()), Code::ERR_DIM_MISMATCH);
  const long stride = (long)meta.stride, C = meta.ishape.channels(), H = meta.ishape.height(), W = meta.ishape.width();
  const long fh = meta.fshape.height(), fw = meta.fshape.width();
  const long ph = meta.padding == Padding::SAME ? fh - 1 : 0, pw = meta.padding == Padding::SAME ? fw - 1 : 0;
  ENSURE_OR_RETURN(stride >= 1 && H + ph >= fh && W + pw >= fw, Code::ERR_CONFIG);
  const long OH = (H + ph - fh + stride) / stride, OW = (W + pw - fw + stride) / stride;
  out_tensor.Reshape(TensorShape({(int64_t)meta.n_filters, OH, OW}));
  const uint64_t base_mod = plain_modulus();
  ENSURE_OR_RETURN(base_mod != (uint64_t)-1, Code::ERR_INTERNAL);
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
                wrap += filters[m](c, r, q) * in_tensor(c, y, x);
            }
        out_tensor((int64_t)m, oh, ow) = base_mod > 1 ? (uint64_t)(sum % base_mod) : wrap;
      }
  return Code::OK;
}

}  // namespace gemini
