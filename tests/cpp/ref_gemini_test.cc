// The reference's OWN protocol classes, compiled unchanged from /root/reference/include/gemini/cheetah/*.cc against
// the seal:: facade of this repository (include/seal, include/eigen_stub) and linked with libmm_hip.so: HomConv2DSS,
// HomFCSS and HomBNSS run their two-party flows here the way SCI/src/cheetah/cheetah-api.cpp drives them (keys and
// contexts :95-141,:151-196; conv2d :397-445; fc :331-375; bn :540-603; BOLE :605-650; bn_direct :652-706), both
// parties in one process, ciphertexts passed through save()/load() as send_encrypted_vector / recv_encrypted_vector do.
// Checks: the reference's acceptance identities (shares reconstruct to HomConv2DSS::idealFunctionality within one unit
// in the shipped USE_APPROX_RESHARE build, FC likewise, BOLE / bn exactly, bn_direct exactly), and that the plaintexts
// the reference's TensorEncoder / HomFCSS encoders produce equal, word for word, what the device encoders of the fast
// path (mmh_conv_encode_image / _filters, mmh_fc_encode_*) produce for the same tensors.
// Built by tests/cpp/Makefile into tests/cpp/_ref/ only where /root/reference exists; the binary travels to the GPU box.
#include <seal/seal.h>

#include <cstdio>
#include <random>
#include <sstream>

#include "gemini/cheetah/hom_bn_ss.h"
#include "gemini/cheetah/hom_conv2d_ss.h"
#include "gemini/cheetah/hom_fc_ss.h"
#include "gemini/cheetah/tensor_encoder.h"
#include "mm_hip.h"

using gemini::HomBNSS;
using gemini::HomConv2DSS;
using gemini::HomFCSS;
using gemini::Padding;
using gemini::Tensor;
using gemini::TensorShape;
using u64 = uint64_t;

static int g_fail = 0;
#define EXPECT(cond, what)                                   \
  do {                                                       \
    const bool ok_ = (cond);                                 \
    std::printf("%s %s\n", ok_ ? "PASS" : "FAIL", what);     \
    if (!ok_) ++g_fail;                                      \
  } while (0)
#define OK(code, what)                                                                  \
  do {                                                                                  \
    const Code c_ = (code);                                                             \
    if (c_ != Code::OK) {                                                               \
      std::printf("FAIL %s -> %s\n", what, CodeMessage(c_).c_str());                    \
      ++g_fail;                                                                         \
      return;                                                                           \
    }                                                                                   \
  } while (0)

struct Party {  // what CheetahLinear's constructor sets up for one plain modulus (cheetah-api.cpp:95-141)
  std::shared_ptr<seal::SEALContext> context;
  std::optional<seal::SecretKey> sk;
  std::shared_ptr<seal::PublicKey> pk;
};

static Party make_keys(u64 plain_mod) {
  seal::EncryptionParameters parms(seal::scheme_type::bfv);
  parms.set_n_special_primes(0);
  parms.set_poly_modulus_degree(4096);
  parms.set_coeff_modulus(seal::CoeffModulus::Create(4096, {60, 49}));
  parms.set_plain_modulus(plain_mod);
  Party p;
  p.context = std::make_shared<seal::SEALContext>(parms, true, seal::sec_level_type::tc128);
  seal::KeyGenerator keygen(*p.context);
  p.sk = keygen.secret_key();
  auto pk = keygen.create_public_key();
  std::stringstream ss;  // the wire between the parties (:113-141)
  pk.save(ss);
  p.pk = std::make_shared<seal::PublicKey>();
  p.pk->load(*p.context, ss);
  return p;
}

// send_encrypted_vector / recv_encrypted_vector (cheetah-api.cpp:11-66, :710-737): save on one side, load on the other
template <class CtVec>
static std::vector<seal::Ciphertext> over_the_wire(const CtVec &cts, const seal::SEALContext &context, bool unsafe) {
  std::vector<seal::Ciphertext> out(cts.size());
  for (size_t i = 0; i < cts.size(); ++i) {
    std::stringstream ss;
    cts[i].save(ss);
    if (unsafe) out[i].unsafe_load(context, ss);
    else out[i].load(context, ss);
  }
  return out;
}

static int64_t to_signed(u64 v, u64 mod) { return v >= (mod + 1) / 2 ? (int64_t)(v - mod) : (int64_t)v; }

static void fill(Tensor<u64> &t, std::mt19937_64 &rng, u64 mod) {
  for (int64_t i = 0; i < t.NumElements(); ++i) t.data()[i] = rng() % mod;
}
static void fill_small(Tensor<u64> &t, std::mt19937_64 &rng, u64 mod, int bits) {  // signed, |v| < 2^bits, mod `mod`
  for (int64_t i = 0; i < t.NumElements(); ++i) {
    const int64_t v = (int64_t)(rng() % (2ull << bits)) - (1ll << bits);
    t.data()[i] = v >= 0 ? (u64)v : mod - (u64)(-v);
  }
}

// the device encoders of the fast path on the same tensors
struct DevBuf {
  void *p = nullptr;
  explicit DevBuf(size_t words) { mmh_malloc(&p, words * 8); }
  ~DevBuf() { mmh_free(p); }
  u64 *get() const { return (u64 *)p; }
};

static void conv_case(const Party &k, u64 t, TensorShape ishape, int fh, int fw, size_t M, Padding pad, size_t stride,
                      bool shared, std::mt19937_64 &rng, const char *name) {
  std::printf("-- HomConv2DSS %s\n", name);
  HomConv2DSS client, server;
  OK(client.setUp(*k.context, k.sk, nullptr), "client setUp");
  OK(server.setUp(*k.context, std::nullopt, k.pk), "server setUp");
  HomConv2DSS::Meta meta;
  meta.ishape = ishape;
  meta.fshape = TensorShape({ishape.channels(), fh, fw});
  meta.n_filters = M;
  meta.padding = pad;
  meta.stride = stride;
  meta.is_shared_input = shared;

  Tensor<u64> image(ishape), share0(ishape), share1(ishape);
  fill_small(image, rng, t, 11);
  fill(share0, rng, t);
  for (int64_t i = 0; i < image.NumElements(); ++i) share1.data()[i] = (image.data()[i] + t - share0.data()[i]) % t;
  std::vector<Tensor<u64>> filters(M, Tensor<u64>(meta.fshape));
  for (auto &f : filters) {
    fill_small(f, rng, t, 11);
    f.data()[0] |= 1;
  }

  std::vector<seal::Serializable<seal::Ciphertext>> enc;
  OK(client.encryptImage(shared ? share0 : image, meta, enc, 2), "encryptImage");
  auto cts = over_the_wire(enc, *k.context, false);
  std::vector<seal::Plaintext> share_pts;
  if (shared) OK(server.encodeImage(share1, meta, share_pts, 2), "encodeImage");
  std::vector<std::vector<seal::Plaintext>> filter_pts;
  OK(server.encodeFilters(filters, meta, filter_pts, 2), "encodeFilters");

  {  // the reference's encoders against the device encoders of the fast path
    mmh_ctx *ctx = nullptr;
    const int bits[2] = {60, 49};
    mmh_ctx_create_bits(&ctx, 4096, bits, 2, t, 0);
    mmh_conv_meta m{ishape.channels(), ishape.height(), ishape.width(), fh, fw, (int64_t)M, pad == Padding::SAME, (int32_t)stride, shared};
    mmh_conv_plan *plan = nullptr;
    bool same = mmh_conv_plan_create(ctx, &m, &plan) == MMH_OK;
    size_t n_in = 0, n_cs = 0, n_sp = 0, n_out = 0;
    if (same) mmh_conv_plan_info(plan, &n_in, &n_cs, &n_sp, &n_out, nullptr, nullptr);
    same = same && n_in == cts.size() && n_cs == filter_pts[0].size();
    if (same) {
      const size_t N = 4096, img_words = (size_t)ishape.num_elements(), f_words = (size_t)meta.fshape.num_elements();
      DevBuf d_img(img_words), d_pts(n_in * N), d_f(M * f_words), d_fp(M * n_cs * N);
      std::vector<u64> host(std::max(n_in, M * n_cs) * N), fh_(M * f_words);
      mmh_memcpy_h2d(d_img.get(), share1.data(), img_words * 8, nullptr);
      for (size_t i = 0; i < M; ++i) std::copy_n(filters[i].data(), f_words, fh_.begin() + i * f_words);
      mmh_memcpy_h2d(d_f.get(), fh_.data(), fh_.size() * 8, nullptr);
      same = mmh_conv_encode_image(plan, d_img.get(), d_pts.get(), nullptr) == MMH_OK &&
             mmh_conv_encode_filters(plan, d_f.get(), d_fp.get(), nullptr) == MMH_OK;
      mmh_memcpy_d2h(host.data(), d_pts.get(), n_in * N * 8, nullptr);
      mmh_stream_sync(nullptr);
      if (shared)
        for (size_t i = 0; i < n_in && same; ++i)
          for (size_t j = 0; j < N && same; ++j) same = host[i * N + j] == (j < share_pts[i].coeff_count() ? share_pts[i][j] : 0);
      mmh_memcpy_d2h(host.data(), d_fp.get(), M * n_cs * N * 8, nullptr);
      mmh_stream_sync(nullptr);
      for (size_t i = 0; i < M && same; ++i)
        for (size_t c = 0; c < n_cs && same; ++c)
          for (size_t j = 0; j < N && same; ++j)
            same = host[(i * n_cs + c) * N + j] == (j < filter_pts[i][c].coeff_count() ? filter_pts[i][c][j] : 0);
    }
    mmh_conv_plan_destroy(plan);
    mmh_ctx_destroy(ctx);
    EXPECT(same, "reference TensorEncoder plaintexts == device encoders of the fast path");
  }

  std::vector<seal::Ciphertext> out_cts;
  Tensor<u64> out_share1, out_share0, ideal;
  OK(server.conv2DSS(cts, share_pts, filter_pts, meta, out_cts, out_share1, 2), "conv2DSS");
  auto back = over_the_wire(out_cts, *k.context, true);  // truncated ciphertexts: unsafe_load (cheetah-api.cpp:410)
  OK(client.decryptToTensor(back, meta, out_share0, 2), "decryptToTensor");
  OK(server.idealFunctionality(image, filters, meta, ideal), "idealFunctionality");
  bool ok = out_share0.shape().IsSameSize(ideal.shape()) && out_share1.shape().IsSameSize(ideal.shape());
  int64_t worst = 0;
  for (int64_t i = 0; ok && i < ideal.NumElements(); ++i) {
    const u64 got = (out_share0.data()[i] + out_share1.data()[i]) % t;
    const int64_t d = to_signed((got + t - ideal.data()[i]) % t, t);
    worst = std::max<int64_t>(worst, std::llabs(d));
  }
  std::printf("   max |reconstruction - ideal| = %lld over %lld outputs, %zu -> %zu ciphertexts\n", (long long)worst,
              (long long)ideal.NumElements(), cts.size(), out_cts.size());
  EXPECT(ok && worst <= 1, "conv2DSS shares reconstruct to idealFunctionality within one unit");
}

static void fc_case(const Party &k, u64 t, int64_t rows, int64_t cols, std::mt19937_64 &rng) {
  std::printf("-- HomFCSS %lld x %lld\n", (long long)rows, (long long)cols);
  HomFCSS client, server;
  OK(client.setUp(*k.context, k.sk, nullptr), "client setUp");
  OK(server.setUp(*k.context, std::nullopt, k.pk), "server setUp");
  HomFCSS::Meta meta;
  meta.input_shape = TensorShape({cols});
  meta.weight_shape = TensorShape({rows, cols});
  meta.is_shared_input = true;
  Tensor<u64> vec(meta.input_shape), s0(meta.input_shape), s1(meta.input_shape), W(meta.weight_shape);
  fill_small(vec, rng, t, 11);
  fill(s0, rng, t);
  for (int64_t i = 0; i < cols; ++i) s1.data()[i] = (vec.data()[i] + t - s0.data()[i]) % t;
  fill_small(W, rng, t, 11);
  std::vector<seal::Serializable<seal::Ciphertext>> enc;
  OK(client.encryptInputVector(s0, meta, enc, 2), "encryptInputVector");
  auto cts = over_the_wire(enc, *k.context, false);
  std::vector<seal::Plaintext> s1_pts;
  OK(server.encodeInputVector(s1, meta, s1_pts, 2), "encodeInputVector");
  std::vector<std::vector<seal::Plaintext>> w_pts;
  OK(server.encodeWeightMatrix(W, meta, w_pts, 2), "encodeWeightMatrix");

  {  // encoders against the device encoders
    mmh_ctx *ctx = nullptr;
    const int bits[2] = {60, 49};
    mmh_ctx_create_bits(&ctx, 4096, bits, 2, t, 0);
    mmh_fc_meta m{rows, cols, 1};
    mmh_fc_plan *plan = nullptr;
    bool same = mmh_fc_plan_create(ctx, &m, &plan) == MMH_OK;
    size_t n_in = 0, n_out = 0;
    if (same) mmh_fc_plan_info(plan, nullptr, &n_in, &n_out);
    same = same && n_in == s1_pts.size() && n_out == w_pts.size();
    if (same) {
      const size_t N = 4096;
      DevBuf d_v(cols), d_vp(n_in * N), d_w(rows * cols), d_wp(n_out * n_in * N);
      std::vector<u64> host(n_out * n_in * N);
      mmh_memcpy_h2d(d_v.get(), s1.data(), cols * 8, nullptr);
      mmh_memcpy_h2d(d_w.get(), W.data(), rows * cols * 8, nullptr);
      same = mmh_fc_encode_input(plan, d_v.get(), d_vp.get(), nullptr) == MMH_OK &&
             mmh_fc_encode_weights(plan, d_w.get(), d_wp.get(), nullptr) == MMH_OK;
      mmh_memcpy_d2h(host.data(), d_vp.get(), n_in * N * 8, nullptr);
      mmh_stream_sync(nullptr);
      for (size_t i = 0; i < n_in && same; ++i)
        for (size_t j = 0; j < N && same; ++j) same = host[i * N + j] == (j < s1_pts[i].coeff_count() ? s1_pts[i][j] : 0);
      mmh_memcpy_d2h(host.data(), d_wp.get(), n_out * n_in * N * 8, nullptr);
      mmh_stream_sync(nullptr);
      for (size_t r = 0; r < n_out && same; ++r)
        for (size_t c = 0; c < n_in && same; ++c)
          for (size_t j = 0; j < N && same; ++j)
            same = host[(r * n_in + c) * N + j] == (j < w_pts[r][c].coeff_count() ? w_pts[r][c][j] : 0);
    }
    mmh_fc_plan_destroy(plan);
    mmh_ctx_destroy(ctx);
    EXPECT(same, "reference HomFCSS encoders == device encoders of the fast path (stale coefficient included)");
  }

  std::vector<seal::Ciphertext> out_cts;
  Tensor<u64> out1, out0;
  OK(server.matVecMul(w_pts, cts, s1_pts, meta, out_cts, out1, 2), "matVecMul");
  auto back = over_the_wire(out_cts, *k.context, true);
  OK(client.decryptToVector(back, meta, out0, 2), "decryptToVector");
  int64_t worst = 0;
  for (int64_t r = 0; r < rows; ++r) {
    unsigned __int128 acc = 0;
    for (int64_t c = 0; c < cols; ++c) acc += (unsigned __int128)W.data()[r * cols + c] * vec.data()[c] % t;
    const u64 want = (u64)(acc % t), got = (out0.data()[r] + out1.data()[r]) % t;
    worst = std::max<int64_t>(worst, std::llabs(to_signed((got + t - want) % t, t)));
  }
  std::printf("   max |reconstruction - W x| = %lld\n", (long long)worst);
  EXPECT(worst <= 1, "matVecMul shares reconstruct to W x within one unit");
}

static void bn_cases(const Party &field, const Party &ring, u64 p, u64 t, std::mt19937_64 &rng) {
  std::printf("-- HomBNSS (SIMD path: bn, BOLE)\n");
  HomBNSS client, server;
  OK(client.setUp_Field(p, *field.context, field.sk, nullptr), "client setUp_Field");
  OK(server.setUp_Field(p, *field.context, std::nullopt, field.pk), "server setUp_Field");
  OK(client.setUp_Base(t, *ring.context, ring.sk, nullptr), "client setUp_Base");
  OK(server.setUp_Base(t, *ring.context, std::nullopt, ring.pk), "server setUp_Base");
  const int64_t n = 4096 + 700;  // two ciphertexts, the second ragged
  HomBNSS::Meta meta;
  meta.vec_shape = TensorShape({n});
  meta.target_base_mod = p;
  meta.is_shared_input = true;
  Tensor<u64> x(meta.vec_shape), y(meta.vec_shape), x0(meta.vec_shape), x1(meta.vec_shape), y0(meta.vec_shape), y1(meta.vec_shape);
  fill_small(x, rng, p, 11), fill_small(y, rng, p, 11);
  fill(x0, rng, p), fill(y0, rng, p);
  for (int64_t i = 0; i < n; ++i) x1.data()[i] = (x.data()[i] + p - x0.data()[i]) % p, y1.data()[i] = (y.data()[i] + p - y0.data()[i]) % p;
  {  // bn: server holds the scale vector y in the clear (cheetah-api.cpp:540-603)
    std::vector<seal::Serializable<seal::Ciphertext>> enc;
    OK(client.encryptVector(x0, meta, enc, 2), "encryptVector");
    auto cts = over_the_wire(enc, *field.context, false);
    std::vector<seal::Plaintext> x1_pts, sc_pts;
    OK(server.encodeVector(x1, meta, x1_pts, 2), "encodeVector");
    OK(server.encodeScales(y, meta, sc_pts, 2), "encodeScales");
    std::vector<seal::Ciphertext> out_cts;
    Tensor<u64> o1, o0;
    OK(server.bn(cts, x1_pts, sc_pts, meta, out_cts, o1, 2), "bn");
    auto back = over_the_wire(out_cts, *field.context, true);
    OK(client.decryptToVector(back, meta, o0, 2), "decryptToVector");
    bool ok = true;
    for (int64_t i = 0; ok && i < n; ++i)
      ok = (o0.data()[i] + o1.data()[i]) % p == (u64)((unsigned __int128)x.data()[i] * y.data()[i] % p);
    EXPECT(ok, "bn shares reconstruct to x * scale mod p exactly");
  }
  {  // BOLE (cheetah-api.cpp:605-650; tests/bole_test.cpp:164-167)
    std::vector<seal::Serializable<seal::Ciphertext>> ex, ey;
    OK(client.encryptVector(x0, meta, ex, 2), "encryptVector x");
    OK(client.encryptVector(y0, meta, ey, 2), "encryptVector y");
    auto cx = over_the_wire(ex, *field.context, false), cy = over_the_wire(ey, *field.context, false);
    std::vector<seal::Plaintext> px, py;
    OK(server.encodeVector(x1, meta, px, 2), "encodeVector x");
    OK(server.encodeVector(y1, meta, py, 2), "encodeVector y");
    std::vector<seal::Ciphertext> out_cts;
    Tensor<u64> o1, o0;
    OK(server.bole(cx, cy, px, py, meta, out_cts, o1, 2), "bole");
    auto back = over_the_wire(out_cts, *field.context, true);
    OK(client.decryptToVector(back, meta, o0, 2), "decryptToVector");
    bool ok = true;
    for (int64_t i = 0; ok && i < n; ++i) {  // + local cross terms x_i y_i on both sides (:648-649)
      const u64 a = (o1.data()[i] + (u64)((unsigned __int128)x1.data()[i] * y1.data()[i] % p)) % p;
      const u64 b = (o0.data()[i] + (u64)((unsigned __int128)x0.data()[i] * y0.data()[i] % p)) % p;
      ok = (a + b) % p == (u64)((unsigned __int128)x.data()[i] * y.data()[i] % p);
    }
    EXPECT(ok, "BOLE shares reconstruct to x * y mod p exactly");
  }
  std::printf("-- HomBNSS (direct path: bn_direct)\n");
  {
    HomBNSS::Meta dm;
    dm.ishape = TensorShape({3, 70, 9});  // a tile ragged in h
    dm.target_base_mod = t;
    dm.is_shared_input = true;
    Tensor<u64> img(dm.ishape), s0(dm.ishape), s1(dm.ishape), scales(TensorShape({3}));
    fill_small(img, rng, t, 11);
    fill(s0, rng, t);
    for (int64_t i = 0; i < img.NumElements(); ++i) s1.data()[i] = (img.data()[i] + t - s0.data()[i]) % t;
    scales.data()[0] = 5, scales.data()[1] = (u64)(-7ll), scales.data()[2] = 1000;  // read as int64 (hom_bn_ss.cc:615)
    std::vector<seal::Serializable<seal::Ciphertext>> enc;
    OK(client.encryptTensor(s0, dm, enc, 2), "encryptTensor");
    auto cts = over_the_wire(enc, *ring.context, false);
    std::vector<seal::Plaintext> s1_pts;
    OK(server.encodeTensor(s1, dm, s1_pts, 2), "encodeTensor");
    std::vector<seal::Ciphertext> out_cts;
    Tensor<u64> o1, o0;
    OK(server.bn_direct(cts, s1_pts, scales, dm, out_cts, o1, 2), "bn_direct");
    auto back = over_the_wire(out_cts, *ring.context, false);  // bn_direct results are load()ed (cheetah-api.cpp:672)
    OK(client.decryptToTensor(back, dm, o0, 2), "decryptToTensor");
    bool ok = true;
    for (int64_t c = 0; ok && c < 3; ++c)
      for (int64_t i = 0; ok && i < 70 * 9; ++i) {
        const int64_t s = (int64_t)scales.data()[c];
        const u64 v = img.data()[c * 630 + i];
        const u64 want = s >= 0 ? (u64)((unsigned __int128)v * (u64)s % t) : (t - (u64)((unsigned __int128)v * (u64)(-s) % t)) % t;
        ok = (o0.data()[c * 630 + i] + o1.data()[c * 630 + i]) % t == want;
      }
    EXPECT(ok, "bn_direct shares reconstruct to the channel-wise product exactly");
  }
}

int main() {
  const u64 t = 1ull << 37, p = 137438822401ull;  // ell = 37 and its field prime (SCI/src/utils/constants.h:47)
  std::mt19937_64 rng(0x4D264D);
  try {
    Party ring = make_keys(t), field = make_keys(p);
    EXPECT(ring.context->first_parms_id() == ring.context->key_parms_id() &&
               ring.context->last_context_data()->parms().coeff_modulus().size() == 1,
           "n_special_primes = 0: first level = key level, last level = {q0} (patch/SEAL.patch:28-37)");
    EXPECT(field.context->first_context_data()->qualifiers().using_batching &&
               !ring.context->first_context_data()->qualifiers().using_batching,
           "batching qualifier: prime plain modulus only");
    conv_case(ring, t, TensorShape({4, 12, 12}), 3, 3, 3, Padding::SAME, 1, true, rng, "4x12x12 3x3 SAME -> 3, shared");
    conv_case(ring, t, TensorShape({5, 11, 11}), 1, 1, 2, Padding::VALID, 2, true, rng, "5x11x11 1x1 stride 2 -> 2, shared");
    conv_case(ring, t, TensorShape({2, 70, 70}), 3, 3, 2, Padding::SAME, 1, false, rng, "2x70x70 3x3 SAME -> 2, spatial slices, not shared");
    conv_case(ring, t, TensorShape({130, 7, 7}), 1, 1, 2, Padding::VALID, 1, true, rng, "130x7x7 1x1 -> 2, two channel slices");
    fc_case(ring, t, 10, 300, rng);
    fc_case(ring, t, 101, 2048, rng);
    bn_cases(field, ring, p, t, rng);
  } catch (const std::exception &e) {
    std::printf("FAIL exception: %s\n", e.what());
    ++g_fail;
  }
  std::printf("%s (%d failures)\n", g_fail ? "FAILED" : "ALL PASSED", g_fail);
  return g_fail ? 1 : 0;
}
