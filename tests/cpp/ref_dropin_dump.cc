// Drives the batched implementations of the reference's protocol classes (openmm_amd/dropin behind the reference's own
// include/gemini/cheetah/hom_*_ss.h) with INJECTED randomness and writes inputs and outputs to a file, so that
// tests/test_dropin_gpu.py can replay the same layer through the CPU oracle (tests/oracle_layers.py) and compare the
// ciphertexts bit for bit.  The server objects are set up on a context whose parameters carry a
// seal::UniformRandomGeneratorFactory with a fixed key: the n-th generator it makes has key K with k[7] += n, and a
// layer call draws its randomness from (that key, consecutive stream ids from 1) through the device samplers, which the
// test reproduces with mmh_sample_reshare / mmh_chacha20_words.
//   usage: ref_dropin_dump OUT_FILE
#include <seal/seal.h>

#include <cstdio>
#include <random>
#include <string>

#include "gemini/cheetah/hom_bn_ss.h"
#include "gemini/cheetah/hom_conv2d_ss.h"
#include "gemini/cheetah/hom_fc_ss.h"

using namespace gemini;
using u64 = uint64_t;

static FILE *g_out = nullptr;
static void put(const std::string &name, const std::vector<u64> &dims, const u64 *data) {
  const uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)dims.size();
  std::fwrite(&nl, 4, 1, g_out), std::fwrite(name.data(), 1, nl, g_out), std::fwrite(&nd, 4, 1, g_out);
  size_t n = 1;
  for (u64 d : dims) std::fwrite(&d, 8, 1, g_out), n *= d;
  std::fwrite(data, 8, n, g_out);
}
static void put_cts(const std::string &name, const std::vector<seal::Ciphertext> &cts) {
  if (cts.empty()) return;
  const size_t L = cts[0].coeff_modulus_size(), N = cts[0].poly_modulus_degree();
  std::vector<u64> all;
  for (const auto &c : cts) all.insert(all.end(), c.data(), c.data() + 2 * L * N);
  put(name, {cts.size(), 2, L, N}, all.data());
}
static std::vector<seal::Ciphertext> open_fresh(const std::vector<seal::Serializable<seal::Ciphertext>> &v) {
  std::vector<seal::Ciphertext> out;
  for (const auto &s : v) out.push_back(s.facade_object());
  return out;
}
#define MUST(expr)                                                                \
  do {                                                                            \
    const Code c_ = (expr);                                                       \
    if (c_ != Code::OK) {                                                         \
      std::printf("FAIL %s -> %s\n", #expr, CodeMessage(c_).c_str());             \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  g_out = std::fopen(argv[1], "wb");
  if (!g_out) return 2;
  std::mt19937_64 rng(0x4D264D);
  const std::array<uint32_t, 8> K{0x4D264D, 1, 2, 3, 4, 5, 6, 100};
  auto fixed = std::make_shared<seal::UniformRandomGeneratorFactory>(K);
  auto make_parms = [&](u64 t, bool inject) {
    seal::EncryptionParameters parms(seal::scheme_type::bfv);
    parms.set_n_special_primes(0);
    parms.set_poly_modulus_degree(4096);
    parms.set_coeff_modulus(seal::CoeffModulus::Create(4096, {60, 49}));
    parms.set_plain_modulus(t);
    if (inject) parms.set_random_generator(fixed);
    return parms;
  };
  const u64 t = 1ull << 37, p = 137438822401ull;
  put("key", {8}, std::vector<u64>(K.begin(), K.end()).data());
  {  // ---------------- ring context: conv and fc
    seal::SEALContext client_ctx(make_parms(t, false), true, seal::sec_level_type::tc128);
    seal::SEALContext server_ctx(make_parms(t, true), true, seal::sec_level_type::tc128);
    seal::KeyGenerator keygen(client_ctx);
    seal::SecretKey sk = keygen.secret_key();
    auto pk = std::make_shared<seal::PublicKey>();
    keygen.create_public_key(*pk);
    put("sk", {2, 4096}, sk.data().data());
    put("pk", {2, 2, 4096}, pk->data().data());
    {  // conv: 20x14x14, 3x3 SAME, 12 filters, shared input (two channel slices)
      HomConv2DSS::Meta meta;
      meta.ishape = TensorShape({20, 14, 14}), meta.fshape = TensorShape({20, 3, 3}), meta.n_filters = 12;
      meta.padding = Padding::SAME, meta.stride = 1, meta.is_shared_input = true;
      Tensor<u64> s0(meta.ishape), s1(meta.ishape), out_share;
      for (int64_t i = 0; i < s0.NumElements(); ++i) s0.data()[i] = rng() % t, s1.data()[i] = rng() % t;
      std::vector<Tensor<u64>> filters(meta.n_filters, Tensor<u64>(meta.fshape));
      for (auto &f : filters)
        for (int64_t i = 0; i < f.NumElements(); ++i) f.data()[i] = (u64)((int64_t)(rng() % 8191) - 4095) & (t - 1);
      HomConv2DSS client, server;
      MUST(client.setUp(client_ctx, sk));
      MUST(server.setUp(server_ctx, std::nullopt, pk));   // generator #0 of the fixed factory
      std::vector<seal::Serializable<seal::Ciphertext>> fresh;
      MUST(client.encryptImage(s0, meta, fresh, 4));
      std::vector<seal::Ciphertext> cts = open_fresh(fresh), out_cts;
      std::vector<seal::Plaintext> enc_share;
      std::vector<std::vector<seal::Plaintext>> enc_filters;
      MUST(server.encodeFilters(filters, meta, enc_filters, 4));
      MUST(server.encodeImage(s1, meta, enc_share, 4));
      MUST(server.conv2DSS(cts, enc_share, enc_filters, meta, out_cts, out_share, 4));
      Tensor<u64> dec;
      MUST(client.decryptToTensor(out_cts, meta, dec, 4));
      std::vector<u64> fl;
      for (auto &f : filters) fl.insert(fl.end(), f.data(), f.data() + f.NumElements());
      put("conv_share0", {20, 14, 14}, s0.data()), put("conv_share1", {20, 14, 14}, s1.data());
      put("conv_filters", {12, 20, 3, 3}, fl.data());
      put_cts("conv_in_cts", cts), put_cts("conv_out_cts", out_cts);
      put("conv_out_share", {(u64)out_share.NumElements()}, out_share.data());
      put("conv_dec", {(u64)dec.NumElements()}, dec.data());
      // a plaintext of the lazy batches, brought to the host: must be what the reference's encoders write
      put("conv_filter_pt_5_1", {enc_filters[5][1].coeff_count()}, enc_filters[5][1].data());
      put("conv_share_pt_1", {enc_share[1].coeff_count()}, enc_share[1].data());
    }
    {  // fc: 70 x 300, shared input
      HomFCSS::Meta meta;
      meta.weight_shape = TensorShape({70, 300}), meta.input_shape = TensorShape({300}), meta.is_shared_input = true;
      Tensor<u64> v0(meta.input_shape), v1(meta.input_shape), W(meta.weight_shape), out_share, dec;
      for (int64_t i = 0; i < 300; ++i) v0.data()[i] = rng() % t, v1.data()[i] = rng() % t;
      for (int64_t i = 0; i < W.NumElements(); ++i) W.data()[i] = ((u64)((int64_t)(rng() % 8191) - 4095) | 1) & (t - 1);
      HomFCSS client, server;
      MUST(client.setUp(client_ctx, sk));
      MUST(server.setUp(server_ctx, std::nullopt, pk));   // generator #1
      std::vector<seal::Serializable<seal::Ciphertext>> fresh;
      MUST(client.encryptInputVector(v0, meta, fresh, 2));
      std::vector<seal::Ciphertext> cts = open_fresh(fresh), out_cts;
      std::vector<seal::Plaintext> enc_vec;
      std::vector<std::vector<seal::Plaintext>> enc_w;
      MUST(server.encodeWeightMatrix(W, meta, enc_w, 2));
      MUST(server.encodeInputVector(v1, meta, enc_vec, 2));
      MUST(server.matVecMul(enc_w, cts, enc_vec, meta, out_cts, out_share, 2));
      MUST(client.decryptToVector(out_cts, meta, dec, 2));
      put("fc_vec0", {300}, v0.data()), put("fc_vec1", {300}, v1.data()), put("fc_w", {70, 300}, W.data());
      put_cts("fc_in_cts", cts), put_cts("fc_out_cts", out_cts);
      put("fc_out_share", {70}, out_share.data()), put("fc_dec", {70}, dec.data());
    }
  }
  {  // ---------------- field context: bn (SIMD)
    seal::SEALContext client_ctx(make_parms(p, false), true, seal::sec_level_type::tc128);
    seal::SEALContext server_ctx(make_parms(p, true), true, seal::sec_level_type::tc128);
    seal::KeyGenerator keygen(client_ctx);
    seal::SecretKey sk = keygen.secret_key();
    auto pk = std::make_shared<seal::PublicKey>();
    keygen.create_public_key(*pk);
    put("bn_sk", {2, 4096}, sk.data().data());
    const int64_t n = 10000;
    HomBNSS::Meta meta;
    meta.vec_shape = TensorShape({n}), meta.target_base_mod = p, meta.is_shared_input = true;
    Tensor<u64> x0(meta.vec_shape), x1(meta.vec_shape), sc(meta.vec_shape), out_share, dec;
    for (int64_t i = 0; i < n; ++i) x0.data()[i] = rng() % p, x1.data()[i] = rng() % p, sc.data()[i] = rng() % p;
    HomBNSS client, server;
    MUST(client.setUp_Field(p, client_ctx, sk));
    MUST(server.setUp_Field(p, server_ctx, std::nullopt, pk));   // generator #2
    std::vector<seal::Serializable<seal::Ciphertext>> fresh;
    MUST(client.encryptVector(x0, meta, fresh, 2));
    std::vector<seal::Ciphertext> cts = open_fresh(fresh), out_cts;
    std::vector<seal::Plaintext> enc_x, enc_sc;
    MUST(server.encodeVector(x1, meta, enc_x, 2));
    MUST(server.encodeScales(sc, meta, enc_sc, 2));
    MUST(server.bn(cts, enc_x, enc_sc, meta, out_cts, out_share, 2));
    MUST(client.decryptToVector(out_cts, meta, dec, 2));
    put("bn_x0", {(u64)n}, x0.data()), put("bn_x1", {(u64)n}, x1.data()), put("bn_scale", {(u64)n}, sc.data());
    put_cts("bn_in_cts", cts), put_cts("bn_out_cts", out_cts);
    put("bn_out_share", {(u64)n}, out_share.data()), put("bn_dec", {(u64)n}, dec.data());
  }
  std::fclose(g_out);
  std::printf("ALL WRITTEN\n");
  return 0;
}
