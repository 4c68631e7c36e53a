// ONE party of the reference's protocol driver as a process of its own: SCI/src/cheetah/cheetah-api.cpp
// (gemini::CheetahLinear) compiled unchanged over sci::NetIO, linked EITHER with the reference's own
// include/gemini/cheetah/hom_*_ss.cc over the seal:: facade (the slow, line-by-line path: _ref/ref_party_slow) OR with
// the batched implementations of openmm_amd/dropin (_ref/ref_party_fast).  Two such processes, of either kind, talk
// SEAL's serialisation (seal/wire.h) over TCP on 127.0.0.1 exactly as scripts/*_test-run.sh start the reference's two
// parties; tests/test_cross_stack_gpu.py pairs reference-code BOB with fast-path ALICE and the other way round.
// Both parties draw the same cleartext inputs from a common seed; BOB sends his output shares to ALICE at the end of
// every call, as the reference's tests do (tests/bole_test.cpp:150-167), and ALICE checks the reconstruction.
//   usage: ref_party <1 = ALICE | 2 = BOB> <port> [big = 1: also BASELINE configs[2] at full size]
#include <seal/seal.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "cheetah/cheetah-api.h"
#include "utils/constants.h"
#include "utils/net_io_channel.h"

using gemini::CheetahLinear;
using gemini::Padding;
using gemini::Tensor;
using gemini::TensorShape;
using u64 = uint64_t;

static int g_fail = 0, g_party = 0;
static sci::NetIO *g_io = nullptr;
#define EXPECT(cond, what)                               \
  do {                                                   \
    const bool ok_ = (cond);                             \
    std::printf("%s %s\n", ok_ ? "PASS" : "FAIL", what); \
    if (!ok_) ++g_fail;                                  \
  } while (0)
static int64_t to_signed(u64 v, u64 mod) { return v >= (mod + 1) / 2 ? (int64_t)(v - mod) : (int64_t)v; }
static void small(Tensor<u64> &t, std::mt19937_64 &rng, u64 mod, int bits) {
  for (int64_t i = 0; i < t.NumElements(); ++i) {
    const int64_t v = (int64_t)(rng() % (2ull << bits)) - (1ll << bits);
    t.data()[i] = v >= 0 ? (u64)v : mod - (u64)(-v);
  }
}
// BOB -> ALICE: the other party's output share (ALICE gets it into `theirs`)
static void exchange(const Tensor<u64> &mine, Tensor<u64> &theirs) {
  if (g_party == sci::BOB) {
    g_io->send_data(mine.data(), mine.NumElements() * 8);
    g_io->flush();
  } else {
    theirs.Reshape(mine.shape());
    g_io->recv_data(theirs.data(), theirs.NumElements() * 8);
  }
}

static void conv(CheetahLinear &cl, u64 t, TensorShape ishape, int fh, int fw, size_t M, Padding pad, size_t stride,
                 bool shared, std::mt19937_64 &rng, const char *name) {
  CheetahLinear::ConvMeta meta;
  meta.ishape = ishape, meta.fshape = TensorShape({ishape.channels(), fh, fw}), meta.n_filters = M;
  meta.padding = pad, meta.stride = stride, meta.is_shared_input = shared;
  Tensor<u64> image(ishape), in[2] = {Tensor<u64>(ishape), Tensor<u64>(ishape)}, out, other;
  small(image, rng, t, 11);
  for (int64_t i = 0; i < image.NumElements(); ++i) {
    in[1].data()[i] = shared ? rng() % t : image.data()[i];
    in[0].data()[i] = shared ? (image.data()[i] + t - in[1].data()[i]) % t : 0;
  }
  std::vector<Tensor<u64>> filters(M, Tensor<u64>(meta.fshape));
  for (auto &f : filters) small(f, rng, t, 11), f.data()[0] |= 1;
  const auto t0 = std::chrono::steady_clock::now();
  cl.conv2d(in[g_party - 1], filters, meta, out);
  g_io->flush();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  exchange(out, other);
  std::printf("   %s: %.1f ms in this party\n", name, ms);
  if (g_party != sci::ALICE) return;
  int64_t worst = 0;
  bool ok = out.shape().IsSameSize(other.shape()) && out.channels() == (int64_t)M;
  for (size_t m = 0; ok && m < M; ++m) {
    Tensor<u64> one;
    image.Conv2D(filters[m], (int)stride, pad, one);
    ok = one.height() == out.height() && one.width() == out.width();
    for (int64_t h = 0; ok && h < one.height(); ++h)
      for (int64_t w = 0; w < one.width(); ++w)
        worst = std::max<int64_t>(worst, std::llabs(to_signed(((out(m, h, w) + other(m, h, w)) % t + t - one(0, h, w) % t) % t, t)));
  }
  EXPECT(ok && worst <= 1, "conv2d reconstructs within one unit across the two processes");
}

int main(int argc, char **argv) {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);
  if (argc < 3) return 2;
  g_party = std::atoi(argv[1]);
  const int port = std::atoi(argv[2]);
  const bool big = argc > 3 && std::atoi(argv[3]);
  const u64 t = 1ull << 37;
  std::mt19937_64 rng(0x4D264D);
  try {
    g_io = new sci::NetIO(g_party == sci::ALICE ? nullptr : "127.0.0.1", port);
    CheetahLinear cl(g_party, g_io, t, 4);
    g_io->flush();
    const u64 p = sci::default_prime_mod.at(37);
    conv(cl, t, TensorShape({4, 12, 12}), 3, 3, 3, Padding::SAME, 1, true, rng, "4x12x12 3x3 SAME -> 3");
    conv(cl, t, TensorShape({3, 40, 40}), 3, 3, 2, Padding::VALID, 2, false, rng, "3x40x40 3x3 stride 2 -> 2 (client's plain input)");
    conv(cl, t, TensorShape({40, 14, 14}), 1, 1, 24, Padding::VALID, 1, true, rng, "40x14x14 1x1 -> 24");
    if (big) conv(cl, t, TensorShape({64, 56, 56}), 3, 3, 64, Padding::SAME, 1, true, rng, "BASELINE configs[2]: 64x56x56 3x3 SAME -> 64");
    {  // fc
      const int64_t rows = 101, cols = 2048;
      CheetahLinear::FCMeta meta;
      meta.input_shape = TensorShape({cols}), meta.weight_shape = TensorShape({rows, cols}), meta.is_shared_input = true;
      Tensor<u64> x(meta.input_shape), in[2] = {Tensor<u64>(meta.input_shape), Tensor<u64>(meta.input_shape)}, W(meta.weight_shape), out, other;
      small(x, rng, t, 11), small(W, rng, t, 11);
      for (int64_t i = 0; i < cols; ++i) in[1].data()[i] = rng() % t, in[0].data()[i] = (x.data()[i] + t - in[1].data()[i]) % t;
      cl.fc(in[g_party - 1], W, meta, out);
      g_io->flush();
      exchange(out, other);
      if (g_party == sci::ALICE) {
        int64_t worst = 0;
        for (int64_t r = 0; r < rows; ++r) {
          unsigned __int128 acc = 0;
          for (int64_t c = 0; c < cols; ++c) acc += (unsigned __int128)W.data()[r * cols + c] * x.data()[c] % t;
          worst = std::max<int64_t>(worst, std::llabs(to_signed(((out.data()[r] + other.data()[r]) % t + t - (u64)(acc % t)) % t, t)));
        }
        EXPECT(worst <= 1, "fc reconstructs W x within one unit across the two processes");
      }
    }
    {  // bn and BOLE over the field
      const int64_t n = 10000;
      CheetahLinear::BNMeta meta;
      meta.vec_shape = TensorShape({n}), meta.target_base_mod = p, meta.is_shared_input = true;
      Tensor<u64> x(meta.vec_shape), y(meta.vec_shape), xs[2] = {Tensor<u64>(meta.vec_shape), Tensor<u64>(meta.vec_shape)},
                  ys[2] = {Tensor<u64>(meta.vec_shape), Tensor<u64>(meta.vec_shape)}, out, other;
      small(x, rng, p, 11), small(y, rng, p, 11);
      for (int64_t i = 0; i < n; ++i) {
        xs[1].data()[i] = rng() % p, xs[0].data()[i] = (x.data()[i] + p - xs[1].data()[i]) % p;
        ys[1].data()[i] = rng() % p, ys[0].data()[i] = (y.data()[i] + p - ys[1].data()[i]) % p;
      }
      cl.bn(xs[g_party - 1], y, meta, out);
      g_io->flush();
      exchange(out, other);
      if (g_party == sci::ALICE) {
        bool ok = true;
        for (int64_t i = 0; ok && i < n; ++i) ok = (out.data()[i] + other.data()[i]) % p == (u64)((unsigned __int128)x.data()[i] * y.data()[i] % p);
        EXPECT(ok, "bn reconstructs x * scale mod p exactly across the two processes");
      }
      cl.BOLE(xs[g_party - 1], ys[g_party - 1], meta, out);
      g_io->flush();
      exchange(out, other);
      if (g_party == sci::ALICE) {
        bool ok = true;
        for (int64_t i = 0; ok && i < n; ++i) ok = (out.data()[i] + other.data()[i]) % p == (u64)((unsigned __int128)x.data()[i] * y.data()[i] % p);
        EXPECT(ok, "BOLE reconstructs x * y mod p exactly across the two processes (tests/bole_test.cpp:164-167)");
      }
    }
    {  // bn_direct
      CheetahLinear::BNMeta meta;
      meta.ishape = TensorShape({3, 20, 30}), meta.target_base_mod = t, meta.is_shared_input = true;
      Tensor<u64> img(meta.ishape), in[2] = {Tensor<u64>(meta.ishape), Tensor<u64>(meta.ishape)}, scales(TensorShape({3})), out, other;
      small(img, rng, t, 11);
      for (int64_t i = 0; i < img.NumElements(); ++i) in[1].data()[i] = rng() % t, in[0].data()[i] = (img.data()[i] + t - in[1].data()[i]) % t;
      scales.data()[0] = 3, scales.data()[1] = (u64)(-11ll), scales.data()[2] = 4095;
      cl.bn_direct(in[g_party - 1], scales, meta, out);
      g_io->flush();
      exchange(out, other);
      if (g_party == sci::ALICE) {
        bool ok = true;
        for (int64_t c = 0; ok && c < 3; ++c)
          for (int64_t i = 0; ok && i < 600; ++i) {
            const int64_t s = (int64_t)scales.data()[c];
            const u64 v = img.data()[c * 600 + i];
            const u64 want = s >= 0 ? (u64)((unsigned __int128)v * (u64)s % t) : (t - (u64)((unsigned __int128)v * (u64)(-s) % t)) % t;
            ok = (out.data()[c * 600 + i] + other.data()[c * 600 + i]) % t == want;
          }
        EXPECT(ok, "bn_direct reconstructs the channel-wise product exactly across the two processes");
      }
    }
    std::printf("INFO party %d sent %llu bytes\n", g_party, (unsigned long long)cl.io_counter());
  } catch (const std::exception &e) {
    std::printf("FAIL party %d threw: %s\n", g_party, e.what());
    ++g_fail;
  }
  std::printf("%s (%d failures)\n", g_fail ? "FAILED" : "ALL PASSED", g_fail);
  std::fflush(stdout);
  _exit(g_fail ? 1 : 0);
}
