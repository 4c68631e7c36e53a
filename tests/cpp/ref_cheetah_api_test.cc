// The reference's protocol driver itself: SCI/src/cheetah/cheetah-api.cpp (gemini::CheetahLinear) compiled UNCHANGED,
// together with the include/gemini/cheetah/*.cc it drives, against the seal:: facade of this repository and linked
// with libmm_hip.so; the two parties run as two threads of this process and talk through the reference's own
// sci::NetIO (SCI/src/utils/net_io_channel.h: a TCP socket on 127.0.0.1), exactly as tests/bole_test.cpp sets them up
// (:196-197).  conv2d, fc, bn, BOLE and bn_direct are called with the reference's signatures and checked the way the
// reference's own programs check them: shares reconstruct to the cleartext result within one unit for conv / fc
// (library_fixed_uniform_cheetah.cpp:199,:404: the shipped build re-shares approximately), exactly for bn, BOLE
// (tests/bole_test.cpp:164-167) and bn_direct.  The BASELINE.json configs[2] layer (64x56x56, 3x3 SAME, 64 filters)
// runs through it at full size.
// Built by tests/cpp/Makefile into tests/cpp/_ref/ only where /root/reference exists.
#include <seal/seal.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <functional>
#include <random>
#include <thread>

#include "cheetah/cheetah-api.h"
#include "utils/constants.h"
#include "utils/net_io_channel.h"

using gemini::CheetahLinear;
using gemini::Padding;
using gemini::Tensor;
using gemini::TensorShape;
using u64 = uint64_t;

static int g_fail = 0;
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

struct Pair {
  sci::NetIO *io[2] = {nullptr, nullptr};
  CheetahLinear *cl[2] = {nullptr, nullptr};
  // run fn(party index 0 = ALICE/server, 1 = BOB/client) on both parties
  // (a party that throws would leave the other one waiting on the socket for ever: report and end the process)
  void both(const std::function<void(int)> &fn) {
    auto run = [&](int s) {
      try {
        fn(s);
        // cheetah-api.cpp leaves its last message in NetIO's buffer (the caller's next receive would flush it,
        // net_io_channel.h:172-173); the two threads meet after every call here, so push it out
        if (io[s]) io[s]->flush();
      } catch (const std::exception &e) {
        std::printf("FAIL %s threw: %s\nFAILED\n", s == 0 ? "ALICE" : "BOB", e.what());
        std::fflush(stdout);
        _exit(1);
      }
    };
    std::thread a(run, 0), b(run, 1);
    a.join();
    b.join();
  }
};

static void conv(Pair &P, u64 t, TensorShape ishape, int fh, int fw, size_t M, Padding pad, size_t stride, bool shared,
                 std::mt19937_64 &rng, const char *name) {
  CheetahLinear::ConvMeta meta;
  meta.ishape = ishape;
  meta.fshape = TensorShape({ishape.channels(), fh, fw});
  meta.n_filters = M;
  meta.padding = pad;
  meta.stride = stride;
  meta.is_shared_input = shared;
  Tensor<u64> image(ishape), in[2] = {Tensor<u64>(ishape), Tensor<u64>(ishape)}, out[2];
  small(image, rng, t, 11);
  for (int64_t i = 0; i < image.NumElements(); ++i) {
    in[1].data()[i] = shared ? rng() % t : image.data()[i];  // BOB holds the (share of the) input
    in[0].data()[i] = shared ? (image.data()[i] + t - in[1].data()[i]) % t : 0;
  }
  std::vector<Tensor<u64>> filters(M, Tensor<u64>(meta.fshape));
  for (auto &f : filters) {
    small(f, rng, t, 11);
    f.data()[0] |= 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  P.both([&](int s) { P.cl[s]->conv2d(in[s], s == 0 ? filters : std::vector<Tensor<u64>>(M, Tensor<u64>(meta.fshape)), meta, out[s]); });
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // cleartext convolution (Tensor::Conv2D, tensor.h:320-362, reduced mod t)
  int64_t worst = 0;
  bool ok = out[0].shape().IsSameSize(out[1].shape()) && out[0].channels() == (int64_t)M;
  for (size_t m = 0; ok && m < M; ++m) {
    Tensor<u64> one;
    image.Conv2D(filters[m], (int)stride, pad, one);
    ok = one.height() == out[0].height() && one.width() == out[0].width();
    for (int64_t h = 0; ok && h < one.height(); ++h)
      for (int64_t w = 0; w < one.width(); ++w) {
        const u64 got = (out[0](m, h, w) + out[1](m, h, w)) % t, want = one(0, h, w) % t;
        worst = std::max<int64_t>(worst, std::llabs(to_signed((got + t - want) % t, t)));
      }
  }
  std::printf("   %s: %.1f ms both parties, max |err| = %lld\n", name, ms, (long long)worst);
  EXPECT(ok && worst <= 1, "CheetahLinear::conv2d (reference driver, unchanged) reconstructs within one unit");
}

int main() {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);
  const u64 t = 1ull << 37;
  const int port = 21000 + (int)(getpid() % 20000);
  std::mt19937_64 rng(0x4D264D);
  Pair P;
  try {
    P.both([&](int s) {
      P.io[s] = new sci::NetIO(s == 0 ? nullptr : "127.0.0.1", port);
      P.cl[s] = new CheetahLinear(s == 0 ? sci::ALICE : sci::BOB, P.io[s], t, 2);
    });
    const u64 p = sci::default_prime_mod.at(37);
    EXPECT(P.cl[0]->party() == sci::ALICE && P.cl[1]->party() == sci::BOB, "two CheetahLinear parties over sci::NetIO, keys exchanged");

    conv(P, t, TensorShape({4, 12, 12}), 3, 3, 3, Padding::SAME, 1, true, rng, "4x12x12 3x3 SAME -> 3");
    conv(P, t, TensorShape({3, 70, 70}), 3, 3, 2, Padding::VALID, 2, false, rng, "3x70x70 3x3 stride 2 -> 2 (client's plain input)");
    conv(P, t, TensorShape({64, 56, 56}), 3, 3, 64, Padding::SAME, 1, true, rng, "BASELINE configs[2]: 64x56x56 3x3 SAME -> 64");
#ifdef MM_DROPIN
    // (the batched classes: once more, warm -- plans, device pools and page-locked staging exist by now)
    conv(P, t, TensorShape({64, 56, 56}), 3, 3, 64, Padding::SAME, 1, true, rng, "BASELINE configs[2] again, warm");
    conv(P, t, TensorShape({64, 56, 56}), 3, 3, 64, Padding::SAME, 1, true, rng, "BASELINE configs[2] a third time");
#endif

    {  // fc (cheetah-api.cpp:298-375)
      const int64_t rows = 101, cols = 2048;
      CheetahLinear::FCMeta meta;
      meta.input_shape = TensorShape({cols});
      meta.weight_shape = TensorShape({rows, cols});
      meta.is_shared_input = true;
      Tensor<u64> x(meta.input_shape), in[2] = {Tensor<u64>(meta.input_shape), Tensor<u64>(meta.input_shape)}, W(meta.weight_shape), out[2];
      small(x, rng, t, 11), small(W, rng, t, 11);
      for (int64_t i = 0; i < cols; ++i) in[1].data()[i] = rng() % t, in[0].data()[i] = (x.data()[i] + t - in[1].data()[i]) % t;
      P.both([&](int s) { P.cl[s]->fc(in[s], W, meta, out[s]); });
      int64_t worst = 0;
      for (int64_t r = 0; r < rows; ++r) {
        unsigned __int128 acc = 0;
        for (int64_t c = 0; c < cols; ++c) acc += (unsigned __int128)W.data()[r * cols + c] * x.data()[c] % t;
        worst = std::max<int64_t>(worst, std::llabs(to_signed(((out[0].data()[r] + out[1].data()[r]) % t + t - (u64)(acc % t)) % t, t)));
      }
      EXPECT(worst <= 1, "CheetahLinear::fc reconstructs W x within one unit");
    }
    {  // bn and BOLE over the field (cheetah-api.cpp:527-650)
      const int64_t n = 10000;
      CheetahLinear::BNMeta meta;
      meta.vec_shape = TensorShape({n});
      meta.target_base_mod = p;
      meta.is_shared_input = true;
      Tensor<u64> x(meta.vec_shape), y(meta.vec_shape), xs[2] = {Tensor<u64>(meta.vec_shape), Tensor<u64>(meta.vec_shape)},
                  ys[2] = {Tensor<u64>(meta.vec_shape), Tensor<u64>(meta.vec_shape)}, out[2];
      small(x, rng, p, 11), small(y, rng, p, 11);
      for (int64_t i = 0; i < n; ++i) {
        xs[1].data()[i] = rng() % p, xs[0].data()[i] = (x.data()[i] + p - xs[1].data()[i]) % p;
        ys[1].data()[i] = rng() % p, ys[0].data()[i] = (y.data()[i] + p - ys[1].data()[i]) % p;
      }
      P.both([&](int s) { P.cl[s]->bn(xs[s], y, meta, out[s]); });
      bool ok = true;
      for (int64_t i = 0; ok && i < n; ++i) ok = (out[0].data()[i] + out[1].data()[i]) % p == (u64)((unsigned __int128)x.data()[i] * y.data()[i] % p);
      EXPECT(ok, "CheetahLinear::bn reconstructs x * scale mod p exactly");
      P.both([&](int s) { P.cl[s]->BOLE(xs[s], ys[s], meta, out[s]); });
      ok = true;
      for (int64_t i = 0; ok && i < n; ++i) ok = (out[0].data()[i] + out[1].data()[i]) % p == (u64)((unsigned __int128)x.data()[i] * y.data()[i] % p);
      EXPECT(ok, "CheetahLinear::BOLE reconstructs x * y mod p exactly (tests/bole_test.cpp:164-167)");
    }
    {  // bn_direct (cheetah-api.cpp:652-706)
      CheetahLinear::BNMeta meta;
      meta.ishape = TensorShape({3, 20, 30});
      meta.target_base_mod = t;
      meta.is_shared_input = true;
      Tensor<u64> img(meta.ishape), in[2] = {Tensor<u64>(meta.ishape), Tensor<u64>(meta.ishape)}, scales(TensorShape({3})), out[2];
      small(img, rng, t, 11);
      for (int64_t i = 0; i < img.NumElements(); ++i) in[1].data()[i] = rng() % t, in[0].data()[i] = (img.data()[i] + t - in[1].data()[i]) % t;
      scales.data()[0] = 3, scales.data()[1] = (u64)(-11ll), scales.data()[2] = 4095;
      P.both([&](int s) { P.cl[s]->bn_direct(in[s], scales, meta, out[s]); });
      bool ok = true;
      for (int64_t c = 0; ok && c < 3; ++c)
        for (int64_t i = 0; ok && i < 600; ++i) {
          const int64_t s = (int64_t)scales.data()[c];
          const u64 v = img.data()[c * 600 + i];
          const u64 want = s >= 0 ? (u64)((unsigned __int128)v * (u64)s % t) : (t - (u64)((unsigned __int128)v * (u64)(-s) % t)) % t;
          ok = (out[0].data()[c * 600 + i] + out[1].data()[c * 600 + i]) % t == want;
        }
      EXPECT(ok, "CheetahLinear::bn_direct reconstructs the channel-wise product exactly");
    }
    std::printf("INFO bytes sent: alice %llu bob %llu\n", (unsigned long long)P.cl[0]->io_counter(), (unsigned long long)P.cl[1]->io_counter());
  } catch (const std::exception &e) {
    std::printf("FAIL exception: %s\n", e.what());
    ++g_fail;
  }
  std::printf("%s (%d failures)\n", g_fail ? "FAILED" : "ALL PASSED", g_fail);
  std::fflush(stdout);
  _exit(g_fail ? 1 : 0);  // the sockets and the library's device state go with the process
}
