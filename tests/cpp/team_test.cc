// ALICE's server side cut over several devices (CheetahLinear::Options::devices) against the one-device run.
// Both parties run in one process over LoopbackIO with fixed sampler keys; for every layer the program compares, between
// the device lists {0}, {0,0} and {0,0,0} (workers sharing one GPU: each has its own stream, contexts, plans and pool):
//   * the bytes ALICE puts on the wire (a hash of everything she sends),
//   * both parties' output shares, word for word,
// and checks the reconstruction of the FC and BN layers against cleartext arithmetic (the convolutions are checked by
// loopback_test, which tests/test_cpp_host_gpu.py also runs with MM_GEMINI_DEVICES=0,0).
// The cuts follow LaunchWorks (hom_conv2d_ss.cc:597-607): filter counts that do not divide by the device count, and
// fewer units than devices, are among the cases.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <thread>

#include "mm_gemini/cheetah_linear.h"

using gemini::CheetahLinear;
using gemini::Padding;
using gemini::Tensor;
using gemini::TensorShape;
using u64 = uint64_t;

static int g_fail = 0;
#define EXPECT(cond, name)                                  \
  do {                                                      \
    if (cond) std::printf("PASS %s\n", name);               \
    else { std::printf("FAIL %s\n", name); ++g_fail; }      \
  } while (0)

// a LoopbackIO end that hashes what its party sends (FNV-1a over the bytes, in order)
class HashIO : public mm::IOChannel {
 public:
  explicit HashIO(std::unique_ptr<mm::LoopbackIO> io) : io_(std::move(io)) {}
  void send_data(const void *d, size_t n) override {
    const uint8_t *b = static_cast<const uint8_t *>(d);
    for (size_t i = 0; i < n; ++i) hash_ = (hash_ ^ b[i]) * 0x100000001B3ULL;
    counter += n;
    io_->send_data(d, n);
  }
  void recv_data(void *d, size_t n) override { io_->recv_data(d, n); }
  void flush() override { io_->flush(); }
  u64 take() {  // hash of the bytes sent since the last take()
    const u64 h = hash_;
    hash_ = 0xCBF29CE484222325ULL;
    return h;
  }

 private:
  std::unique_ptr<mm::LoopbackIO> io_;
  u64 hash_ = 0xCBF29CE484222325ULL;
};

struct Record {  // what one run of the layer list left behind
  std::vector<u64> wire;                  // ALICE's bytes per layer
  std::vector<std::vector<u64>> out[2];   // both parties' outputs per layer
  size_t alice_devices = 0;
};

static const int kBits = 37;
static const u64 kMask = (1ULL << kBits) - 1;

static Record run(const std::vector<int> &devices) {
  auto ios = mm::LoopbackIO::MakePair();
  HashIO io_a(std::move(ios.first)), io_b(std::move(ios.second));
  HashIO *io[2] = {&io_a, &io_b};
  std::unique_ptr<CheetahLinear> lin[2];
  const uint32_t key_a[8] = {1, 2, 3, 4, 5, 6, 7, 8}, key_b[8] = {11, 12, 13, 14, 15, 16, 17, 18};
  auto both = [&](const std::function<void(int)> &fn) {
    std::exception_ptr e[2];
    std::thread t0([&] { try { fn(0); } catch (...) { e[0] = std::current_exception(); } });
    std::thread t1([&] { try { fn(1); } catch (...) { e[1] = std::current_exception(); } });
    t0.join();
    t1.join();
    for (auto &x : e)
      if (x) std::rethrow_exception(x);
  };
  both([&](int s) {
    CheetahLinear::Options opt;
    opt.devices = devices;
    opt.sampler_key = s == 0 ? key_a : key_b;
    lin[s].reset(new CheetahLinear(s + 1, io[s], 1ULL << kBits, opt));
  });
  Record rec;
  rec.alice_devices = lin[0]->n_devices();
  io_a.take();
  std::mt19937_64 rng(0x7EA3);
  auto rnd = [&](size_t n) {
    std::vector<u64> v(n);
    for (auto &x : v) x = rng() & kMask;
    return v;
  };
  auto done = [&](std::vector<u64> a, std::vector<u64> b) {
    rec.wire.push_back(io_a.take());
    rec.out[0].push_back(std::move(a));
    rec.out[1].push_back(std::move(b));
  };

  // ---- conv2d on tensors: (C, H, f, M, stride, same) ----
  const long convs[][6] = {{64, 14, 3, 10, 1, 1}, {32, 28, 1, 7, 2, 0}, {16, 9, 3, 2, 1, 1}, {3, 32, 5, 4, 2, 1}};
  for (const auto &c : convs) {
    CheetahLinear::ConvMeta meta{TensorShape({c[0], c[1], c[1]}), TensorShape({c[0], c[2], c[2]}), (size_t)c[3],
                                 c[5] ? Padding::SAME : Padding::VALID, (size_t)c[4], true};
    Tensor<u64> img[2] = {Tensor<u64>(meta.ishape), Tensor<u64>(meta.ishape)}, out[2];
    std::vector<Tensor<u64>> filters(meta.n_filters, Tensor<u64>(meta.fshape));
    for (int s = 0; s < 2; ++s) {
      auto v = rnd((size_t)meta.ishape.num_elements());
      std::copy(v.begin(), v.end(), img[s].data());
    }
    for (auto &f : filters) {
      auto v = rnd((size_t)meta.fshape.num_elements());
      std::copy(v.begin(), v.end(), f.data());
    }
    both([&](int s) { lin[s]->conv2d(img[s], s == 0 ? filters : std::vector<Tensor<u64>>(meta.n_filters, Tensor<u64>(meta.fshape)), meta, out[s]); });
    done(std::vector<u64>(out[0].data(), out[0].data() + out[0].NumElements()),
         std::vector<u64>(out[1].data(), out[1].data() + out[1].NumElements()));
  }
  // ---- conv2d_nhwc (Conv2DWrapper's layouts, raw 64-bit values): (C, H, f, M, stride) ----
  const long nhwc[][5] = {{256, 14, 1, 66, 1}, {64, 14, 3, 5, 2}};
  for (const auto &c : nhwc) {
    CheetahLinear::ConvMeta meta{TensorShape({c[0], c[1], c[1]}), TensorShape({c[0], c[2], c[2]}), (size_t)c[3],
                                 Padding::SAME, (size_t)c[4], true};
    const size_t oh = (size_t)((c[1] + c[4] - 1) / c[4]);
    std::vector<u64> in[2], filt((size_t)(c[2] * c[2] * c[0] * c[3])), out[2];
    for (int s = 0; s < 2; ++s) {
      in[s].resize((size_t)(c[0] * c[1] * c[1]));
      for (auto &x : in[s]) x = rng();  // unreduced, as the networks' arrays are
      out[s].assign(oh * oh * (size_t)c[3], 0);
    }
    for (auto &x : filt) x = rng();
    both([&](int s) { lin[s]->conv2d_nhwc(in[s].data(), s == 0 ? filt.data() : nullptr, meta, out[s].data(), (int64_t)oh, (int64_t)oh); });
    done(out[0], out[1]);
  }
  // ---- fc, both weight orientations: (rows, cols) ----
  const long fcs[][2] = {{1000, 512}, {10, 4096}, {4100, 3}, {1, 64}};
  for (const auto &c : fcs)
    for (int in_out = 0; in_out < 2; ++in_out) {
      const size_t rows = (size_t)c[0], cols = (size_t)c[1];
      CheetahLinear::FCMeta meta{TensorShape({c[1]}), TensorShape({c[0], c[1]}), true};
      Tensor<u64> x[2] = {Tensor<u64>(meta.input_shape), Tensor<u64>(meta.input_shape)}, out[2];
      Tensor<u64> w(meta.weight_shape);
      std::vector<u64> wt(rows * cols);
      for (int s = 0; s < 2; ++s) {
        auto v = rnd(cols);
        std::copy(v.begin(), v.end(), x[s].data());
      }
      for (size_t i = 0; i < rows * cols; ++i) w.data()[i] = (rng() & kMask) | 1;
      for (size_t r = 0; r < rows; ++r)
        for (size_t k = 0; k < cols; ++k) wt[k * rows + r] = w.data()[r * cols + k];
      both([&](int s) {
        if (in_out) lin[s]->fc_in_out(x[s], s == 0 ? wt.data() : nullptr, meta, out[s]);
        else lin[s]->fc(x[s], w, meta, out[s]);
      });
      size_t bad = 0;
      for (size_t r = 0; r < rows; ++r) {
        u64 acc = 0;
        for (size_t k = 0; k < cols; ++k) acc += w.data()[r * cols + k] * ((x[0].data()[k] + x[1].data()[k]) & kMask);
        const u64 got = (out[0].data()[r] + out[1].data()[r]) & kMask;
        const u64 d = (got - acc) & kMask;
        bad += !(d == 0 || d == 1 || d == kMask);  // the approximate re-sharing is off by at most one unit
      }
      char name[96];
      std::snprintf(name, sizeof name, "fc %zux%zu%s reconstructs on %zu device(s)", rows, cols, in_out ? " (in x out)" : "",
                    rec.alice_devices);
      EXPECT(bad == 0, name);
      done(std::vector<u64>(out[0].data(), out[0].data() + rows), std::vector<u64>(out[1].data(), out[1].data() + rows));
    }
  // ---- bn (SIMD path over the field prime): lengths around the ciphertext size ----
  const u64 p = lin[0]->field_mod();
  for (size_t n : {(size_t)100, (size_t)4096, (size_t)3 * 4096 + 100, (size_t)9 * 4096}) {
    CheetahLinear::BNMeta meta{TensorShape({0, 0, 0}), TensorShape({(int64_t)n}), p, true};
    Tensor<u64> x[2] = {Tensor<u64>(meta.vec_shape), Tensor<u64>(meta.vec_shape)}, scale(meta.vec_shape), out[2];
    for (int s = 0; s < 2; ++s)
      for (size_t i = 0; i < n; ++i) x[s].data()[i] = rng() % p;
    for (size_t i = 0; i < n; ++i) scale.data()[i] = rng() % p;
    both([&](int s) { lin[s]->bn(x[s], scale, meta, out[s]); });
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
      const u64 xi = (x[0].data()[i] + x[1].data()[i]) % p;
      const u64 want = (u64)((unsigned __int128)xi * scale.data()[i] % p);
      bad += (out[0].data()[i] + out[1].data()[i]) % p != want;
    }
    char name[96];
    std::snprintf(name, sizeof name, "bn of %zu elements reconstructs on %zu device(s)", n, rec.alice_devices);
    EXPECT(bad == 0, name);
    done(std::vector<u64>(out[0].data(), out[0].data() + n), std::vector<u64>(out[1].data(), out[1].data() + n));
  }
  return rec;
}

int main() {
  const Record one = run({0});
  EXPECT(one.alice_devices == 1, "one device listed, one used");
  for (const std::vector<int> &devices : {std::vector<int>{0, 0}, std::vector<int>{0, 0, 0}}) {
    const Record many = run(devices);
    char name[128];
    std::snprintf(name, sizeof name, "%zu workers made for %zu listed devices", many.alice_devices, devices.size());
    EXPECT(many.alice_devices == devices.size(), name);
    size_t wire_diff = 0, out_diff = 0;
    for (size_t i = 0; i < one.wire.size(); ++i) {
      wire_diff += one.wire[i] != many.wire[i];
      out_diff += one.out[0][i] != many.out[0][i] || one.out[1][i] != many.out[1][i];
    }
    std::snprintf(name, sizeof name, "%zu layers: ALICE's bytes on the wire equal the one-device run's (%zu devices)",
                  one.wire.size(), devices.size());
    EXPECT(many.wire.size() == one.wire.size() && wire_diff == 0, name);
    std::snprintf(name, sizeof name, "%zu layers: both parties' shares equal the one-device run's (%zu devices)",
                  one.wire.size(), devices.size());
    EXPECT(out_diff == 0, name);
  }
  std::printf(g_fail ? "team_test: %d FAILED\n" : "team_test: all passed\n", g_fail);
  return g_fail ? 1 : 0;
}
