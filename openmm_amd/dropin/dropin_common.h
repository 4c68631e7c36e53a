// Shared plumbing of the batched implementations of gemini::HomConv2DSS / HomFCSS / HomBNSS that sit behind the
// REFERENCE'S OWN headers (include/gemini/cheetah/hom_{conv2d,fc,bn}_ss.h, compiled unchanged from /root/reference):
// these translation units define the member functions those headers declare, on the seal:: facade types
// (include/seal/seal.h, SEAL's data layout), and make one libmm_hip layer call per method where the reference's .cc
// files loop over ciphertexts.  A maintainer links them in place of include/gemini/cheetah/hom_*_ss.cc
// (INTEGRATION.md 6); SCI/src/cheetah/cheetah-api.cpp and everything above it stay as they are.
#pragma once
#include <seal/seal.h>

#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "gemini/core/types.h"
#include "mm_hip.h"

namespace mm_dropin {
using u64 = std::uint64_t;
using ::Code;  // (the reference declares it at global scope, include/gemini/core/types.h:30-40)

#define DI_TRY(expr)                                  \
  do {                                                \
    const int rc_ = (expr);                           \
    if (rc_ != MMH_OK) return static_cast<Code>(rc_); \
  } while (0)
#define DI_ENSURE(cond, code) \
  do {                        \
    if (!(cond)) return code; \
  } while (0)

// MM_DROPIN_TRACE=1: wall time of every method of the batched classes on stderr (where a layer call's host-to-host time
// goes: seed expansion, copies into and out of the seal:: objects, the device call)
struct Trace {
  const char *what;
  std::chrono::steady_clock::time_point t0;
  static bool on() {
    static const bool v = std::getenv("MM_DROPIN_TRACE") != nullptr;
    return v;
  }
  explicit Trace(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {}
  void lap(const char *step) {
    if (!on()) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[dropin] %s / %s: %.3f ms\n", what, step, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// Device buffers recycled by size (hipMalloc / hipFree cost 0.1-0.3 ms each and a layer call needs a dozen).  All work
// of one Fast object runs on its one stream, which orders a buffer's next use after its last.
class DevPool {
 public:
  ~DevPool() { release(); }
  u64 *get(std::size_t words) {
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = free_.find(words);
      if (it != free_.end()) {
        u64 *p = it->second;
        free_.erase(it);
        return p;
      }
    }
    void *q = nullptr;
    if (mmh_malloc(&q, (words ? words : 1) * 8) != MMH_OK) {
      release();
      if (mmh_malloc(&q, (words ? words : 1) * 8) != MMH_OK) return nullptr;
    }
    return static_cast<u64 *>(q);
  }
  void put(u64 *p, std::size_t words) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    free_.emplace(words, p);
  }
  void release() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto &kv : free_) (void)mmh_free(kv.second);
    free_.clear();
  }

 private:
  std::mutex mu_;
  std::multimap<std::size_t, u64 *> free_;
};

struct Dev {  // device buffer of 64-bit words, from and back to a pool
  DevPool *pool = nullptr;
  u64 *p = nullptr;
  std::size_t n = 0;
  Dev() = default;
  Dev(DevPool &pl, std::size_t words) : pool(&pl), p(pl.get(words)), n(words) {}
  Dev(const Dev &) = delete;
  Dev(Dev &&o) noexcept : pool(o.pool), p(o.p), n(o.n) { o.p = nullptr; }
  Dev &operator=(Dev &&o) noexcept {
    if (this != &o) {
      if (p) pool->put(p, n);
      pool = o.pool, p = o.p, n = o.n, o.p = nullptr;
    }
    return *this;
  }
  ~Dev() {
    if (p) pool->put(p, n);
  }
  explicit operator bool() const { return p != nullptr; }
};

// What a protocol object keeps on the device: hung on the seal::Evaluator it owns (Evaluator::facade_ext), so it lives
// exactly as long as the reference's object does.
struct Fast {
  std::shared_ptr<seal::facade::Backend> backend;
  mmh_ctx *ctx = nullptr;
  std::size_t N = 0, L = 0;
  void *stream = nullptr;
  DevPool pool;
  Dev sk, sk_last, pk, pk_last;  // secret key [L][N] / its last-level restriction [N]; public key [2][L][N] / [2][N]
  std::shared_ptr<seal::UniformRandomGenerator> prng;  // from parms.random_generator(): a test can inject it
  std::recursive_mutex mu;                             // one layer call at a time per object
  // page-locked staging (grow-only): ciphertext batches cross the link from / to one block at the link's rate
  u64 *staging = nullptr;
  std::size_t staging_words = 0;

  ~Fast() {
    if (stream) (void)mmh_stream_sync(stream);
    sk = Dev(), sk_last = Dev(), pk = Dev(), pk_last = Dev();
    pool.release();
    if (staging) (void)mmh_host_free(staging);
    if (stream) (void)mmh_stream_destroy(stream);
  }
  Dev mk(std::size_t words) { return Dev(pool, words); }
  int up(const Dev &d, const void *src, std::size_t words, std::size_t at = 0) {
    return words ? mmh_memcpy_h2d(d.p + at, src, words * 8, stream) : MMH_OK;
  }
  int down(void *dst, const Dev &d, std::size_t words, std::size_t at = 0) {
    return words ? mmh_memcpy_d2h(dst, d.p + at, words * 8, stream) : MMH_OK;
  }
  int sync() { return mmh_stream_sync(stream); }
  u64 *stage(std::size_t words) {
    if (staging_words < words) {
      (void)mmh_stream_sync(stream);
      if (staging) (void)mmh_host_free(staging);
      staging = nullptr, staging_words = 0;
      void *q = nullptr;
      if (mmh_host_alloc(&q, words * 8) != MMH_OK) return nullptr;
      staging = static_cast<u64 *>(q), staging_words = words;
    }
    return staging;
  }
  // key + `count` consecutive stream ids for one device sampler launch
  std::uint64_t streams(std::uint32_t key[8], std::uint64_t count) { return prng->take_streams(key, count); }

  // stream, sampler and device copies of whichever keys the party holds (hom_conv2d_ss.cc:467-495 and its siblings)
  Code init(const seal::SEALContext &context, const seal::SecretKey *secret, const seal::PublicKey *pub) {
    auto cd = context.first_context_data();
    backend = cd->backend();
    ctx = backend->ctx, N = backend->N, L = backend->L;
    prng = cd->parms().random_generator()->create();
    DI_TRY(mmh_stream_create(&stream));
    if (secret) {
      sk = mk(L * N), sk_last = mk(N);
      DI_ENSURE(sk && sk_last, Code::ERR_SEAL_MEMORY);
      DI_TRY(up(sk, secret->data().data(), L * N));
      DI_TRY(up(sk_last, secret->data().data(), N));
    }
    if (pub) {
      pk = mk(2 * L * N), pk_last = mk(2 * N);
      DI_ENSURE(pk && pk_last, Code::ERR_SEAL_MEMORY);
      const u64 *src = pub->data().data();
      DI_TRY(up(pk, src, 2 * L * N));
      DI_TRY(up(pk_last, src, N));  // the last level keeps limb 0 of both polynomials
      DI_TRY(up(pk_last, src + L * N, N, N));
    }
    DI_TRY(sync());
    return Code::OK;
  }
};

// fn(i) for i in [0, n) on up to `nthreads` host threads (the reference's ThreadPool + LaunchWorks,
// hom_conv2d_ss.cc:31-57, for the host-side work that is left: seed expansion, copies into the objects)
template <class Fn>
void parallel_for(std::size_t n, std::size_t nthreads, Fn &&fn) {
  nthreads = std::min<std::size_t>(std::max<std::size_t>(1, nthreads), 16);
  if (nthreads <= 1 || n <= 1) {
    for (std::size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  const std::size_t per = (n + nthreads - 1) / nthreads;
  std::vector<std::thread> workers;
  for (std::size_t w = 0; w < nthreads; ++w) {
    const std::size_t lo = std::min(w * per, n), hi = std::min(lo + per, n);
    if (lo == hi) break;
    workers.emplace_back([lo, hi, &fn] {
      for (std::size_t i = lo; i < hi; ++i) fn(i);
    });
  }
  for (auto &t : workers) t.join();
}

// A batch of coefficient-form plaintexts [count][N] that stays on the device (facade::LazyPlain).  `encode`, when set,
// fills `pts` the first time a host copy of an element is asked for (a layer whose plaintexts are normally never
// built: the filters of a convolution travel as the raw tensor).
struct PlainBatch : seal::facade::LazyPlain {
  std::shared_ptr<Fast> fast;
  int kind = 0;                   // who made it (kKind* below): the consumer checks before taking the short cut
  std::array<std::int64_t, 10> tag{};  // the layer geometry it was made for
  std::size_t count = 0, N = 0;
  mutable Dev pts;                // [count][N] coefficient form (may be empty until `encode` ran)
  Dev raw;                        // the tensor the batch was encoded from, as the layer call reads it
  std::vector<std::uint8_t> zero;  // per element: all coefficients zero
  mutable std::function<int(const PlainBatch &)> encode;
  mutable std::mutex mu;

  std::size_t coeff_count(std::size_t) const override { return N; }
  bool is_zero(std::size_t i) const override {
    if (!zero.empty()) return i < zero.size() && zero[i];
    std::vector<std::uint64_t> v;   // (not recorded when the batch was made: look)
    materialize(i, v);
    return std::all_of(v.begin(), v.end(), [](std::uint64_t c) { return c == 0; });
  }
  void materialize(std::size_t i, std::vector<std::uint64_t> &out) const override {
    std::lock_guard<std::mutex> g(mu);
    std::lock_guard<std::recursive_mutex> g2(fast->mu);
    if (!pts) {
      if (!encode || encode(*this) != MMH_OK || !pts) throw std::logic_error("lazy plaintext batch: encoding failed");
    }
    out.assign(N, 0);
    seal::facade::check(fast->down(out.data(), pts, N, i * N), "lazy plaintext: d2h");
    seal::facade::check(fast->sync(), "lazy plaintext: sync");
  }
};
enum { kKindConvImage = 1, kKindConvFilters, kKindFcInput, kKindFcWeights, kKindBnVector, kKindBnTensor };

// Does `v` hold exactly the elements first .. first + n - 1 of ONE batch of the given kind and tag?
inline const PlainBatch *whole_batch(const seal::Plaintext *v, std::size_t n, int kind,
                                     const std::array<std::int64_t, 10> &tag, std::size_t first = 0) {
  if (n == 0) return nullptr;
  const auto *b = dynamic_cast<const PlainBatch *>(v[0].facade_lazy().get());
  if (!b || b->kind != kind || b->tag != tag) return nullptr;
  for (std::size_t i = 0; i < n; ++i)
    if (v[i].facade_lazy().get() != b || v[i].facade_lazy_index() != first + i) return nullptr;
  return b;
}

// host ciphertexts (the vectors of `parts`, one after the other) -> one device array [n][2][limbs][N]; every
// ciphertext must have two polynomials of `limbs` limbs in coefficient form
inline Code upload_cts(Fast &f, std::initializer_list<const std::vector<seal::Ciphertext> *> parts, std::size_t limbs,
                       Dev &out, std::size_t nthreads = 4) {
  const std::size_t words = 2 * limbs * f.N;
  std::size_t n = 0;
  for (const auto *v : parts) {
    // fresh ciphertexts arrive with c1 as a seed (Ciphertext::load leaves it so): expand them side by side
    if (!v->empty() && (*v)[0].facade_pending())
      parallel_for(v->size(), nthreads, [&](std::size_t i) { (*v)[i].facade_expand_now(); });
    n += v->size();
    for (const auto &c : *v)
      DI_ENSURE(c.size() == 2 && c.coeff_modulus_size() == limbs && c.poly_modulus_degree() == f.N && !c.is_ntt_form(),
                Code::ERR_INVALID_ARG);
  }
  out = f.mk(n * words);
  DI_ENSURE(out, Code::ERR_SEAL_MEMORY);
  u64 *host = f.stage(n * words);
  DI_ENSURE(host, Code::ERR_SEAL_MEMORY);
  std::size_t at = 0;
  for (const auto *v : parts) {
    u64 *dst = host + at * words;
    parallel_for(v->size(), 8, [&](std::size_t i) { std::memcpy(dst + i * words, (*v)[i].data(), words * 8); });
    at += v->size();
  }
  DI_TRY(f.up(out, host, n * words));
  return Code::OK;
}
inline Code upload_cts(Fast &f, const std::vector<seal::Ciphertext> &cts, std::size_t limbs, Dev &out,
                       std::size_t nthreads = 4) {
  return upload_cts(f, {&cts}, limbs, out, nthreads);
}
// device array [n][2][1][N] at the last level -> host ciphertexts (enqueue + wait)
inline Code download_last_level(Fast &f, const seal::SEALContext &context, const Dev &d, std::size_t n,
                                std::vector<seal::Ciphertext> &out) {
  const std::size_t words = 2 * f.N;
  u64 *host = f.stage(n * words);
  DI_ENSURE(host, Code::ERR_SEAL_MEMORY);
  DI_TRY(f.down(host, d, n * words));
  out.resize(n);
  for (auto &c : out) {   // (sizes the objects while the copy runs)
    c.resize(context, context.last_parms_id(), 2);
    c.is_ntt_form() = false, c.scale() = 1.0;
  }
  DI_TRY(f.sync());
  parallel_for(n, 8, [&](std::size_t i) { std::memcpy(out[i].data(), host + i * words, words * 8); });
  return Code::OK;
}
// Fresh symmetric ciphertexts in their Serializable (seeded) form: c1 of each is expanded on the host from a fresh
// 64-byte seed exactly as a SEAL loader expands it again (wire::Blake2xbPRNG + sample_poly_uniform), on `nthreads`
// threads, and handed to the batched encryption as its `a`.  Returns the device array [n][L][N] and the seeds.
inline Code fresh_a(Fast &f, const seal::SEALContext &context, std::size_t n, std::size_t nthreads, Dev &a,
                    std::vector<std::array<std::uint8_t, seal::wire::kSeedBytes>> &seeds) {
  const std::size_t words = f.L * f.N;
  std::vector<u64> q;
  for (const auto &m : context.first_context_data()->parms().coeff_modulus()) q.push_back(m.value());
  seeds.resize(n);
  f.prng->generate(n * seal::wire::kSeedBytes, reinterpret_cast<seal::seal_byte *>(seeds.data()));
  a = f.mk(n * words);
  DI_ENSURE(a, Code::ERR_SEAL_MEMORY);
  u64 *host = f.stage(n * words);
  DI_ENSURE(host, Code::ERR_SEAL_MEMORY);
  parallel_for(n, nthreads, [&](std::size_t i) {
    std::uint8_t seed[seal::wire::kSeedBytes];
    std::memcpy(seed, seeds[i].data(), sizeof(seed));
    seal::wire::Blake2xbPRNG prng(seed);
    seal::wire::sample_poly_uniform(prng, q, f.N, host + i * words);
  });
  DI_TRY(f.up(a, host, n * words));
  DI_TRY(f.sync());  // the staging block is reused by the download that follows
  return Code::OK;
}
// device array of fresh first-level ciphertexts [n][2][L][N] -> Serializable objects carrying their seeds
inline Code download_fresh(Fast &f, const seal::SEALContext &context, const Dev &d, std::size_t n,
                           const std::vector<std::array<std::uint8_t, seal::wire::kSeedBytes>> &seeds,
                           std::vector<seal::Serializable<seal::Ciphertext>> &out) {
  const std::size_t words = 2 * f.L * f.N;
  u64 *host = f.stage(n * words);
  DI_ENSURE(host, Code::ERR_SEAL_MEMORY);
  DI_TRY(f.down(host, d, n * words));
  std::vector<seal::Ciphertext> cts(n);
  for (auto &c : cts) c.resize(context, context.first_parms_id(), 2);
  DI_TRY(f.sync());
  parallel_for(n, 8, [&](std::size_t i) {
    std::memcpy(cts[i].data(), host + i * words, words * 8);
    std::uint8_t seed[seal::wire::kSeedBytes];
    std::memcpy(seed, seeds[i].data(), sizeof(seed));
    cts[i].facade_set_seed(seed);
  });
  out.clear();
  out.reserve(n);
  for (auto &c : cts) out.emplace_back(std::move(c));
  return Code::OK;
}
}  // namespace mm_dropin
