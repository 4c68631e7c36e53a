// gemini::CheetahLinear on libmm_hip (C ABI only: no HIP, no SEAL in this file).
// Choreography follows SCI/src/cheetah/cheetah-api.cpp: keys and pk exchange :73-197, fc :298-375,
// conv2d :377-446, bn :527-603, BOLE :605-650.
#include "mm_gemini/cheetah_linear.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <iterator>
#include <map>
#include <mutex>

#include "mm_hip.h"

namespace gemini {

std::string CodeMessage(Code code) {  // include/gemini/core/types.cc:3-24
  switch (code) {
    case Code::OK: return "ok";
    case Code::ERR_CONFIG: return "Error: configuration";
    case Code::ERR_NULL_POINTER: return "Error: null pointer";
    case Code::ERR_DIM_MISMATCH: return "Error: dimension mismatch";
    case Code::ERR_SEAL_MEMORY: return "Error: memory allocation";
    case Code::ERR_OUT_BOUND: return "Error: out-of-bound";
    case Code::ERR_INVALID_ARG: return "Error: invalid arguments";
    case Code::ERR_INTERNAL: return "Error: internal error";
    default: return "Unknown code";
  }
}

namespace {
const int kAlice = 1, kBob = 2;
// SCI/src/utils/constants.h:45-50
const std::map<int, uint64_t> kFieldPrimes{{32, 4293918721ULL},   {33, 8585084929ULL},   {34, 17171218433ULL},
                                           {35, 34359214081ULL},  {36, 68686184449ULL},  {37, 137438822401ULL},
                                           {38, 274824036353ULL}, {39, 549753716737ULL}, {40, 1099480956929ULL},
                                           {41, 2198100901889ULL}};

void must(int rc, const char *what) {
  if (rc != MMH_OK)
    throw std::runtime_error(std::string(what) + " [" + CodeMessage(static_cast<Code>(rc)) + "] " + mmh_last_error());
}

// Device buffers are recycled by size, per object (hence per device): a layer call needs a dozen of them and
// hipMalloc / hipFree cost 0.1-0.3 ms each.  All work of an object runs on its one stream, which orders a buffer's
// next use after its last.  Idle blocks beyond `cap_bytes` go back to the driver (largest first); release() frees
// all idle blocks.
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, uint64_t *> free_;
  size_t idle_bytes = 0, cap_bytes = size_t(2) << 30;
  ~DevPool() { release(); }
  uint64_t *get(size_t words) {
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = free_.find(words);
      if (it != free_.end()) {
        uint64_t *p = it->second;
        free_.erase(it);
        idle_bytes -= words * 8;
        return p;
      }
    }
    void *q = nullptr;
    if (mmh_malloc(&q, (words ? words : 1) * 8) != MMH_OK) {  // give the idle blocks back and try once more
      release();
      must(mmh_malloc(&q, (words ? words : 1) * 8), "mmh_malloc");
    }
    return static_cast<uint64_t *>(q);
  }
  void put(uint64_t *p, size_t words) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu);
    free_.emplace(words, p);
    idle_bytes += words * 8;
    while (idle_bytes > cap_bytes && !free_.empty()) {  // callers sync their stream before buffers die
      auto it = std::prev(free_.end());
      idle_bytes -= it->first * 8;
      (void)mmh_free(it->second);
      free_.erase(it);
    }
  }
  void release() {
    std::lock_guard<std::mutex> g(mu);
    for (auto &kv : free_) (void)mmh_free(kv.second);
    free_.clear();
    idle_bytes = 0;
  }
};

// device buffer of 64-bit words, taken from and returned to its object's pool
struct Dev {
  DevPool *pool = nullptr;
  uint64_t *p = nullptr;
  size_t n = 0;
  Dev() = default;
  Dev(DevPool *pl, size_t words) : pool(pl), p(pl->get(words)), n(words) {}
  Dev(const Dev &) = delete;
  Dev(Dev &&o) noexcept : pool(o.pool), p(o.p), n(o.n) { o.p = nullptr; }
  Dev &operator=(Dev &&o) noexcept {
    if (this != &o) {
      if (p) pool->put(p, n);
      pool = o.pool, p = o.p, n = o.n;
      o.p = nullptr;
    }
    return *this;
  }
  ~Dev() {
    if (p) pool->put(p, n);
  }
};

struct Keys {  // one BFV context with the party's key material on the device
  mmh_ctx *ctx = nullptr;
  Dev sk, pk, pk_last, sk_last;  // sk [L][N] (BOB), pk [2][L][N] (ALICE), restrictions to the last level
};

// the object's device is current inside its methods and the caller's afterwards
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    must(mmh_get_device(&prev), "mmh_get_device");
    if (prev != device) {
      must(mmh_set_device(device), "mmh_set_device");
      switched = true;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)mmh_set_device(prev);
  }
};

// One device's working set: its stream and buffer pool, the two BFV contexts with the key material that device
// needs, and the layer plans made on it.  The object's first device does all of BOB's work and all of ALICE's when
// there is one device; ALICE's further devices each take a range of a layer's output ciphertexts (conv_team /
// fc_team / bn_team below).
struct Worker {
  static constexpr size_t N = 4096, L = 2;
  int device = 0;
  void *stream = nullptr;  // every copy and kernel of this device; one wait before bytes go to the channel
  mutable DevPool pool;
  Keys ring, field;

  // layer geometry is computed once per shape (the reference redoes it on every call)
  mutable std::mutex plan_mu;
  mutable std::map<std::array<int64_t, 9>, mmh_conv_plan *> conv_plans;
  mutable std::map<std::array<int64_t, 5>, mmh_fc_plan *> fc_plans;

  Worker() = default;
  Worker(const Worker &) = delete;
  ~Worker() {
    if (stream) (void)mmh_stream_sync(stream);
    for (auto &kv : conv_plans) mmh_conv_plan_destroy(kv.second);
    for (auto &kv : fc_plans) mmh_fc_plan_destroy(kv.second);
    mmh_ctx_destroy(ring.ctx);
    mmh_ctx_destroy(field.ctx);
    mmh_stream_destroy(stream);
  }  // the key buffers return to the pool and the pool frees its blocks as the members go (hipFree needs no stream)
  Dev mk(size_t words) const { return Dev(&pool, words); }
  void up(const Dev &d, const uint64_t *src, size_t words) const {  // asynchronous; src may be reused on return
    must(mmh_memcpy_h2d(d.p, src, words * 8, stream), "h2d");
  }
  void down(uint64_t *dst, const Dev &d, size_t words) const {  // valid after sync()
    must(mmh_memcpy_d2h(dst, d.p, words * 8, stream), "d2h");
  }
  void sync() const { must(mmh_stream_sync(stream), "sync"); }

  mmh_conv_plan *conv_plan(const mmh_conv_meta &m) const {
    const std::array<int64_t, 9> key{m.in_c, m.in_h, m.in_w, m.f_h, m.f_w, m.n_filters, m.padding_same, m.stride,
                                     m.is_shared_input};
    std::lock_guard<std::mutex> g(plan_mu);
    auto it = conv_plans.find(key);
    if (it != conv_plans.end()) return it->second;
    mmh_conv_plan *plan = nullptr;
    must(mmh_conv_plan_create(ring.ctx, &m, &plan), "CheetahLinear::conv2d shape inference");
    return conv_plans[key] = plan;
  }
  // split: the (d0, d1) of the whole matrix when m is a range of its row blocks, else null (getSplit's own)
  mmh_fc_plan *fc_plan(const mmh_fc_meta &m, const int64_t *split = nullptr) const {
    const std::array<int64_t, 5> key{m.rows, m.cols, m.is_shared_input, split ? split[0] : 0, split ? split[1] : 0};
    std::lock_guard<std::mutex> g(plan_mu);
    auto it = fc_plans.find(key);
    if (it != fc_plans.end()) return it->second;
    mmh_fc_plan *plan = nullptr;
    must(split ? mmh_fc_plan_create_split(ring.ctx, &m, split, &plan) : mmh_fc_plan_create(ring.ctx, &m, &plan),
         "CheetahLinear::fc getSplit");
    return fc_plans[key] = plan;
  }
};
}  // namespace

struct CheetahLinear::Impl : Worker {
  uint32_t key[8];
  bool fixed_key = false;  // Options::sampler_key: ciphertext seeds follow the key as well (tests)
  mutable uint64_t stream_id = 1, seed_count = 0;
  mm::IOChannel *io = nullptr;
  std::vector<std::unique_ptr<Worker>> helpers;  // ALICE's devices after the first
  std::vector<uint64_t> pk_host;                 // BOB's ring public key as received, for the helpers

  ~Impl() {
    if (stream) (void)mmh_stream_sync(stream);
    if (staging.p) mmh_host_free(staging.p);
    if (flag_host) mmh_host_free(flag_host);
  }
  uint64_t sid() const { return stream_id++; }

  // ---- randomness (device ChaCha20 keyed from the OS) ----
  Dev uniform(const mmh_ctx *c, int lv, size_t n) const {
    Dev d = mk(n * (lv == 0 ? L : 1) * N);
    must(mmh_sample_uniform(c, lv, key, sid(), d.p, n, stream), "sample_uniform");
    return d;
  }
  Dev ternary(size_t n) const {
    Dev d = mk(n * N);
    must(mmh_sample_ternary(key, sid(), (int64_t *)d.p, n * N, stream), "sample_ternary");
    return d;
  }
  Dev noise(size_t n) const {
    Dev d = mk(n * N);
    must(mmh_sample_noise(key, sid(), (int64_t *)d.p, n * N, stream), "sample_noise");
    return d;
  }
  // (u, e1, mask) of n re-sharings in one launch (mmh_sample_reshare)
  struct Reshare {
    Dev u, e1, mask;
  };
  Reshare reshare(size_t n) const {
    Reshare r{mk(n * N), mk(n * N), mk(n * N)};
    const uint64_t s0 = stream_id;
    stream_id += 3;
    must(mmh_sample_reshare(ring.ctx, key, s0, n, (int64_t *)r.u.p, (int64_t *)r.e1.p, r.mask.p, stream), "sample_reshare");
    return r;
  }
  Dev words(size_t n) const {
    Dev d = mk(n);
    must(mmh_chacha20_words(key, sid(), 0, d.p, n, stream), "chacha20");
    return d;
  }

  // ---- keys: BOB generates and sends pk, ALICE receives (cheetah-api.cpp:107-141, :164-196) ----
  void setup_keys(Keys &k, int party) {
    std::vector<uint64_t> host(2 * L * N);
    k.pk = mk(2 * L * N);
    k.pk_last = mk(2 * N);
    if (party == kBob) {
      k.sk = mk(L * N);
      k.sk_last = mk(N);
      Dev s = ternary(1), a = uniform(k.ctx, 0, 1), e = noise(1);
      must(mmh_signed_to_ntt(k.ctx, 0, (const int64_t *)s.p, 1, k.sk.p, stream), "keygen");
      must(mmh_make_public_key(k.ctx, k.sk.p, a.p, (const int64_t *)e.p, k.pk.p, stream), "create_public_key");
      std::vector<uint64_t> skh(L * N);
      down(host.data(), k.pk, host.size());
      down(skh.data(), k.sk, skh.size());
      sync();
      up(k.sk_last, skh.data(), N);
      uint64_t sz = host.size() * 8;
      io->send_data(&sz, sizeof(sz));
      io->send_data(host.data(), sz);
      io->flush();
    } else {
      uint64_t sz = 0;
      io->recv_data(&sz, sizeof(sz));
      if (sz != host.size() * 8) throw std::runtime_error("CheetahLinear: public key size mismatch");
      io->recv_data(host.data(), sz);
      up(k.pk, host.data(), host.size());
      if (&k == &ring) pk_host = host;
    }
    // pk restricted to the last level {q0}: limb 0 of both polynomials
    std::vector<uint64_t> last(2 * N);
    std::copy(host.begin(), host.begin() + N, last.begin());
    std::copy(host.begin() + L * N, host.begin() + L * N + N, last.begin() + N);
    up(k.pk_last, last.data(), last.size());
    sync();
  }

  // ---- ciphertext transport: u32 count, then the raw coefficient words (cheetah-api.cpp:11-29) ----
  // Host staging for ciphertext batches (up to 128 MiB per layer): one grow-only page-locked buffer per object, so
  // copies run asynchronously at the link's rate and the pages are touched once, not on every call.  Every public
  // method ends with the stream drained, so the buffer is free when the next call starts.
  struct Pinned {
    uint64_t *p = nullptr;
    size_t words = 0;
  };
  mutable Pinned staging;
  mutable uint32_t *flag_host = nullptr;
  uint64_t *stage(size_t words) const {
    if (staging.words < words) {
      if (staging.p) mmh_host_free(staging.p);
      staging.p = nullptr, staging.words = 0;
      void *q = nullptr;
      must(mmh_host_alloc(&q, words * 8), "mmh_host_alloc");
      staging.p = static_cast<uint64_t *>(q), staging.words = words;
    }
    return staging.p;
  }
  // enqueue the copy of a batch to the staging buffer; after sync(), send_staged() puts it on the wire
  void stage_cts(const Dev &d, size_t words) const { down(stage(words), d, words); }
  void send_staged(size_t n_cts, size_t words_per_ct) const {
    uint32_t n = (uint32_t)n_cts;
    io->send_data(&n, sizeof(n));
    io->send_data(staging.p, n_cts * words_per_ct * 8);
    io->flush();
  }
  Dev recv_cts(size_t expect, size_t words_per_ct) const {  // waits for the peer, then copies asynchronously
    uint32_t n = 0;
    io->recv_data(&n, sizeof(n));
    if (n != expect) throw std::runtime_error("CheetahLinear: ciphertext count mismatch");
    const size_t words = (size_t)n * words_per_ct;
    uint64_t *host = stage(words);
    io->recv_data(host, words * 8);
    Dev d = mk(words);
    up(d, host, words);
    return d;
  }

  // ---- fresh symmetric ciphertexts: c0 and the 32-byte seed of c1 = a on the wire, as seal::Serializable<Ciphertext>
  // (encrypt_symmetric(...).save(), cheetah-api.cpp:11-29 on :401-406); the receiver expands the seed again ----
  struct Fresh {
    Dev a;
    uint32_t seed[8];
  };
  Fresh fresh(const mmh_ctx *c, size_t n) const {  // `a` for n ciphertexts at the first level, from a public seed
    Fresh f;
    if (fixed_key) {  // tests: the n-th seed is the key with n folded in
      for (int i = 0; i < 8; ++i) f.seed[i] = key[i] ^ (uint32_t)(0x9E3779B9u * (uint32_t)(seed_count + 1 + i));
      ++seed_count;
    } else {
      FILE *r = std::fopen("/dev/urandom", "rb");
      const bool ok = r && std::fread(f.seed, 1, sizeof(f.seed), r) == sizeof(f.seed);
      if (r) std::fclose(r);
      if (!ok) throw std::runtime_error("CheetahLinear: cannot draw a ciphertext seed");
    }
    f.a = mk(n * L * N);
    must(mmh_sample_uniform(c, 0, f.seed, 0, f.a.p, n, stream), "sample_uniform");
    return f;
  }
  void send_fresh(const mmh_ctx *c, const Dev &cts, const Fresh &f, size_t n) const {  // drains the stream
    Dev c0 = mk(n * L * N);
    must(mmh_ct_split_c0(c, 0, cts.p, n, c0.p, stream), "split");
    stage_cts(c0, n * L * N);
    sync();
    uint32_t cnt = (uint32_t)n;
    io->send_data(&cnt, sizeof(cnt));
    io->send_data(f.seed, sizeof(f.seed));
    io->send_data(staging.p, n * L * N * 8);
    io->flush();
  }
  struct Inbound {  // a fresh batch as it came off the wire: c0 at the head of the staging buffer, and a's seed
    uint32_t seed[8];
    size_t n = 0;
  };
  // waits for the peer; `more_words` of the staging buffer after the batch are the caller's (its outbound bytes)
  Inbound recv_fresh_host(size_t expect, size_t more_words = 0) const {
    Inbound in;
    uint32_t n = 0;
    io->recv_data(&n, sizeof(n));
    if (n != expect) throw std::runtime_error("CheetahLinear: ciphertext count mismatch");
    io->recv_data(in.seed, sizeof(in.seed));
    in.n = n;
    io->recv_data(stage(in.n * L * N + more_words), in.n * L * N * 8);
    return in;
  }
  // the batch on device w (its own device current): copy of c0, expansion of the seed, both enqueued on w's stream.
  // `c` is w's context of the batch's kind.
  Dev expand_fresh(const Worker &w, const mmh_ctx *c, const Inbound &in) const {
    const size_t half = in.n * L * N;
    Dev c0 = w.mk(half), a = w.mk(half), cts = w.mk(2 * half);
    w.up(c0, staging.p, half);
    must(mmh_sample_uniform(c, 0, in.seed, 0, a.p, in.n, w.stream), "sample_uniform");
    must(mmh_ct_join(c, 0, c0.p, a.p, in.n, cts.p, w.stream), "join");
    return cts;
  }
  Dev recv_fresh(const mmh_ctx *c, size_t expect) const {  // waits for the peer; the rest is enqueued
    return expand_fresh(*this, c, recv_fresh_host(expect));
  }

  // conv2DOneFilter's zero-filter verdict (hom_conv2d_ss.cc:573-580), written by the layer call on the device
  uint32_t *zero_flag_host() const {
    if (!flag_host) {
      void *q = nullptr;
      must(mmh_host_alloc(&q, 64), "mmh_host_alloc");
      flag_host = static_cast<uint32_t *>(q);
    }
    return flag_host;
  }

  // CheetahLinear::conv2d on device-resident tensors: d_img [C][H][W] reduced into the ring, d_filters
  // [M][C][fh][fw] (ALICE); the share / decrypted tensor [M][oh][ow] is left in d_out.  Ends with the stream drained.
  void conv_core(int party, const ConvMeta &meta, mmh_conv_plan *plan, size_t n_in, size_t n_out, const Dev &d_img,
                 const Dev *d_filters, const Dev &d_out, const std::function<void()> &before_wait) const {
    if (party == kBob) {
      Fresh f = fresh(ring.ctx, n_in);
      Dev e = noise(n_in), cts = mk(n_in * 2 * L * N);
      must(mmh_conv_encrypt_image(plan, ring.sk.p, d_img.p, f.a.p, (const int64_t *)e.p, cts.p, stream),
           "CheetahLinear::conv2d encryptImage");
      send_fresh(ring.ctx, cts, f, n_in);
      // the result ciphertexts arrive in their compact form (mmh_conv_result_pack: the cleared low bits and the
      // removed coefficients of c0 are not on the wire, as after seal::Ciphertext::save's compression)
      size_t packed_words = 0;
      must(mmh_conv_result_packed_words(plan, &packed_words), "CheetahLinear::conv2d");
      Dev wire = recv_cts(n_out, packed_words), res = mk(n_out * 2 * N);
      must(mmh_conv_result_unpack(plan, wire.p, res.p, stream), "CheetahLinear::conv2d unpack");
      must(mmh_conv_decrypt(plan, ring.sk_last.p, res.p, d_out.p, stream), "CheetahLinear::conv2d decryptToTensor");
      before_wait();
      sync();
    } else {
      // randomness and the flag are enqueued before the (blocking) receive, so they run while the client encrypts
      Reshare rs = reshare(n_out);
      Dev &u = rs.u, &e1 = rs.e1, &mask = rs.mask;
      Dev out_cts = mk(n_out * 2 * N), flag = mk(1);
      Dev cts = recv_fresh(ring.ctx, n_in);
      // encodeImage + encodeFilters + conv2DSS (cheetah-api.cpp:401-445) in one device call
      must(mmh_conv_server_tensors(plan, cts.p, meta.is_shared_input ? d_img.p : nullptr, d_filters->p, ring.pk_last.p,
                                   (const int64_t *)u.p, (const int64_t *)e1.p, mask.p, out_cts.p, d_out.p,
                                   (uint32_t *)flag.p, stream),
           "CheetahLinear::conv2d conv2DSS:");
      uint32_t *verdict = zero_flag_host();
      must(mmh_memcpy_d2h(verdict, flag.p, 4, stream), "d2h");
      size_t packed_words = 0;
      must(mmh_conv_result_packed_words(plan, &packed_words), "CheetahLinear::conv2d");
      Dev wire = mk(n_out * packed_words);
      must(mmh_conv_result_pack(plan, out_cts.p, wire.p, stream), "CheetahLinear::conv2d pack");
      stage_cts(wire, n_out * packed_words);
      before_wait();
      sync();
      if (*verdict) throw std::runtime_error("CheetahLinear::conv2d conv2DSS: " + CodeMessage(Code::ERR_INTERNAL));
      send_staged(n_out, packed_words);
    }
  }

  // ---- ALICE over several devices ---------------------------------------------------------------------------------
  // A layer's output ciphertexts do not depend on each other, so its server side is cut the way the reference cuts it
  // over threads (LaunchWorks: contiguous ranges of ceil(n / workers) jobs, hom_conv2d_ss.cc:597-607): ranges of
  // output filters (conv2d), of row blocks (fc), of ciphertexts (bn).  Every device takes the client's batch from the
  // one page-locked buffer it was received into (each device has its own link to the host), draws its range of the
  // layer's randomness (mmh_sample_reshare_range: the wire carries the bytes of the one-device run) and copies its
  // packed results into its slice of the same buffer, which goes to the channel in one piece.  One host thread
  // enqueues for all devices (every call is asynchronous), then waits for each stream.
  size_t team_size() const { return 1 + helpers.size(); }
  const Worker &member(size_t i) const { return i == 0 ? static_cast<const Worker &>(*this) : *helpers[i - 1]; }
  struct Range {
    size_t lo, hi;
    size_t count() const { return hi - lo; }
  };
  static Range cut(size_t total, size_t workers, size_t w) {
    const size_t per = (total + workers - 1) / workers;
    return Range{std::min(total, w * per), std::min(total, (w + 1) * per)};
  }
  void send_words(size_t n_cts, const uint64_t *host, size_t words) const {
    uint32_t n = (uint32_t)n_cts;
    io->send_data(&n, sizeof(n));
    io->send_data(host, words * 8);
    io->flush();
  }

  // input / filters / out: NHWC, HWIO and NHWC raw values (nhwc, as conv2d_nhwc) or [C][H][W] ring elements, the
  // filter tensors and [M][oh][ow] (as conv2d)
  void conv_team(const mmh_conv_meta &m, bool nhwc, const uint64_t *input, const uint64_t *filters_hwio,
                 const std::vector<Tensor<uint64_t>> *filters, uint64_t *out, uint64_t ring_mask) const {
    mmh_conv_plan *whole = conv_plan(m);
    size_t n_in, n_cs, n_sp, n_out, packed = 0;
    int64_t os[3];
    must(mmh_conv_plan_info(whole, &n_in, &n_cs, &n_sp, &n_out, os, nullptr), "plan_info");
    must(mmh_conv_result_packed_words(whole, &packed), "CheetahLinear::conv2d");
    const size_t C = (size_t)m.in_c, H = (size_t)m.in_h, W = (size_t)m.in_w, fh = (size_t)m.f_h, fw = (size_t)m.f_w,
                 M = (size_t)m.n_filters, f_words = C * fh * fw, img_words = C * H * W, px = (size_t)(os[1] * os[2]);
    const uint64_t sid0 = stream_id;
    stream_id += 3;
    struct Job {
      const Worker *w = nullptr;
      Range r{0, 0};
      mmh_conv_plan *plan = nullptr;
      const uint64_t *filt_p = nullptr;
      Dev img, filt, u, e1, mask, out_cts, share, share_t, flag, wire, cts;
    };
    std::vector<Job> jobs;
    for (size_t i = 0; i < team_size(); ++i) {
      const Range r = cut(M, team_size(), i);
      if (!r.count()) continue;
      const Worker &w = member(i);
      DeviceGuard dg(w.device);
      Job j;
      j.w = &w, j.r = r;
      mmh_conv_meta mw = m;
      mw.n_filters = (int64_t)r.count();
      j.plan = w.conv_plan(mw);
      if (m.is_shared_input) {
        j.img = w.mk(img_words);
        if (nhwc) {
          Dev raw = w.mk(img_words);
          w.up(raw, input, img_words);
          must(mmh_repack_nhwc_to_chw(raw.p, H, W, C, ring_mask, j.img.p, w.stream), "repack input");
        } else {
          w.up(j.img, input, img_words);
        }
      }
      if (nhwc) {  // the [FH][FW][CI][CO] array is cut along its fastest axis: it crosses whole and is cut on the device
        Dev fraw = w.mk(M * f_words);
        j.filt = w.mk(M * f_words);
        w.up(fraw, filters_hwio, M * f_words);
        must(mmh_repack_hwio_to_oihw(fraw.p, fh, fw, C, M, ring_mask, j.filt.p, w.stream), "repack filters");
        j.filt_p = j.filt.p + r.lo * f_words;
      } else {
        j.filt = w.mk(r.count() * f_words);
        for (size_t k = r.lo; k < r.hi; ++k)
          must(mmh_memcpy_h2d(j.filt.p + (k - r.lo) * f_words, (*filters)[k].data(), f_words * 8, w.stream), "h2d");
        j.filt_p = j.filt.p;
      }
      const size_t cts_w = r.count() * n_sp;
      j.u = w.mk(cts_w * N), j.e1 = w.mk(cts_w * N), j.mask = w.mk(cts_w * N);
      must(mmh_sample_reshare_range(w.ring.ctx, key, sid0, r.lo * n_sp, cts_w, (int64_t *)j.u.p, (int64_t *)j.e1.p,
                                    j.mask.p, w.stream),
           "sample_reshare");
      j.out_cts = w.mk(cts_w * 2 * N), j.share = w.mk(r.count() * px), j.flag = w.mk(1);
      j.wire = w.mk(cts_w * packed);
      if (nhwc) j.share_t = w.mk(r.count() * px);
      jobs.push_back(std::move(j));
    }
    // page-locked buffer: [client's c0 | packed results | shares | one verdict word per device]
    const size_t half = n_in * L * N, out_words = M * px;
    const Inbound in = recv_fresh_host(n_in, n_out * packed + out_words + team_size());
    uint64_t *wire_h = staging.p + half, *share_h = wire_h + n_out * packed, *flags_h = share_h + out_words;
    for (size_t k = 0; k < jobs.size(); ++k) {
      Job &j = jobs[k];
      const Worker &w = *j.w;
      DeviceGuard dg(w.device);
      j.cts = expand_fresh(w, w.ring.ctx, in);
      must(mmh_conv_server_tensors(j.plan, j.cts.p, m.is_shared_input ? j.img.p : nullptr, j.filt_p, w.ring.pk_last.p,
                                   (const int64_t *)j.u.p, (const int64_t *)j.e1.p, j.mask.p, j.out_cts.p, j.share.p,
                                   (uint32_t *)j.flag.p, w.stream),
           "CheetahLinear::conv2d conv2DSS:");
      flags_h[k] = 0;
      must(mmh_memcpy_d2h(flags_h + k, j.flag.p, 4, w.stream), "d2h");
      must(mmh_conv_result_pack(j.plan, j.out_cts.p, j.wire.p, w.stream), "CheetahLinear::conv2d pack");
      w.down(wire_h + j.r.lo * n_sp * packed, j.wire, j.r.count() * n_sp * packed);
      if (nhwc) {
        must(mmh_repack_chw_to_nhwc(j.share.p, j.r.count(), (size_t)os[1], (size_t)os[2], j.share_t.p, w.stream),
             "repack output");
        w.down(share_h + j.r.lo * px, j.share_t, j.r.count() * px);
      } else {
        w.down(share_h + j.r.lo * px, j.share, j.r.count() * px);
      }
    }
    for (Job &j : jobs) {
      DeviceGuard dg(j.w->device);
      j.w->sync();
    }
    for (size_t k = 0; k < jobs.size(); ++k)
      if ((uint32_t)flags_h[k])
        throw std::runtime_error("CheetahLinear::conv2d conv2DSS: " + CodeMessage(Code::ERR_INTERNAL));
    send_words(n_out, wire_h, n_out * packed);
    if (!nhwc) {
      std::memcpy(out, share_h, out_words * 8);
    } else {  // each device left [oh][ow][its filters]; the rows interleave into [oh][ow][M]
      for (const Job &j : jobs)
        for (size_t q = 0; q < px; ++q)
          std::memcpy(out + q * M + j.r.lo, share_h + j.r.lo * px + q * j.r.count(), j.r.count() * 8);
    }
  }

  // wd: rows x cols row-major ring elements, or (in_out) cols x rows raw values; the sub-matrix check is the caller's
  void fc_team(const mmh_fc_meta &m, const uint64_t *vec_h, const uint64_t *wd, bool in_out, uint64_t ring_mask,
               uint64_t *out) const {
    mmh_fc_plan *whole = fc_plan(m);
    int64_t split[2];
    size_t n_in, n_out;
    must(mmh_fc_plan_info(whole, split, &n_in, &n_out), "plan_info");
    const size_t rows = (size_t)m.rows, cols = (size_t)m.cols;
    const uint64_t sid0 = stream_id;
    stream_id += 3;
    struct Job {
      const Worker *w = nullptr;
      Range b{0, 0};          // row blocks
      size_t r0 = 0, r1 = 0;  // their rows
      mmh_fc_plan *plan = nullptr;
      Dev wmat, wpts, spts, u, e1, mask, out_cts, share, cts;
    };
    std::vector<Job> jobs;
    for (size_t i = 0; i < team_size(); ++i) {
      const Range b = cut(n_out, team_size(), i);
      if (!b.count()) continue;
      const Worker &w = member(i);
      DeviceGuard dg(w.device);
      Job j;
      j.w = &w, j.b = b, j.r0 = b.lo * (size_t)split[0], j.r1 = std::min(rows, b.hi * (size_t)split[0]);
      const mmh_fc_meta sub{(int64_t)(j.r1 - j.r0), m.cols, m.is_shared_input};
      j.plan = w.fc_plan(sub, split);
      const uint64_t *wp;
      if (in_out) {  // [cols][rows] -> [rows][cols] on the device, then the block's rows
        Dev raw = w.mk(rows * cols);
        j.wmat = w.mk(rows * cols);
        w.up(raw, wd, rows * cols);
        must(mmh_repack_hwio_to_oihw(raw.p, 1, 1, cols, rows, ring_mask, j.wmat.p, w.stream), "CheetahLinear::fc transpose");
        wp = j.wmat.p + j.r0 * cols;
      } else {
        j.wmat = w.mk((j.r1 - j.r0) * cols);
        w.up(j.wmat, wd + j.r0 * cols, (j.r1 - j.r0) * cols);
        wp = j.wmat.p;
      }
      j.wpts = w.mk(b.count() * n_in * N);
      must(mmh_fc_encode_weights(j.plan, wp, j.wpts.p, w.stream), "CheetahLinear::fc encodeWeightMatrix error");
      if (m.is_shared_input) {
        Dev vec = w.mk(cols);
        w.up(vec, vec_h, cols);
        j.spts = w.mk(n_in * N);
        must(mmh_fc_encode_input(j.plan, vec.p, j.spts.p, w.stream), "CheetahLinear::fc encodeInputVector error");
      }
      j.u = w.mk(b.count() * N), j.e1 = w.mk(b.count() * N), j.mask = w.mk(b.count() * N);
      must(mmh_sample_reshare_range(w.ring.ctx, key, sid0, b.lo, b.count(), (int64_t *)j.u.p, (int64_t *)j.e1.p, j.mask.p,
                                    w.stream),
           "sample_reshare");
      j.out_cts = w.mk(b.count() * 2 * N), j.share = w.mk(j.r1 - j.r0);
      jobs.push_back(std::move(j));
    }
    const size_t half = n_in * L * N;
    const Inbound in = recv_fresh_host(n_in, n_out * 2 * N + rows);
    uint64_t *wire_h = staging.p + half, *share_h = wire_h + n_out * 2 * N;
    for (Job &j : jobs) {
      const Worker &w = *j.w;
      DeviceGuard dg(w.device);
      j.cts = expand_fresh(w, w.ring.ctx, in);
      must(mmh_fc_server(j.plan, j.cts.p, j.spts.p, j.wpts.p, w.ring.pk_last.p, (const int64_t *)j.u.p,
                         (const int64_t *)j.e1.p, j.mask.p, j.out_cts.p, j.share.p, w.stream),
           "CheetahLinear::fc matmul2D error");
      w.down(share_h + j.r0, j.share, j.r1 - j.r0);
      w.down(wire_h + j.b.lo * 2 * N, j.out_cts, j.b.count() * 2 * N);
    }
    for (Job &j : jobs) {
      DeviceGuard dg(j.w->device);
      j.w->sync();
    }
    send_words(n_out, wire_h, n_out * 2 * N);
    std::memcpy(out, share_h, rows * 8);
  }

  // HomBNSS::bn (SIMD path): ranges of the vector's ciphertexts
  void bn_team(size_t n, bool shared, const uint64_t *in_h, const uint64_t *scale_h, uint64_t *out) const {
    const size_t sub = std::min(n, N), n_ct = (n + sub - 1) / sub;
    const uint64_t rsid = sid();
    struct Job {
      const Worker *w = nullptr;
      Range c{0, 0};          // ciphertexts
      size_t e0 = 0, e1 = 0;  // their elements
      Dev sc_pts, in_pts, rnd, out_cts, share, cts;
    };
    std::vector<Job> jobs;
    for (size_t i = 0; i < team_size(); ++i) {
      const Range c = cut(n_ct, team_size(), i);
      if (!c.count()) continue;
      const Worker &w = member(i);
      DeviceGuard dg(w.device);
      Job j;
      j.w = &w, j.c = c, j.e0 = c.lo * sub, j.e1 = std::min(n, c.hi * sub);
      const size_t cnt = j.e1 - j.e0;
      Dev scale = w.mk(cnt);
      w.up(scale, scale_h + j.e0, cnt);
      j.sc_pts = w.mk(c.count() * N);
      must(mmh_bn_encode_vector(w.field.ctx, scale.p, cnt, j.sc_pts.p, w.stream), "bn encodeScales");
      if (shared) {
        Dev vec = w.mk(cnt);
        w.up(vec, in_h + j.e0, cnt);
        j.in_pts = w.mk(c.count() * N);
        must(mmh_bn_encode_vector(w.field.ctx, vec.p, cnt, j.in_pts.p, w.stream), "bn encodeVector");
      }
      j.rnd = w.mk(cnt);  // words [e0, e1) of the vector's mask stream (e0 is a multiple of N, a block holds 8 words)
      must(mmh_chacha20_words(key, rsid, j.e0 / 8, j.rnd.p, cnt, w.stream), "chacha20");
      j.out_cts = w.mk(c.count() * 2 * N), j.share = w.mk(cnt);
      jobs.push_back(std::move(j));
    }
    const size_t half = n_ct * L * N;
    const Inbound in = recv_fresh_host(n_ct, n_ct * 2 * N + n);
    uint64_t *wire_h = staging.p + half, *share_h = wire_h + n_ct * 2 * N;
    for (Job &j : jobs) {
      const Worker &w = *j.w;
      DeviceGuard dg(w.device);
      j.cts = expand_fresh(w, w.field.ctx, in);
      must(mmh_bn_server(w.field.ctx, j.cts.p + j.c.lo * 2 * L * N, j.in_pts.p, j.sc_pts.p, j.e1 - j.e0, j.rnd.p,
                         j.out_cts.p, j.share.p, w.stream),
           "bn failed");
      w.down(share_h + j.e0, j.share, j.e1 - j.e0);
      w.down(wire_h + j.c.lo * 2 * N, j.out_cts, j.c.count() * 2 * N);
    }
    for (Job &j : jobs) {
      DeviceGuard dg(j.w->device);
      j.w->sync();
    }
    send_words(n_ct, wire_h, n_ct * 2 * N);
    std::memcpy(out, share_h, n * 8);
  }
};

namespace {
// MM_GEMINI_DEVICES="0,1,..." gives the device list to callers that only know the reference's constructor
// (InitLinear in library.cc, the tests' drivers)
CheetahLinear::Options legacy_options(size_t nthreads, int device) {
  CheetahLinear::Options opt;
  opt.nthreads = nthreads;
  opt.devices.assign(1, device);
  if (const char *env = std::getenv("MM_GEMINI_DEVICES")) {
    std::vector<int> list;
    for (const char *p = env; *p;) {
      char *end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p || v < 0) throw std::invalid_argument("CheetahLinear: MM_GEMINI_DEVICES is not a list of devices");
      list.push_back((int)v);
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') throw std::invalid_argument("CheetahLinear: MM_GEMINI_DEVICES is not a list of devices");
    }
    if (!list.empty()) opt.devices = list;
  }
  return opt;
}
}  // namespace

CheetahLinear::CheetahLinear(int party, mm::IOChannel *io, uint64_t base_mod, size_t nthreads, int device)
    : CheetahLinear(party, io, base_mod, legacy_options(nthreads, device)) {}

CheetahLinear::CheetahLinear(int party, mm::IOChannel *io, uint64_t base_mod, const Options &opt)
    : party_(party), io_(io), base_mod_(base_mod), impl_(new Impl()) {
  if (base_mod < 2ULL || (int)std::log2(base_mod) >= 45)  // cheetah-api.cpp:76-78
    throw std::logic_error("CheetahLinear: base_mod out-of-bound [2, 2^45)");
  if (party != kAlice && party != kBob) throw std::invalid_argument("CheetahLinear: party");
  if (opt.devices.empty()) throw std::invalid_argument("CheetahLinear: no device");
  const int device = opt.devices[0];
  impl_->io = io;
  impl_->device = device;
  if (opt.sampler_key) {
    std::copy(opt.sampler_key, opt.sampler_key + 8, impl_->key);
    impl_->fixed_key = true;
  } else {
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (!f || std::fread(impl_->key, 1, sizeof(impl_->key), f) != sizeof(impl_->key)) {
      if (f) std::fclose(f);
      throw std::runtime_error("CheetahLinear: cannot seed the sampler");
    }
    std::fclose(f);
  }
  const bool is_mod_2k = (base_mod & (base_mod - 1)) == 0;
  if (!is_mod_2k) throw std::runtime_error("Not implemented yet.");  // :143-147
  field_mod_ = kFieldPrimes.at((int)std::log2(base_mod));            // setUpForBN :153
  const int bits[2] = {60, 49};                                      // :99-103
  {
    DeviceGuard dg(device);
    must(mmh_stream_create(&impl_->stream), "stream");
    must(mmh_ctx_create_bits(&impl_->ring.ctx, Impl::N, bits, 2, base_mod, device), "SEALContext");
    impl_->setup_keys(impl_->ring, party);
    must(mmh_ctx_create_bits(&impl_->field.ctx, Impl::N, bits, 2, field_mod_, device), "SEALContext (SIMD)");
    impl_->setup_keys(impl_->field, party);
  }
  // ALICE's further devices: contexts and BOB's ring public key (the SIMD path's server side uses no key)
  for (size_t i = 1; party == kAlice && i < opt.devices.size(); ++i) {
    std::unique_ptr<Worker> w(new Worker());
    w->device = opt.devices[i];
    DeviceGuard dg(w->device);
    must(mmh_stream_create(&w->stream), "stream");
    must(mmh_ctx_create_bits(&w->ring.ctx, Impl::N, bits, 2, base_mod, w->device), "SEALContext");
    must(mmh_ctx_create_bits(&w->field.ctx, Impl::N, bits, 2, field_mod_, w->device), "SEALContext (SIMD)");
    const std::vector<uint64_t> &pk = impl_->pk_host;
    std::vector<uint64_t> last(2 * Impl::N);
    std::copy(pk.begin(), pk.begin() + Impl::N, last.begin());
    std::copy(pk.begin() + Impl::L * Impl::N, pk.begin() + Impl::L * Impl::N + Impl::N, last.begin() + Impl::N);
    w->ring.pk = w->mk(pk.size());
    w->ring.pk_last = w->mk(last.size());
    w->up(w->ring.pk, pk.data(), pk.size());
    w->up(w->ring.pk_last, last.data(), last.size());
    w->sync();
    impl_->helpers.push_back(std::move(w));
  }
}

CheetahLinear::~CheetahLinear() = default;

void CheetahLinear::release_device_buffers() const {
  for (size_t i = 0; i < impl_->team_size(); ++i) {
    const Worker &w = impl_->member(i);
    DeviceGuard dg(w.device);
    w.sync();
    w.pool.release();
  }
}
size_t CheetahLinear::n_devices() const { return impl_->team_size(); }

void CheetahLinear::conv2d(const Tensor<uint64_t> &in_tensor, const std::vector<Tensor<uint64_t>> &filters,
                           const ConvMeta &meta, Tensor<uint64_t> &out_tensor) const {
  if (!meta.ishape.IsSameSize(in_tensor.shape())) throw std::invalid_argument("CheetahLinear::conv2d meta.ishape mismatch");
  if (meta.n_filters != filters.size()) throw std::invalid_argument("CheetahLinear::conv2d meta.n_filters mismatch");
  for (const auto &f : filters)
    if (!meta.fshape.IsSameSize(f.shape())) throw std::invalid_argument("CheetahLinear::conv2d meta.fshape mismatch");
  const Impl &I = *impl_;
  DeviceGuard dg(I.device);
  mmh_conv_meta m{meta.ishape.channels(), meta.ishape.height(), meta.ishape.width(), meta.fshape.height(),
                  meta.fshape.width(), (int64_t)meta.n_filters, meta.padding == Padding::SAME, (int32_t)meta.stride,
                  meta.is_shared_input};
  mmh_conv_plan *plan = I.conv_plan(m);
  size_t n_in, n_cs, n_sp, n_out;
  int64_t oshape[3];
  must(mmh_conv_plan_info(plan, &n_in, &n_cs, &n_sp, &n_out, oshape, nullptr), "plan_info");
  TensorShape out_shape({oshape[0], oshape[1], oshape[2]});
  if (party_ == kAlice && I.team_size() > 1 && meta.n_filters > 1) {
    if (!out_tensor.shape().IsSameSize(out_shape)) out_tensor.Reshape(out_shape);
    I.conv_team(m, false, in_tensor.data(), nullptr, &filters, out_tensor.data(), base_mod_ - 1);
    return;
  }
  if (!out_tensor.shape().IsSameSize(out_shape)) out_tensor.Reshape(out_shape);
  const size_t img_words = (size_t)meta.ishape.num_elements(), out_words = (size_t)out_shape.num_elements();
  Dev img = I.mk(img_words), out = I.mk(out_words), dfilt;
  I.up(img, in_tensor.data(), img_words);
  if (party_ == kAlice) {
    const size_t f_words = (size_t)meta.fshape.num_elements();
    dfilt = I.mk(meta.n_filters * f_words);
    for (size_t k = 0; k < filters.size(); ++k)
      must(mmh_memcpy_h2d(dfilt.p + k * f_words, filters[k].data(), f_words * 8, I.stream), "h2d");
  }
  I.conv_core(party_, meta, plan, n_in, n_out, img, &dfilt, out, [&] { I.down(out_tensor.data(), out, out_words); });
}

// Conv2DWrapper's work (library_fixed_uniform_cheetah.cpp:216-300) without the host-side repacking: the NHWC input,
// the [FH][FW][CI][CO] filters and the NHWC output cross the link as they are and change layout on the device.
void CheetahLinear::conv2d_nhwc(const uint64_t *input_nhwc, const uint64_t *filters_hwio, const ConvMeta &meta,
                                uint64_t *out_nhwc, int64_t expect_oh, int64_t expect_ow) const {
  if (!input_nhwc || !out_nhwc || (party_ == kAlice && !filters_hwio))
    throw std::invalid_argument("CheetahLinear::conv2d_nhwc null array");
  const Impl &I = *impl_;
  DeviceGuard dg(I.device);
  const size_t C = (size_t)meta.ishape.channels(), H = (size_t)meta.ishape.height(), W = (size_t)meta.ishape.width();
  const size_t fh = (size_t)meta.fshape.height(), fw = (size_t)meta.fshape.width(), M = meta.n_filters;
  mmh_conv_meta m{(int64_t)C, (int64_t)H, (int64_t)W, (int64_t)fh, (int64_t)fw, (int64_t)M,
                  meta.padding == Padding::SAME, (int32_t)meta.stride, meta.is_shared_input};
  mmh_conv_plan *plan = I.conv_plan(m);
  size_t n_in, n_cs, n_sp, n_out;
  int64_t oshape[3];
  must(mmh_conv_plan_info(plan, &n_in, &n_cs, &n_sp, &n_out, oshape, nullptr), "plan_info");
  if ((expect_oh && expect_oh != oshape[1]) || (expect_ow && expect_ow != oshape[2]))
    throw std::runtime_error("Conv2DWrapper: output shape differs from the caller's (pads / stride)");
  const size_t img_words = C * H * W, out_words = (size_t)(oshape[0] * oshape[1] * oshape[2]);
  const uint64_t ring_mask = base_mod_ - 1;  // getRingElt
  if (party_ == kAlice && I.team_size() > 1 && M > 1) {
    I.conv_team(m, true, input_nhwc, filters_hwio, nullptr, out_nhwc, ring_mask);
    return;
  }
  Dev raw = I.mk(img_words), img = I.mk(img_words), out = I.mk(out_words), out_t = I.mk(out_words), dfilt;
  I.up(raw, input_nhwc, img_words);
  must(mmh_repack_nhwc_to_chw(raw.p, H, W, C, ring_mask, img.p, I.stream), "repack input");
  if (party_ == kAlice) {
    const size_t f_words = fh * fw * C * M;
    Dev fraw = I.mk(f_words);
    dfilt = I.mk(f_words);
    I.up(fraw, filters_hwio, f_words);
    must(mmh_repack_hwio_to_oihw(fraw.p, fh, fw, C, M, ring_mask, dfilt.p, I.stream), "repack filters");
  }
  I.conv_core(party_, meta, plan, n_in, n_out, img, &dfilt, out, [&] {
    must(mmh_repack_chw_to_nhwc(out.p, (size_t)oshape[0], (size_t)oshape[1], (size_t)oshape[2], out_t.p, I.stream),
         "repack output");
    I.down(out_nhwc, out_t, out_words);
  });
}

void CheetahLinear::fc(const Tensor<uint64_t> &input_vector, const Tensor<uint64_t> &weight_matrix, const FCMeta &meta,
                       Tensor<uint64_t> &out_vec_share) const {
  if (party_ == kAlice && !weight_matrix.shape().IsSameSize(meta.weight_shape))
    throw std::invalid_argument("CheetahLinear::fc weight shape mismatch");
  fc_any(input_vector, party_ == kAlice ? weight_matrix.data() : nullptr, false, meta, out_vec_share);
}
void CheetahLinear::fc_in_out(const Tensor<uint64_t> &input_vector, const uint64_t *weights_in_out, const FCMeta &meta,
                              Tensor<uint64_t> &out_vec_share) const {
  if (party_ == kAlice && !weights_in_out) throw std::invalid_argument("CheetahLinear::fc weights missing");
  fc_any(input_vector, weights_in_out, true, meta, out_vec_share);
}
// wd: rows x cols row-major, or (in_out) cols x rows row-major with unreduced entries
void CheetahLinear::fc_any(const Tensor<uint64_t> &input_vector, const uint64_t *wd, bool in_out, const FCMeta &meta,
                           Tensor<uint64_t> &out_vec_share) const {
  if (!input_vector.shape().IsSameSize(meta.input_shape)) throw std::invalid_argument("CheetahLinear::fc input shape mismatch");
  TensorShape out_shape({meta.weight_shape.dim_size(0)});
  if (!out_vec_share.shape().IsSameSize(out_shape)) out_vec_share.Reshape(out_shape);
  const Impl &I = *impl_;
  DeviceGuard dg(I.device);
  const size_t N = Impl::N, L = Impl::L;
  mmh_fc_meta m{meta.weight_shape.rows(), meta.weight_shape.cols(), meta.is_shared_input};
  mmh_fc_plan *plan = I.fc_plan(m);
  size_t n_in, n_out;
  must(mmh_fc_plan_info(plan, nullptr, &n_in, &n_out), "plan_info");
  const size_t rows = (size_t)m.rows, cols = (size_t)m.cols;
  Dev vec = I.mk(cols);
  I.up(vec, input_vector.data(), cols);
  if (party_ == kBob) {
    Impl::Fresh f = I.fresh(I.ring.ctx, n_in);
    Dev e = I.noise(n_in), cts = I.mk(n_in * 2 * L * N);
    must(mmh_fc_encrypt_input(plan, I.ring.sk.p, vec.p, f.a.p, (const int64_t *)e.p, cts.p, I.stream),
         "CheetahLinear::fc encryptInputVector");
    I.send_fresh(I.ring.ctx, cts, f, n_in);
    Dev res = I.recv_cts(n_out, 2 * N), out = I.mk(rows);
    must(mmh_fc_decrypt(plan, I.ring.sk_last.p, res.p, out.p, I.stream), "CheetahLinear::fc decryptToVector");
    I.down(out_vec_share.data(), out, rows);
    I.sync();
  } else {
    {  // matVecMul rejects a sub-matrix that encodes to zero (hom_fc_ss.cc:414-422); base_mod is a power of two (ctor)
      int64_t split[2];
      must(mmh_fc_plan_info(plan, split, nullptr, nullptr), "plan_info");
      const uint64_t ring_mask = base_mod_ - 1;
      std::vector<uint8_t> nz(n_out * n_in, 0);
      for (size_t r = 0; r < rows; ++r) {
        uint8_t *row_nz = nz.data() + (r / (size_t)split[0]) * n_in;
        for (size_t cb = 0; cb < n_in; ++cb) {
          if (row_nz[cb]) continue;
          const size_t c1 = std::min<size_t>((cb + 1) * split[1], cols);
          for (size_t c = cb * split[1]; c < c1; ++c)
            if ((in_out ? wd[c * rows + r] : wd[r * cols + c]) & ring_mask) {
              row_nz[cb] = 1;
              break;
            }
        }
      }
      for (uint8_t v : nz)
        if (!v) throw std::runtime_error("CheetahLinear::fc matmul2D error [" + CodeMessage(Code::ERR_INVALID_ARG) + "]");
    }
    if (I.team_size() > 1 && n_out > 1) {
      I.fc_team(m, input_vector.data(), wd, in_out, base_mod_ - 1, out_vec_share.data());
      return;
    }
    Dev w = I.mk(rows * cols), wpts = I.mk(n_out * n_in * N), spts;
    if (in_out) {  // [cols][rows] -> [rows][cols], reduced into the ring on the way
      Dev raw = I.mk(rows * cols);
      I.up(raw, wd, rows * cols);
      must(mmh_repack_hwio_to_oihw(raw.p, 1, 1, cols, rows, base_mod_ - 1, w.p, I.stream), "CheetahLinear::fc transpose");
    } else {
      I.up(w, wd, rows * cols);
    }
    must(mmh_fc_encode_weights(plan, w.p, wpts.p, I.stream), "CheetahLinear::fc encodeWeightMatrix error");
    if (meta.is_shared_input) {
      spts = I.mk(n_in * N);
      must(mmh_fc_encode_input(plan, vec.p, spts.p, I.stream), "CheetahLinear::fc encodeInputVector error");
    }
    Impl::Reshare rs = I.reshare(n_out);
    Dev &u = rs.u, &e1 = rs.e1, &mask = rs.mask;
    Dev out_cts = I.mk(n_out * 2 * N), share = I.mk(rows);
    Dev cts = I.recv_fresh(I.ring.ctx, n_in);
    must(mmh_fc_server(plan, cts.p, spts.p, wpts.p, I.ring.pk_last.p, (const int64_t *)u.p, (const int64_t *)e1.p, mask.p,
                       out_cts.p, share.p, I.stream),
         "CheetahLinear::fc matmul2D error");
    I.down(out_vec_share.data(), share, rows);
    I.stage_cts(out_cts, n_out * 2 * N);
    I.sync();
    I.send_staged(n_out, 2 * N);
  }
}

void CheetahLinear::bn(const Tensor<uint64_t> &input_vector, const Tensor<uint64_t> &scale_vector, const BNMeta &meta,
                       Tensor<uint64_t> &out_vector) const {
  if (meta.is_shared_input && !input_vector.shape().IsSameSize(meta.vec_shape))
    throw std::runtime_error("bn input_vector shape mismatch");
  if (meta.target_base_mod != field_mod_)
    throw std::runtime_error("bn encryptVector [" + CodeMessage(Code::ERR_OUT_BOUND) + "]");  // hom_bn_ss.cc:156
  const Impl &I = *impl_;
  DeviceGuard dg(I.device);
  const size_t N = Impl::N, L = Impl::L, n = (size_t)meta.vec_shape.length();
  const size_t n_ct = (n + std::min(n, N) - 1) / std::min(n, N);
  if (!out_vector.shape().IsSameSize(meta.vec_shape)) out_vector.Reshape(meta.vec_shape);
  if (party_ == kBob) {
    Impl::Fresh f = I.fresh(I.field.ctx, n_ct);
    Dev vec = I.mk(n), e = I.noise(n_ct), cts = I.mk(n_ct * 2 * L * N);
    I.up(vec, input_vector.data(), n);
    must(mmh_bn_encrypt_vector(I.field.ctx, I.field.sk.p, vec.p, n, f.a.p, (const int64_t *)e.p, cts.p, I.stream),
         "bn encryptVector");
    I.send_fresh(I.field.ctx, cts, f, n_ct);
    Dev res = I.recv_cts(n_ct, 2 * N), out = I.mk(n);
    must(mmh_bn_decrypt_vector(I.field.ctx, I.field.sk_last.p, res.p, n, out.p, I.stream), "bn decryptToVector");
    I.down(out_vector.data(), out, n);
    I.sync();
  } else {
    if (!scale_vector.shape().IsSameSize(meta.vec_shape)) throw std::runtime_error("bn scale_vector shape mismatch");
    if (I.team_size() > 1 && n_ct > 1) {
      I.bn_team(n, meta.is_shared_input, input_vector.data(), scale_vector.data(), out_vector.data());
      return;
    }
    Dev scale = I.mk(n), sc_pts = I.mk(n_ct * N), in_pts;
    I.up(scale, scale_vector.data(), n);
    must(mmh_bn_encode_vector(I.field.ctx, scale.p, n, sc_pts.p, I.stream), "bn encodeScales");
    if (meta.is_shared_input) {
      Dev vec = I.mk(n);
      I.up(vec, input_vector.data(), n);
      in_pts = I.mk(n_ct * N);
      must(mmh_bn_encode_vector(I.field.ctx, vec.p, n, in_pts.p, I.stream), "bn encodeVector");
    }
    Dev rnd = I.words(n), out_cts = I.mk(n_ct * 2 * N), share = I.mk(n);
    Dev cts = I.recv_fresh(I.field.ctx, n_ct);
    must(mmh_bn_server(I.field.ctx, cts.p, in_pts.p, sc_pts.p, n, rnd.p, out_cts.p, share.p, I.stream), "bn failed");
    I.down(out_vector.data(), share, n);
    I.stage_cts(out_cts, n_ct * 2 * N);
    I.sync();
    I.send_staged(n_ct, 2 * N);
  }
}

// cheetah-api.cpp:652-706
void CheetahLinear::bn_direct(const Tensor<uint64_t> &input_tensor, const Tensor<uint64_t> &scale_vector,
                              const BNMeta &meta, Tensor<uint64_t> &out_tensor) const {
  if (meta.is_shared_input && !input_tensor.shape().IsSameSize(meta.ishape))
    throw std::runtime_error("bn_direct input_vector shape mismatch");
  if (meta.ishape.dims() != 3) throw std::runtime_error("bn_direct encryptVector [" + CodeMessage(Code::ERR_CONFIG) + "]");
  const Impl &I = *impl_;
  DeviceGuard dg(I.device);
  const size_t N = Impl::N, L = Impl::L;
  mmh_bn_tensor_meta m{meta.ishape.channels(), meta.ishape.height(), meta.ishape.width(), meta.is_shared_input};
  size_t n_ct = 0;
  must(mmh_bn_tensor_count(I.ring.ctx, &m, &n_ct, nullptr), "bn_direct getSplitBN");
  const size_t words = (size_t)meta.ishape.num_elements();
  if (!out_tensor.shape().IsSameSize(meta.ishape)) out_tensor.Reshape(meta.ishape);
  if (party_ == kBob) {
    if (!input_tensor.shape().IsSameSize(meta.ishape))
      throw std::runtime_error("bn_direct encryptVector [" + CodeMessage(Code::ERR_CONFIG) + "]");  // hom_bn_ss.cc:527
    Impl::Fresh f = I.fresh(I.ring.ctx, n_ct);
    Dev x = I.mk(words), e = I.noise(n_ct), cts = I.mk(n_ct * 2 * L * N);
    I.up(x, input_tensor.data(), words);
    must(mmh_bn_encrypt_tensor(I.ring.ctx, &m, I.ring.sk.p, x.p, f.a.p, (const int64_t *)e.p, cts.p, I.stream),
         "bn_direct encryptVector");
    I.send_fresh(I.ring.ctx, cts, f, n_ct);
    Dev res = I.recv_cts(n_ct, 2 * N), out = I.mk(words);
    must(mmh_bn_decrypt_tensor(I.ring.ctx, &m, I.ring.sk_last.p, res.p, out.p, I.stream), "bn_direct decryptToTensor");
    I.down(out_tensor.data(), out, words);
    I.sync();
  } else {
    if (scale_vector.dims() != 1 || scale_vector.length() != input_tensor.channels())
      throw std::runtime_error("bn_direct scale_vector shape mismatch");
    Dev scales = I.mk((size_t)scale_vector.length()), in_pts;
    I.up(scales, scale_vector.data(), (size_t)scale_vector.length());
    if (meta.is_shared_input) {
      Dev x = I.mk(words);
      I.up(x, input_tensor.data(), words);
      in_pts = I.mk(n_ct * N);
      must(mmh_bn_encode_tensor(I.ring.ctx, &m, x.p, in_pts.p, I.stream), "bn_direct encodeVector");
    }
    Dev u = I.ternary(n_ct), e0 = I.noise(n_ct), e1 = I.noise(n_ct), share1 = I.words(words), out_cts = I.mk(n_ct * 2 * N);
    Dev cts = I.recv_fresh(I.ring.ctx, n_ct);
    must(mmh_bn_direct_server(I.ring.ctx, &m, cts.p, in_pts.p, scales.p, I.ring.pk.p, (const int64_t *)u.p,
                              (const int64_t *)e0.p, (const int64_t *)e1.p, share1.p, out_cts.p, I.stream),
         "bn_direct failed");
    I.down(out_tensor.data(), share1, words);  // hom_bn_ss.cc:712-723: the random tensor, reduced mod base_mod
    I.stage_cts(out_cts, n_ct * 2 * N);
    I.sync();
    for (size_t i = 0; i < words; ++i) out_tensor.data()[i] &= base_mod_ - 1;
    I.send_staged(n_ct, 2 * N);
  }
}

void CheetahLinear::BOLE(const Tensor<uint64_t> &x, const Tensor<uint64_t> &y, const BNMeta &meta,
                         Tensor<uint64_t> &out_vector) const {
  if (!x.shape().IsSameSize(meta.vec_shape) || !y.shape().IsSameSize(meta.vec_shape))
    throw std::runtime_error("BOLE input_vector shape mismatch");
  if (meta.target_base_mod != field_mod_) throw std::runtime_error("BOLE [" + CodeMessage(Code::ERR_OUT_BOUND) + "]");
  const Impl &I = *impl_;
  DeviceGuard dg(I.device);
  const size_t N = Impl::N, L = Impl::L, n = (size_t)meta.vec_shape.length();
  const size_t n_ct = (n + std::min(n, N) - 1) / std::min(n, N), ct_words = 2 * L * N;
  if (!out_vector.shape().IsSameSize(meta.vec_shape)) out_vector.Reshape(meta.vec_shape);
  Dev dx = I.mk(n), dy = I.mk(n), out = I.mk(n);
  I.up(dx, x.data(), n);
  I.up(dy, y.data(), n);
  if (party_ == kBob) {
    Dev cts = I.mk(2 * n_ct * ct_words);
    Impl::Fresh f = I.fresh(I.field.ctx, 2 * n_ct);  // one seed for both vectors' ciphertexts
    for (int k = 0; k < 2; ++k) {  // encryptVector(x), encryptVector(y)  (cheetah-api.cpp:617-621)
      Dev e = I.noise(n_ct);
      must(mmh_bn_encrypt_vector(I.field.ctx, I.field.sk.p, k ? dy.p : dx.p, n, f.a.p + k * n_ct * L * N,
                                 (const int64_t *)e.p, cts.p + k * n_ct * ct_words, I.stream),
           "BOLE encryptVector");
    }
    I.send_fresh(I.field.ctx, cts, f, 2 * n_ct);
    Dev res = I.recv_cts(n_ct, 2 * N);
    must(mmh_bn_decrypt_vector(I.field.ctx, I.field.sk_last.p, res.p, n, out.p, I.stream), "BOLE decryptToVector");
    // local cross term x_i * y_i (cheetah-api.cpp:648-649)
    must(mmh_bole_cross_term(out.p, dx.p, dy.p, n, field_mod_, I.stream), "BOLE cross term");
    I.down(out_vector.data(), out, n);
    I.sync();
  } else {
    Dev pts = I.mk(2 * n_ct * N);  // [py ; px]
    must(mmh_bn_encode_vector(I.field.ctx, dy.p, n, pts.p, I.stream), "BOLE encodeVector");
    must(mmh_bn_encode_vector(I.field.ctx, dx.p, n, pts.p + n_ct * N, I.stream), "BOLE encodeVector");
    Dev rnd = I.words(n), out_cts = I.mk(n_ct * 2 * N);
    Dev cts = I.recv_fresh(I.field.ctx, 2 * n_ct);
    must(mmh_bole_server(I.field.ctx, cts.p, pts.p, n, rnd.p, out_cts.p, out.p, I.stream), "BOLE bole");
    must(mmh_bole_cross_term(out.p, dx.p, dy.p, n, field_mod_, I.stream), "BOLE cross term");
    I.down(out_vector.data(), out, n);
    I.stage_cts(out_cts, n_ct * 2 * N);
    I.sync();
    I.send_staged(n_ct, 2 * N);
  }
}

// cheetah-api.cpp:206-296.  Shares are combined mod base_mod; the cleartext convolution follows
// Tensor::Conv2D (tensor.h:274-362): SAME pads (f - 1) in total with the larger half first.
bool CheetahLinear::verify(const Tensor<uint64_t> &in_tensor_share, const std::vector<Tensor<uint64_t>> &filters,
                           const ConvMeta &meta, const Tensor<uint64_t> &out_tensor_share, int nbit_precision,
                           size_t *mismatches) const {
  if (filters.empty()) {
    fprintf(stderr, "CheetahLinear::verify number of filters = 0\n");
    return false;
  }
  const TensorShape &ishape = in_tensor_share.shape(), &fshape = filters[0].shape();
  const long stride = (long)meta.stride, H = ishape.height(), W = ishape.width(), fh = fshape.height(),
             fw = fshape.width();
  const long ph = meta.padding == Padding::SAME ? fh - 1 : 0, pw = meta.padding == Padding::SAME ? fw - 1 : 0;
  if (stride < 1 || H + ph < fh || W + pw < fw) return false;
  const TensorShape oshape({(int64_t)meta.n_filters, (H + ph - fh + stride) / stride, (W + pw - fw + stride) / stride});
  if (!oshape.IsSameSize(out_tensor_share.shape())) {
    fprintf(stderr, "CheetahLinear::verify oshape mismatch\n");
    return false;
  }
  if (!ishape.IsSameSize(meta.ishape) || !fshape.IsSameSize(meta.fshape)) {
    fprintf(stderr, "CheetahLinear::verify shape mismatch\n");
    return false;
  }
  const size_t n_in = (size_t)ishape.num_elements(), n_out = (size_t)oshape.num_elements();
  if (party_ == kBob) {
    io_->send_data(in_tensor_share.data(), 8 * n_in);
    io_->send_data(out_tensor_share.data(), 8 * n_out);
    io_->flush();
    return true;
  }
  std::vector<uint64_t> in(n_in), out(n_out);
  io_->recv_data(in.data(), 8 * n_in);
  io_->recv_data(out.data(), 8 * n_out);
  const bool pow2 = (base_mod_ & (base_mod_ - 1)) == 0;
  auto reduce = [&](uint64_t v) { return pow2 ? v & (base_mod_ - 1) : v % base_mod_; };
  auto get_signed = [&](uint64_t v) { return v >= (base_mod_ + 1) / 2 ? (int64_t)(v - base_mod_) : (int64_t)v; };
  double mn = 0, mx = 0;
  for (size_t i = 0; i < n_in; ++i) {
    in[i] = reduce(in[i] + in_tensor_share.data()[i]);
    const double d = get_signed(in[i]) / std::pow(2., nbit_precision);
    mn = i ? std::min(mn, d) : d, mx = i ? std::max(mx, d) : d;
  }
  printf("in_tensor values in [%g,%g]\n", mn, mx);
  for (size_t i = 0; i < n_out; ++i) out[i] = reduce(out[i] + out_tensor_share.data()[i]);
  const long padh = (ph + 1) / 2, padw = (pw + 1) / 2, OH = oshape.height(), OW = oshape.width(), C = ishape.channels();
  size_t cnt_err = 0;
  for (size_t m = 0; m < meta.n_filters; ++m)
    for (long oh = 0, ih = -padh; oh < OH; ++oh, ih += stride)
      for (long ow = 0, iw = -padw; ow < OW; ++ow, iw += stride) {
        unsigned __int128 sum = 0;
        for (long c = 0; c < C; ++c)
          for (long r = 0; r < fh; ++r)
            for (long q = 0; q < fw; ++q) {
              const long y = ih + r, x = iw + q;
              if (y < 0 || y >= H || x < 0 || x >= W) continue;  // zero padding
              const unsigned __int128 prod = (unsigned __int128)reduce(filters[m](c, r, q)) * in[(c * H + y) * W + x];
              sum += (uint64_t)(prod % base_mod_);
            }
        const uint64_t ground = (uint64_t)(sum % base_mod_);
        if (get_signed(ground) != get_signed(out[(m * OH + oh) * OW + ow])) ++cnt_err;
      }
  printf(cnt_err == 0 ? "HomConv: matches\n" : "HomConv: failed\n");
  if (mismatches) *mismatches = cnt_err;
  return true;
}

}  // namespace gemini
