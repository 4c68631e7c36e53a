// gemini::CheetahLinear on libmm_hip (C ABI only: no HIP, no SEAL in this file).
// Choreography follows SCI/src/cheetah/cheetah-api.cpp: keys and pk exchange :73-197, fc :298-375,
// conv2d :377-446, bn :527-603, BOLE :605-650.
#include "mm_gemini/cheetah_linear.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
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
}  // namespace

struct CheetahLinear::Impl {
  static constexpr size_t N = 4096, L = 2;
  int device = 0;
  void *stream = nullptr;  // every copy and kernel of this object; one wait before bytes go to the channel
  mutable DevPool pool;
  Keys ring, field;
  uint32_t key[8];
  mutable uint64_t stream_id = 1;
  mm::IOChannel *io = nullptr;

  // layer geometry is computed once per shape (the reference redoes it on every call)
  mutable std::mutex plan_mu;
  mutable std::map<std::array<int64_t, 9>, mmh_conv_plan *> conv_plans;
  mutable std::map<std::array<int64_t, 3>, mmh_fc_plan *> fc_plans;

  ~Impl() {
    if (stream) (void)mmh_stream_sync(stream);
    for (auto &kv : conv_plans) mmh_conv_plan_destroy(kv.second);
    for (auto &kv : fc_plans) mmh_fc_plan_destroy(kv.second);
    mmh_ctx_destroy(ring.ctx);
    mmh_ctx_destroy(field.ctx);
    if (staging.p) mmh_host_free(staging.p);
    if (flag_host) mmh_host_free(flag_host);
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
  mmh_fc_plan *fc_plan(const mmh_fc_meta &m) const {
    const std::array<int64_t, 3> key{m.rows, m.cols, m.is_shared_input};
    std::lock_guard<std::mutex> g(plan_mu);
    auto it = fc_plans.find(key);
    if (it != fc_plans.end()) return it->second;
    mmh_fc_plan *plan = nullptr;
    must(mmh_fc_plan_create(ring.ctx, &m, &plan), "CheetahLinear::fc getSplit");
    return fc_plans[key] = plan;
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
    FILE *r = std::fopen("/dev/urandom", "rb");
    const bool ok = r && std::fread(f.seed, 1, sizeof(f.seed), r) == sizeof(f.seed);
    if (r) std::fclose(r);
    if (!ok) throw std::runtime_error("CheetahLinear: cannot draw a ciphertext seed");
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
  Dev recv_fresh(const mmh_ctx *c, size_t expect) const {  // waits for the peer; the rest is enqueued
    uint32_t n = 0, seed[8];
    io->recv_data(&n, sizeof(n));
    if (n != expect) throw std::runtime_error("CheetahLinear: ciphertext count mismatch");
    io->recv_data(seed, sizeof(seed));
    const size_t half = (size_t)n * L * N;
    uint64_t *host = stage(half);
    io->recv_data(host, half * 8);
    Dev c0 = mk(half), a = mk(half), cts = mk(2 * half);
    up(c0, host, half);
    must(mmh_sample_uniform(c, 0, seed, 0, a.p, n, stream), "sample_uniform");
    must(mmh_ct_join(c, 0, c0.p, a.p, n, cts.p, stream), "join");
    return cts;
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
};

CheetahLinear::CheetahLinear(int party, mm::IOChannel *io, uint64_t base_mod, size_t /*nthreads*/, int device)
    : party_(party), io_(io), base_mod_(base_mod), impl_(new Impl()) {
  if (base_mod < 2ULL || (int)std::log2(base_mod) >= 45)  // cheetah-api.cpp:76-78
    throw std::logic_error("CheetahLinear: base_mod out-of-bound [2, 2^45)");
  if (party != kAlice && party != kBob) throw std::invalid_argument("CheetahLinear: party");
  impl_->io = io;
  impl_->device = device;
  FILE *f = std::fopen("/dev/urandom", "rb");
  if (!f || std::fread(impl_->key, 1, sizeof(impl_->key), f) != sizeof(impl_->key)) {
    if (f) std::fclose(f);
    throw std::runtime_error("CheetahLinear: cannot seed the sampler");
  }
  std::fclose(f);
  DeviceGuard dg(device);
  must(mmh_stream_create(&impl_->stream), "stream");
  const int bits[2] = {60, 49};  // :99-103
  must(mmh_ctx_create_bits(&impl_->ring.ctx, Impl::N, bits, 2, base_mod, device), "SEALContext");
  impl_->setup_keys(impl_->ring, party);
  const bool is_mod_2k = (base_mod & (base_mod - 1)) == 0;
  if (!is_mod_2k) throw std::runtime_error("Not implemented yet.");  // :143-147
  field_mod_ = kFieldPrimes.at((int)std::log2(base_mod));            // setUpForBN :153
  must(mmh_ctx_create_bits(&impl_->field.ctx, Impl::N, bits, 2, field_mod_, device), "SEALContext (SIMD)");
  impl_->setup_keys(impl_->field, party);
}

CheetahLinear::~CheetahLinear() = default;

void CheetahLinear::release_device_buffers() const {
  DeviceGuard dg(impl_->device);
  impl_->sync();
  impl_->pool.release();
}

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
    Dev w = I.mk(rows * cols), wpts = I.mk(n_out * n_in * N), spts;
    if (in_out) {  // [cols][rows] -> [rows][cols], reduced into the ring on the way
      Dev raw = I.mk(rows * cols);
      I.up(raw, wd, rows * cols);
      must(mmh_repack_hwio_to_oihw(raw.p, 1, 1, cols, rows, base_mod_ - 1, w.p, I.stream), "CheetahLinear::fc transpose");
    } else {
      I.up(w, wd, rows * cols);
    }
    must(mmh_fc_encode_weights(plan, w.p, wpts.p, I.stream), "CheetahLinear::fc encodeWeightMatrix error");
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
