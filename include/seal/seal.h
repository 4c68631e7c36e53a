// seal/seal.h -- the SEAL 3.7.2 surface that M&M's protocol classes touch, over libmm_hip (include/mm_hip.h).
//
// The reference's include/gemini/cheetah/*.cc, SCI/src/cheetah/cheetah-api.cpp and their headers are written
// against Microsoft SEAL (absent from the reference tree: scripts/build-deps.sh fetches it).  This header set gives
// them exactly the symbols they use so that they compile UNCHANGED (-I include, nothing
// copied), with SEAL's data layout -- ct.data(k)[l*N + i], Plaintext coefficient vectors, parms_id chain with
// patch/SEAL.patch's n_special_primes = 0 rule -- and with every polynomial operation executed by the MI355X library
// through its C ABI.  Ciphertext and Plaintext live in host memory, as in SEAL, because the reference's own loops
// index them (hom_conv2d_ss.cc:128-139, :423, :438); each Evaluator / Encryptor / Decryptor / util call stages its
// operands to the device, runs the mmh_* entry point that replaces it, and copies the result back.  That makes this
// facade the PROOF of the boundary (tests/cpp/ref_gemini_test), not the fast path: the fast path is
// include/mm_gemini, which keeps batches resident and calls one entry point per layer.
//
// What is not SEAL: randomness comes from the device ChaCha20 sampler (seeded from the OS or from the caller's
// UniformRandomGenerator), so streams differ from SEAL's Blake2xb ones; save/load write SEAL 3.7's framing with
// compr_mode none (seal/wire.h: header, members, seeded form, BLAKE2b parms_id; parity unpinned) and inflate a
// zstd-compressed stream, which an unmodified SEAL writes by default, with the decoder of seal/zstd_decode.h (zlib
// streams are refused).  Only BFV, n_special_primes = 0, N in {4096, 8192},
// at most two coefficient moduli -- what cheetah-api.cpp:95-105,:151-162 sets up.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <optional>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mm_hip.h"
#include "seal/wire.h"

namespace seal {

using seal_byte = std::byte;
using parms_id_type = std::array<std::uint64_t, 4>;
static constexpr parms_id_type parms_id_zero = {0, 0, 0, 0};

enum class scheme_type : std::uint8_t { none = 0x0, bfv = 0x1, ckks = 0x2 };
enum class sec_level_type : int { none = 0, tc128 = 128, tc192 = 192, tc256 = 256 };
enum class mm_prof_opt : std::uint64_t {
  mm_default = 0x0, mm_force_global = 0x1, mm_force_new = 0x2, mm_force_thread_local = 0x4,
  DEFAULT = 0x0, FORCE_GLOBAL = 0x1, FORCE_NEW = 0x2, FORCE_THREAD_LOCAL = 0x4
};
static constexpr mm_prof_opt mm_force_thread_local = mm_prof_opt::mm_force_thread_local;
static constexpr mm_prof_opt mm_force_new = mm_prof_opt::mm_force_new;
static constexpr mm_prof_opt mm_force_global = mm_prof_opt::mm_force_global;

// SEAL's pools matter for its allocator only; buffers here are std::vector / unique_ptr
class MemoryPoolHandle {
 public:
  explicit operator bool() const { return true; }
};
class MemoryManager {
 public:
  static MemoryPoolHandle GetPool() { return MemoryPoolHandle(); }
  template <typename... Args>
  static MemoryPoolHandle GetPool(mm_prof_opt, Args &&...) { return MemoryPoolHandle(); }
};

namespace facade {
inline void check(int rc, const char *what) {
  if (rc != MMH_OK) throw std::logic_error(std::string(what) + ": " + mmh_last_error());
}
// device buffer for the length of one call; copies are complete when up()/down() return
class DBuf {
 public:
  explicit DBuf(std::size_t words) : words_(words) { check(mmh_malloc(&p_, (words ? words : 1) * 8), "mmh_malloc"); }
  ~DBuf() { (void)mmh_free(p_); }
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  std::uint64_t *get() const { return static_cast<std::uint64_t *>(p_); }
  void up(const void *src, std::size_t words, std::size_t at = 0) const {
    check(mmh_memcpy_h2d(get() + at, src, words * 8, nullptr), "h2d");
    check(mmh_stream_sync(nullptr), "sync");
  }
  void down(void *dst, std::size_t words, std::size_t at = 0) const {
    check(mmh_memcpy_d2h(dst, get() + at, words * 8, nullptr), "d2h");
    check(mmh_stream_sync(nullptr), "sync");
  }
  void zero() const { check(mmh_memset(p_, 0, words_ * 8, nullptr), "memset"); }

 private:
  void *p_ = nullptr;
  std::size_t words_;
};
inline void os_random(void *dst, std::size_t n) {
  FILE *f = std::fopen("/dev/urandom", "rb");
  if (!f || std::fread(dst, 1, n, f) != n) {
    if (f) std::fclose(f);
    throw std::runtime_error("seal facade: cannot read /dev/urandom");
  }
  std::fclose(f);
}
// the device library's context behind one SEALContext
struct Backend {
  mmh_ctx *ctx = nullptr;
  std::size_t N = 0, L = 0;
  std::uint64_t t = 0;
  ~Backend() { (void)mmh_ctx_destroy(ctx); }
};
}  // namespace facade

// ------------------------------------------------------------------------------------------ Modulus
class Modulus {
 public:
  Modulus(std::uint64_t value = 0) { set_value(value); }
  std::uint64_t value() const { return value_; }
  int bit_count() const { return bit_count_; }
  bool is_zero() const { return value_ == 0; }
  const std::array<std::uint64_t, 3> &const_ratio() const { return const_ratio_; }
  bool operator==(const Modulus &o) const { return value_ == o.value_; }
  bool operator!=(const Modulus &o) const { return value_ != o.value_; }

 private:
  void set_value(std::uint64_t v) {
    if (v >> 61) throw std::invalid_argument("Modulus: value can be at most 61 bits");
    value_ = v;
    bit_count_ = 0;
    for (std::uint64_t x = v; x; x >>= 1) ++bit_count_;
    if (v > 1) {  // floor(2^128 / v) and the remainder, as SEAL keeps them
      const unsigned __int128 all = ~(unsigned __int128)0;
      unsigned __int128 quo = all / v;
      std::uint64_t rem = (std::uint64_t)(all % v) + 1;
      if (rem == v) rem = 0, quo += 1;
      const_ratio_ = {(std::uint64_t)quo, (std::uint64_t)(quo >> 64), rem};
    }
  }
  std::uint64_t value_ = 0;
  int bit_count_ = 0;
  std::array<std::uint64_t, 3> const_ratio_{0, 0, 0};
};

class CoeffModulus {
 public:
  // SEAL's prime search (largest primes = 1 mod 2N below 2^bits, smallest of a repeated size first) lives in the
  // device library's context builder; ask it
  static std::vector<Modulus> Create(std::size_t poly_modulus_degree, std::vector<int> bit_sizes) {
    if (bit_sizes.empty() || bit_sizes.size() > 4) throw std::invalid_argument("CoeffModulus::Create: 1 to 4 primes");
    mmh_ctx *c = nullptr;
    facade::check(mmh_ctx_create_bits(&c, poly_modulus_degree, bit_sizes.data(), bit_sizes.size(), 0, current_device()),
                  "CoeffModulus::Create");
    std::uint64_t q[4] = {0, 0, 0, 0};
    const int rc = mmh_ctx_info(c, nullptr, nullptr, q, nullptr, nullptr, nullptr);
    (void)mmh_ctx_destroy(c);
    facade::check(rc, "CoeffModulus::Create");
    std::vector<Modulus> out;
    for (std::size_t i = 0; i < bit_sizes.size(); ++i) out.emplace_back(q[i]);
    return out;
  }
  static int current_device() {
    int d = 0;
    facade::check(mmh_get_device(&d), "mmh_get_device");
    return d;
  }
};

// ------------------------------------------------------------------------------------------ randomness
// SEAL's UniformRandomGenerator (Blake2xb) hands out bytes on the host; this one hands out the device ChaCha20
// keystream (mmh_chacha20_words), keyed from the OS per generator.
class UniformRandomGenerator {
 public:
  UniformRandomGenerator() { facade::os_random(key_, sizeof(key_)); }
  // a generator with a caller-chosen key: what a factory made with UniformRandomGeneratorFactory(key) hands out, so
  // that a test can replay the randomness of a call (SEAL's Blake2xbPRNGFactory(seed) plays that role there)
  explicit UniformRandomGenerator(const std::array<std::uint32_t, 8> &key) { std::memcpy(key_, key.data(), sizeof(key_)); }
  void generate(std::size_t byte_count, seal_byte *destination) {
    if (byte_count == 0) return;
    const std::size_t words = (byte_count + 7) / 8;
    facade::DBuf d(words);
    std::uint64_t sid;
    {
      std::lock_guard<std::mutex> g(mu_);
      sid = next_stream_++;
    }
    facade::check(mmh_chacha20_words(key_, sid, 0, d.get(), words, nullptr), "UniformRandomGenerator::generate");
    if (byte_count % 8 == 0) {
      d.down(destination, words);
    } else {
      std::vector<std::uint64_t> tmp(words);
      d.down(tmp.data(), words);
      std::memcpy(destination, tmp.data(), byte_count);
    }
  }
  // 32 bytes + a stream id for one device sampler call, derived from this generator
  void sampler_key(std::uint32_t key[8], std::uint64_t *stream_id) {
    generate(32, reinterpret_cast<seal_byte *>(key));
    std::lock_guard<std::mutex> g(mu_);
    *stream_id = next_stream_++;
  }
  // the generator's own key and `count` consecutive stream ids of it, for the batched layer calls: one device sampler
  // launch draws a whole layer's randomness from (key, first .. first + count - 1)
  std::uint64_t take_streams(std::uint32_t key[8], std::uint64_t count) {
    std::memcpy(key, key_, sizeof(key_));
    std::lock_guard<std::mutex> g(mu_);
    const std::uint64_t first = next_stream_;
    next_stream_ += count;
    return first;
  }

 private:
  std::uint32_t key_[8];
  std::uint64_t next_stream_ = 1;
  std::mutex mu_;
};
class UniformRandomGeneratorFactory {
 public:
  UniformRandomGeneratorFactory() = default;
  // every generator this factory creates is keyed by `key` with its last word offset by the number of generators made
  // before it: repeatable streams for tests (set through EncryptionParameters::set_random_generator)
  explicit UniformRandomGeneratorFactory(const std::array<std::uint32_t, 8> &key) : fixed_(true), key_(key) {}
  std::shared_ptr<UniformRandomGenerator> create() const {
    if (!fixed_) return std::make_shared<UniformRandomGenerator>();
    std::array<std::uint32_t, 8> k = key_;
    {
      std::lock_guard<std::mutex> g(mu_);
      k[7] += made_++;
    }
    return std::make_shared<UniformRandomGenerator>(k);
  }
  static std::shared_ptr<UniformRandomGeneratorFactory> DefaultFactory() {
    static std::shared_ptr<UniformRandomGeneratorFactory> f = std::make_shared<UniformRandomGeneratorFactory>();
    return f;
  }

 private:
  bool fixed_ = false;
  std::array<std::uint32_t, 8> key_{};
  mutable std::uint32_t made_ = 0;
  mutable std::mutex mu_;
};

// ------------------------------------------------------------------------------------------ parameters
class SEALContext;
class EncryptionParameters {
 public:
  EncryptionParameters(scheme_type scheme = scheme_type::none) : scheme_(scheme) {}
  void set_poly_modulus_degree(std::size_t n) { poly_modulus_degree_ = n, rehash(); }
  void set_coeff_modulus(const std::vector<Modulus> &q) { coeff_modulus_ = q, rehash(); }
  void set_plain_modulus(const Modulus &t) { plain_modulus_ = t, rehash(); }
  void set_plain_modulus(std::uint64_t t) { plain_modulus_ = Modulus(t), rehash(); }
  void set_n_special_primes(int n) {  // patch/SEAL.patch:98-107
    if (n != 0 && n != 1) throw std::logic_error("set_n_special_primes: n is not supported");
    n_special_primes_ = (std::size_t)n, rehash();
  }
  void set_random_generator(std::shared_ptr<UniformRandomGeneratorFactory> f) { random_generator_ = std::move(f); }
  scheme_type scheme() const { return scheme_; }
  std::size_t poly_modulus_degree() const { return poly_modulus_degree_; }
  const std::vector<Modulus> &coeff_modulus() const { return coeff_modulus_; }
  const Modulus &plain_modulus() const { return plain_modulus_; }
  std::size_t n_special_primes() const { return n_special_primes_; }
  std::shared_ptr<UniformRandomGeneratorFactory> random_generator() const { return random_generator_; }
  const parms_id_type &parms_id() const { return parms_id_; }
  bool operator==(const EncryptionParameters &o) const { return parms_id_ == o.parms_id_; }

 private:
  friend class SEALContext;
  void rehash() {
    // EncryptionParameters::compute_parms_id: BLAKE2b-256 over (scheme, N, n_special_primes [patch/SEAL.patch:43-71],
    // moduli, t) -- the value that travels in front of every serialised ciphertext (seal/wire.h)
    std::vector<std::uint64_t> q;
    for (const auto &m : coeff_modulus_) q.push_back(m.value());
    parms_id_ = wire::parms_id_hash((std::uint8_t)scheme_, poly_modulus_degree_, n_special_primes_, q, plain_modulus_.value());
  }
  scheme_type scheme_;
  std::size_t n_special_primes_ = 1, poly_modulus_degree_ = 0;
  std::vector<Modulus> coeff_modulus_;
  Modulus plain_modulus_;
  std::shared_ptr<UniformRandomGeneratorFactory> random_generator_ = UniformRandomGeneratorFactory::DefaultFactory();
  parms_id_type parms_id_ = parms_id_zero;
};

struct EncryptionParameterQualifiers {
  bool using_batching = false;
  bool parameters_set() const { return true; }
};

namespace util {
// one modulus of one level, as seal::util::NTTTables is handed to the transforms
class NTTTables {
 public:
  NTTTables() = default;
  NTTTables(std::shared_ptr<facade::Backend> b, std::size_t limb, Modulus m) : backend_(std::move(b)), limb_(limb), modulus_(m) {}
  const Modulus &modulus() const { return modulus_; }
  std::size_t coeff_count() const { return backend_->N; }
  const facade::Backend &backend() const { return *backend_; }
  std::size_t limb() const { return limb_; }

 private:
  std::shared_ptr<facade::Backend> backend_;
  std::size_t limb_ = 0;
  Modulus modulus_;
};

class ConstRNSIter {  // `ConstRNSIter wrap(ptr, n)`: limb l of the value starts at ptr + l * n
 public:
  ConstRNSIter(const std::uint64_t *ptr, std::size_t poly_modulus_degree) : ptr_(ptr), n_(poly_modulus_degree) {}
  const std::uint64_t *ptr() const { return ptr_; }
  std::size_t poly_modulus_degree() const { return n_; }

 private:
  const std::uint64_t *ptr_;
  std::size_t n_;
};

class RNSTool {
 public:
  RNSTool(std::shared_ptr<facade::Backend> b, int lv, std::size_t limbs) : backend_(std::move(b)), lv_(lv), limbs_(limbs) {}
  // patched form (patch/SEAL.patch:119-175): the input may be shorter than N
  void decrypt_scale_and_round(ConstRNSIter input, std::uint64_t *destination, MemoryPoolHandle) const {
    const std::size_t N = backend_->N, n = input.poly_modulus_degree();
    if (n > N) throw std::invalid_argument("input is not valid for encryption parameters");
    if (n == 0) return;
    facade::DBuf in(limbs_ * N), out(N);
    in.zero();
    for (std::size_t l = 0; l < limbs_; ++l) in.up(input.ptr() + l * n, n, l * N);
    facade::check(mmh_scale_and_round(backend_->ctx, lv_, in.get(), 1, nullptr, N, out.get(), nullptr),
                  "decrypt_scale_and_round");
    out.down(destination, n);
  }

 private:
  std::shared_ptr<facade::Backend> backend_;
  int lv_;
  std::size_t limbs_;
};
}  // namespace util

class SEALContext {
 public:
  class ContextData {
   public:
    const EncryptionParameters &parms() const { return parms_; }
    const parms_id_type &parms_id() const { return parms_.parms_id(); }
    const util::NTTTables *small_ntt_tables() const { return ntt_tables_.data(); }
    const util::RNSTool *rns_tool() const { return rns_tool_.get(); }
    int total_coeff_modulus_bit_count() const { return total_bits_; }
    EncryptionParameterQualifiers qualifiers() const { return qualifiers_; }
    std::shared_ptr<const ContextData> prev_context_data() const { return prev_.lock(); }
    std::shared_ptr<const ContextData> next_context_data() const { return next_; }
    std::size_t chain_index() const { return chain_index_; }
    // facade: the device context and this level's index in it (0 = first, 1 = last)
    const std::shared_ptr<facade::Backend> &backend() const { return backend_; }
    int level() const { return level_; }

   private:
    friend class SEALContext;
    EncryptionParameters parms_;
    std::vector<util::NTTTables> ntt_tables_;
    std::shared_ptr<util::RNSTool> rns_tool_;
    EncryptionParameterQualifiers qualifiers_;
    std::weak_ptr<const ContextData> prev_;
    std::shared_ptr<const ContextData> next_;
    std::shared_ptr<facade::Backend> backend_;
    int total_bits_ = 0, level_ = 0;
    std::size_t chain_index_ = 0;
  };

  SEALContext(const EncryptionParameters &parms, bool expand_mod_chain = true,
              sec_level_type = sec_level_type::tc128) {
    if (parms.scheme() != scheme_type::bfv) throw std::invalid_argument("seal facade: BFV only");
    if (parms.n_special_primes() != 0)
      throw std::invalid_argument("seal facade: set_n_special_primes(0) as cheetah-api.cpp:97 does");
    const auto &q = parms.coeff_modulus();
    if (q.empty() || q.size() > 2) throw std::invalid_argument("seal facade: one or two coefficient moduli");
    auto b = std::make_shared<facade::Backend>();
    std::uint64_t qv[2] = {q[0].value(), q.size() > 1 ? q[1].value() : 0};
    facade::check(mmh_ctx_create(&b->ctx, parms.poly_modulus_degree(), qv, q.size(), parms.plain_modulus().value(),
                                 CoeffModulus::current_device()),
                  "SEALContext");
    b->N = parms.poly_modulus_degree(), b->L = q.size(), b->t = parms.plain_modulus().value();
    const std::uint64_t t = b->t, two_n = 2 * b->N;
    const bool batching = t > 1 && t % two_n == 1 && is_prime(t);
    // chain: every modulus at the first (= key, patch/SEAL.patch:28-37) level, {q0} at the last
    auto first = std::make_shared<ContextData>();
    fill(*first, parms, b, 0, q.size(), batching, q.size() > 1 ? 1 : 0);
    first_ = key_ = last_ = first;
    if (expand_mod_chain && q.size() > 1) {
      EncryptionParameters lp = parms;
      lp.set_coeff_modulus({q[0]});
      auto last = std::make_shared<ContextData>();
      fill(*last, lp, b, 1, 1, batching, 0);
      last->prev_ = first;
      first->next_ = last;
      last_ = last;
    }
  }
  std::shared_ptr<const ContextData> first_context_data() const { return first_; }
  std::shared_ptr<const ContextData> key_context_data() const { return key_; }
  std::shared_ptr<const ContextData> last_context_data() const { return last_; }
  std::shared_ptr<const ContextData> get_context_data(const parms_id_type &id) const {
    if (id == first_->parms_id()) return first_;
    if (id == last_->parms_id()) return last_;
    return nullptr;
  }
  const parms_id_type &first_parms_id() const { return first_->parms_id(); }
  const parms_id_type &key_parms_id() const { return key_->parms_id(); }
  const parms_id_type &last_parms_id() const { return last_->parms_id(); }
  bool parameters_set() const { return true; }
  bool using_keyswitching() const { return false; }

 private:
  static bool is_prime(std::uint64_t v) {
    if (v < 2) return false;
    for (std::uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
      if (v % p == 0) return v == p;
    }
    auto mulmod = [](std::uint64_t a, std::uint64_t b, std::uint64_t m) { return (std::uint64_t)((unsigned __int128)a * b % m); };
    auto powmod = [&](std::uint64_t a, std::uint64_t e, std::uint64_t m) {
      std::uint64_t r = 1;
      for (a %= m; e; e >>= 1, a = mulmod(a, a, m))
        if (e & 1) r = mulmod(r, a, m);
      return r;
    };
    std::uint64_t d = v - 1;
    int s = 0;
    while (!(d & 1)) d >>= 1, ++s;
    for (std::uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
      std::uint64_t x = powmod(a, d, v);
      if (x == 1 || x == v - 1) continue;
      bool comp = true;
      for (int i = 1; i < s && comp; ++i) {
        x = mulmod(x, x, v);
        if (x == v - 1) comp = false;
      }
      if (comp) return false;
    }
    return true;
  }
  static void fill(ContextData &cd, const EncryptionParameters &p, const std::shared_ptr<facade::Backend> &b, int level,
                   std::size_t limbs, bool batching, std::size_t chain_index) {
    cd.parms_ = p;
    cd.backend_ = b;
    cd.level_ = level;
    cd.chain_index_ = chain_index;
    cd.qualifiers_.using_batching = batching;
    for (std::size_t l = 0; l < limbs; ++l) {
      cd.ntt_tables_.emplace_back(b, l, p.coeff_modulus()[l]);
      cd.total_bits_ += p.coeff_modulus()[l].bit_count();
    }
    if (limbs == 2) {  // bits of the product, not the sum of the factors' bits
      const unsigned __int128 Q = (unsigned __int128)p.coeff_modulus()[0].value() * p.coeff_modulus()[1].value();
      cd.total_bits_ = 0;
      for (unsigned __int128 x = Q; x; x >>= 1) ++cd.total_bits_;
    }
    cd.rns_tool_ = std::make_shared<util::RNSTool>(b, level, limbs);
  }
  std::shared_ptr<ContextData> first_, key_, last_;
};

// ------------------------------------------------------------------------------------------ Plaintext / Ciphertext
namespace facade {
// A batch of plaintexts that exists on the device in a cheaper form than N host coefficients each (the raw filter
// tensor of a layer, encoded vectors that never leave HBM): what the batched implementations of the protocol classes
// (openmm_amd/dropin) hand from encodeFilters / encodeImage / encodeVector to conv2DSS / matVecMul / bn through the
// reference's own std::vector<seal::Plaintext> arguments.  A Plaintext that is an element of such a batch behaves like
// any other; its coefficients are brought to the host the first time somebody asks for them.
struct LazyPlain {
  virtual ~LazyPlain() = default;
  virtual std::size_t coeff_count(std::size_t index) const = 0;
  virtual bool is_zero(std::size_t index) const = 0;
  virtual void materialize(std::size_t index, std::vector<std::uint64_t> &out) const = 0;  // coefficient form
};
}  // namespace facade

class Plaintext {
 public:
  Plaintext() = default;
  explicit Plaintext(MemoryPoolHandle) {}
  explicit Plaintext(std::size_t coeff_count) : data_(coeff_count, 0) {}
  void resize(std::size_t coeff_count) {
    if (parms_id_ != parms_id_zero) throw std::logic_error("cannot resize an NTT transformed Plaintext");
    realize();
    data_.resize(coeff_count, 0);
  }
  void release() { lazy_.reset(), data_.clear(), data_.shrink_to_fit(), parms_id_ = parms_id_zero; }
  void set_zero() {
    realize();
    std::fill(data_.begin(), data_.end(), 0);
  }
  std::uint64_t *data() { return realize(), data_.data(); }
  const std::uint64_t *data() const { return realize(), data_.data(); }
  std::size_t coeff_count() const { return lazy_ ? lazy_->coeff_count(lazy_index_) : data_.size(); }
  std::size_t significant_coeff_count() const {
    realize();
    std::size_t n = data_.size();
    while (n && data_[n - 1] == 0) --n;
    return n;
  }
  bool is_zero() const {
    if (lazy_) return lazy_->is_zero(lazy_index_);
    return data_.empty() || std::all_of(data_.begin(), data_.end(), [](std::uint64_t c) { return c == 0; });
  }
  bool is_ntt_form() const { return parms_id_ != parms_id_zero; }
  parms_id_type &parms_id() { return parms_id_; }
  const parms_id_type &parms_id() const { return parms_id_; }
  double &scale() { return scale_; }
  const double &scale() const { return scale_; }
  std::uint64_t &operator[](std::size_t i) { return realize(), data_.at(i); }
  const std::uint64_t &operator[](std::size_t i) const { return realize(), data_.at(i); }

  // facade: element `index` of a device-resident batch (see facade::LazyPlain)
  void facade_set_lazy(std::shared_ptr<const facade::LazyPlain> batch, std::size_t index) {
    data_.clear(), parms_id_ = parms_id_zero, scale_ = 1.0;
    lazy_ = std::move(batch), lazy_index_ = index;
  }
  const std::shared_ptr<const facade::LazyPlain> &facade_lazy() const { return lazy_; }
  std::size_t facade_lazy_index() const { return lazy_index_; }

 private:
  void realize() const {  // (not thread-safe on one object, like any first write)
    if (!lazy_) return;
    auto batch = std::move(lazy_);
    lazy_.reset();
    batch->materialize(lazy_index_, data_);
  }
  parms_id_type parms_id_ = parms_id_zero;
  double scale_ = 1.0;
  mutable std::vector<std::uint64_t> data_;
  mutable std::shared_ptr<const facade::LazyPlain> lazy_;
  std::size_t lazy_index_ = 0;
};

class Ciphertext {
 public:
  Ciphertext() = default;
  explicit Ciphertext(MemoryPoolHandle) {}
  explicit Ciphertext(const SEALContext &context) { resize(context, context.first_parms_id(), 2); }
  void resize(const SEALContext &context, const parms_id_type &parms_id, std::size_t size) {
    auto cd = context.get_context_data(parms_id);
    if (!cd) throw std::invalid_argument("parms_id is not valid for encryption parameters");
    parms_id_ = parms_id;
    size_ = size;
    seeded_ = pending_ = false;
    n_ = cd->parms().poly_modulus_degree();
    l_ = cd->parms().coeff_modulus().size();
    data_.resize(size_ * n_ * l_, 0);
  }
  void resize(std::size_t size) { expand_pending(), size_ = size, seeded_ = false, data_.resize(size_ * n_ * l_, 0); }
  void release() {
    parms_id_ = parms_id_zero, is_ntt_form_ = seeded_ = pending_ = false, size_ = n_ = l_ = 0, scale_ = 1.0;
    data_.clear(), data_.shrink_to_fit();
  }
  // (a writable view ends the seeded state: c1 may no longer be what the seed expands to)
  std::uint64_t *data() { return expand_pending(), seeded_ = false, data_.data(); }
  const std::uint64_t *data() const { return expand_pending(), data_.data(); }
  std::uint64_t *data(std::size_t k) {
    if (k >= size_) throw std::out_of_range("poly_index must be within [0, size)");
    expand_pending();
    seeded_ = false;
    return data_.data() + k * n_ * l_;
  }
  const std::uint64_t *data(std::size_t k) const {
    if (k >= size_) throw std::out_of_range("poly_index must be within [0, size)");
    expand_pending();
    return data_.data() + k * n_ * l_;
  }
  std::size_t size() const { return size_; }
  std::size_t poly_modulus_degree() const { return n_; }
  std::size_t coeff_modulus_size() const { return l_; }
  parms_id_type &parms_id() { return parms_id_; }
  const parms_id_type &parms_id() const { return parms_id_; }
  bool &is_ntt_form() { return is_ntt_form_; }
  bool is_ntt_form() const { return is_ntt_form_; }
  double &scale() { return scale_; }
  const double &scale() const { return scale_; }
  bool is_transparent() const {
    expand_pending();
    return size_ < 2 || std::all_of(data_.begin() + n_ * l_, data_.end(), [](std::uint64_t c) { return c == 0; });
  }

  // SEAL 3.7's serialisation with compr_mode none (seal/wire.h; parity unpinned).  A ciphertext that
  // came out of encrypt_symmetric as a Serializable carries the seed of its c1 and is written in the seeded form
  // (c0 + UniformRandomGeneratorInfo): half the bytes, and the loader regenerates c1 with SEAL's Blake2xb expansion.
  std::streamoff save(std::ostream &os) const {
    expand_pending();
    wire::CtMeta m;
    m.parms_id = parms_id_, m.is_ntt_form = is_ntt_form_, m.size = size_, m.poly_modulus_degree = n_;
    m.coeff_modulus_size = l_, m.scale = scale_;
    const std::string bytes = wire::save_ciphertext(m, data_.data(), seeded_ ? seed_.data() : nullptr);
    os.write(bytes.data(), (std::streamsize)bytes.size());
    return (std::streamoff)bytes.size();
  }
  std::streamoff unsafe_load(const SEALContext &context, std::istream &is) {
    std::uint8_t head[wire::kHeaderBytes];
    is.read(reinterpret_cast<char *>(head), sizeof(head));
    if (!is) throw std::logic_error("Ciphertext::load: truncated stream");
    const std::size_t total = wire::peek_total(head);
    if (total < sizeof(head) || total > (std::size_t(1) << 32)) throw std::logic_error("Ciphertext::load: bad size field");
    std::vector<std::uint8_t> bytes(total);
    std::memcpy(bytes.data(), head, sizeof(head));
    is.read(reinterpret_cast<char *>(bytes.data() + sizeof(head)), (std::streamsize)(total - sizeof(head)));
    if (!is) throw std::logic_error("Ciphertext::load: truncated stream");
    wire::CtMeta m;
    std::vector<std::uint64_t> words;
    bool seeded = false;
    std::uint8_t seed[wire::kSeedBytes];
    wire::load_ciphertext(bytes.data(), bytes.size(), m, words, seeded, seed);
    auto cd = context.get_context_data(m.parms_id);
    if (!cd) throw std::logic_error("ciphertext data is invalid");   // parms_id unknown to this context
    resize(context, m.parms_id, (std::size_t)m.size);
    if (n_ != m.poly_modulus_degree || l_ != m.coeff_modulus_size) throw std::logic_error("ciphertext data is invalid");
    is_ntt_form_ = m.is_ntt_form, scale_ = m.scale;
    data_ = std::move(words);
    seeded_ = pending_ = false;
    if (seeded) {
      // Ciphertext::expand_seed, deferred: c1 is regenerated from the seed the first time somebody looks at the data
      // (SEAL expands inside load).  cheetah-api.cpp loads a layer's ciphertexts one by one on one thread
      // (:710-737); the batched protocol classes expand all of them side by side (facade_expand_now from worker
      // threads) before they upload -- same polynomial either way.
      pending_q_.clear();
      for (const auto &mod : cd->parms().coeff_modulus()) pending_q_.push_back(mod.value());
      std::memcpy(seed_.data(), seed, wire::kSeedBytes);
      pending_ = true;
    }
    return (std::streamoff)total;
  }
  std::streamoff load(const SEALContext &context, std::istream &is) {
    const std::streamoff n = unsafe_load(context, is);
    auto cd = context.get_context_data(parms_id_);
    const auto &q = cd->parms().coeff_modulus();
    // is_valid_for: every coefficient below its modulus (a polynomial still to be expanded from its seed is reduced
    // by construction)
    for (std::size_t k = 0; k < (pending_ ? 1 : size_); ++k)
      for (std::size_t l = 0; l < l_; ++l) {
        const std::uint64_t *p = data_.data() + (k * l_ + l) * n_, m = q[l].value();
        if (std::any_of(p, p + n_, [m](std::uint64_t c) { return c >= m; }))
          throw std::logic_error("ciphertext data is invalid");
      }
    return n;
  }
  // facade: the seed c1 was expanded from (set by Encryptor::encrypt_symmetric's Serializable form)
  void facade_set_seed(const std::uint8_t (&seed)[wire::kSeedBytes]) { std::memcpy(seed_.data(), seed, wire::kSeedBytes), seeded_ = true; }
  bool facade_seeded() const { return seeded_; }
  // facade: a loaded seeded ciphertext whose c1 has not been expanded yet; facade_expand_now does it (safe to call
  // for different objects from different threads)
  bool facade_pending() const { return pending_; }
  void facade_expand_now() const { expand_pending(); }

 private:
  void expand_pending() const {
    if (!pending_) return;
    std::uint8_t seed[wire::kSeedBytes];
    std::memcpy(seed, seed_.data(), sizeof(seed));
    wire::Blake2xbPRNG prng(seed);
    wire::sample_poly_uniform(prng, pending_q_, n_, data_.data() + n_ * l_);
    pending_ = false;
  }
  parms_id_type parms_id_ = parms_id_zero;
  bool is_ntt_form_ = false, seeded_ = false;
  mutable bool pending_ = false;
  std::vector<std::uint64_t> pending_q_;
  std::array<std::uint8_t, wire::kSeedBytes> seed_{};
  std::size_t size_ = 0, n_ = 0, l_ = 0;
  double scale_ = 1.0;
  mutable std::vector<std::uint64_t> data_;
};

// SEAL's Serializable<T> defers seed expansion to save(); here it simply owns the object
template <class T>
class Serializable {
 public:
  Serializable(const T &obj) : obj_(obj) {}
  Serializable(T &&obj) : obj_(std::move(obj)) {}
  Serializable(const Serializable &) = default;
  Serializable &operator=(const Serializable &) = default;
  std::streamoff save(std::ostream &os) const { return obj_.save(os); }
  const T &facade_object() const { return obj_; }  // facade: what load() would give back

 private:
  T obj_;
};

// ------------------------------------------------------------------------------------------ keys
class SecretKey {
 public:
  Plaintext &data() { return sk_; }  // NTT form at the key level, [L][N]
  const Plaintext &data() const { return sk_; }
  parms_id_type &parms_id() { return sk_.parms_id(); }
  const parms_id_type &parms_id() const { return sk_.parms_id(); }

 private:
  Plaintext sk_;
};
class PublicKey {
 public:
  Ciphertext &data() { return pk_; }  // (-(a s + e), a) in NTT form at the key level
  const Ciphertext &data() const { return pk_; }
  parms_id_type &parms_id() { return pk_.parms_id(); }
  const parms_id_type &parms_id() const { return pk_.parms_id(); }
  std::streamoff save(std::ostream &os) const { return pk_.save(os); }
  std::streamoff load(const SEALContext &context, std::istream &is) { return pk_.load(context, is); }

 private:
  Ciphertext pk_;
};
class KSwitchKeys {};
class GaloisKeys : public KSwitchKeys {};
class RelinKeys : public KSwitchKeys {};

inline bool is_metadata_valid_for(const SecretKey &sk, const SEALContext &c) {
  return sk.parms_id() == c.key_parms_id() && sk.data().coeff_count() != 0;
}
inline bool is_metadata_valid_for(const PublicKey &pk, const SEALContext &c) {
  return pk.parms_id() == c.key_parms_id() && pk.data().size() == 2 && pk.data().is_ntt_form();
}
inline bool is_metadata_valid_for(const Ciphertext &ct, const SEALContext &c) {
  return c.get_context_data(ct.parms_id()) != nullptr && ct.size() >= 2;
}
inline bool is_valid_for(const Ciphertext &ct, const SEALContext &c) {
  auto cd = c.get_context_data(ct.parms_id());
  if (!cd || ct.size() < 2) return false;
  const auto &q = cd->parms().coeff_modulus();
  const std::size_t n = ct.poly_modulus_degree();
  for (std::size_t k = 0; k < ct.size(); ++k)
    for (std::size_t l = 0; l < q.size(); ++l) {
      const std::uint64_t *p = ct.data(k) + l * n, m = q[l].value();
      if (std::any_of(p, p + n, [m](std::uint64_t v) { return v >= m; })) return false;
    }
  return true;
}

// ------------------------------------------------------------------------------------------ seal::util
namespace util {
template <typename T>
inline T mul_safe(T a, T b) {
  T r;
  if (__builtin_mul_overflow(a, b, &r)) throw std::logic_error("unsigned overflow");
  return r;
}
inline std::size_t mul_safe(std::size_t a, std::size_t b) { return mul_safe<std::size_t>(a, b); }
inline void seal_memzero(void *p, std::size_t n) {
  volatile unsigned char *v = static_cast<volatile unsigned char *>(p);
  while (n--) *v++ = 0;
}
inline bool are_close(double a, double b) { return std::fabs(a - b) < std::max(std::fabs(a), std::fabs(b)) * 2.220446049250313e-16 * 4; }

// scalar helpers (one value each; no polynomial work happens on the host)
inline std::uint64_t barrett_reduce_64(std::uint64_t x, const Modulus &m) {
  const std::uint64_t q = (std::uint64_t)(((unsigned __int128)x * m.const_ratio()[1]) >> 64);
  const std::uint64_t r = x - q * m.value();
  return r >= m.value() ? r - m.value() : r;
}
inline std::uint64_t barrett_reduce_128(const std::uint64_t *x, const Modulus &m) {
  return (std::uint64_t)((((unsigned __int128)x[1] << 64) | x[0]) % m.value());
}
inline std::uint64_t multiply_uint_mod(std::uint64_t a, std::uint64_t b, const Modulus &m) {
  return (std::uint64_t)((unsigned __int128)a * b % m.value());
}
inline std::uint64_t negate_uint_mod(std::uint64_t a, const Modulus &m) { return a ? m.value() - a : 0; }
inline unsigned char add_uint(const std::uint64_t *a, std::size_t na, const std::uint64_t *b, std::size_t nb,
                              unsigned char carry, std::size_t nr, std::uint64_t *r) {
  for (std::size_t i = 0; i < nr; ++i) {
    const unsigned __int128 s = (unsigned __int128)(i < na ? a[i] : 0) + (i < nb ? b[i] : 0) + carry;
    r[i] = (std::uint64_t)s;
    carry = (unsigned char)(s >> 64);
  }
  return carry;
}

struct PolyBuffer {  // what allocate_poly / allocate_uint return: owning, .get()
  std::unique_ptr<std::uint64_t[]> p;
  std::uint64_t *get() const { return p.get(); }
  std::uint64_t &operator[](std::size_t i) const { return p[i]; }
};
inline PolyBuffer allocate_uint(std::size_t n, MemoryPoolHandle) { return PolyBuffer{std::unique_ptr<std::uint64_t[]>(new std::uint64_t[n]())}; }
inline PolyBuffer allocate_poly(std::size_t coeff_count, std::size_t coeff_modulus_size, MemoryPoolHandle pool) {
  return allocate_uint(mul_safe(coeff_count, coeff_modulus_size), pool);
}

namespace detail {
inline void poly_op(int op, const std::uint64_t *a, const std::uint64_t *b, std::uint64_t scalar, std::size_t n,
                    const Modulus &m, std::uint64_t *out) {
  if (n == 0) return;
  facade::DBuf da(n), db(b ? n : 1);
  da.up(a, n);
  if (b) db.up(b, n);
  facade::check(mmh_poly_op(op, da.get(), b ? db.get() : nullptr, scalar, n, m.value(), da.get(), nullptr), "poly_op");
  da.down(out, n);
}
inline void ntt(std::uint64_t *operand, const NTTTables &t, bool inverse) {
  const std::size_t N = t.coeff_count();
  facade::DBuf d(N);
  d.up(operand, N);
  facade::check(inverse ? mmh_ntt_inverse_limb(t.backend().ctx, d.get(), 1, t.limb(), nullptr)
                        : mmh_ntt_forward_limb(t.backend().ctx, d.get(), 1, t.limb(), nullptr),
                "ntt_negacyclic_harvey");
  d.down(operand, N);
}
}  // namespace detail

// polyarithsmallmod.h
inline void add_poly_coeffmod(const std::uint64_t *a, const std::uint64_t *b, std::size_t n, const Modulus &m, std::uint64_t *r) {
  detail::poly_op(0, a, b, 0, n, m, r);
}
inline void sub_poly_coeffmod(const std::uint64_t *a, const std::uint64_t *b, std::size_t n, const Modulus &m, std::uint64_t *r) {
  detail::poly_op(1, a, b, 0, n, m, r);
}
inline void dyadic_product_coeffmod(const std::uint64_t *a, const std::uint64_t *b, std::size_t n, const Modulus &m, std::uint64_t *r) {
  detail::poly_op(2, a, b, 0, n, m, r);
}
inline void multiply_poly_scalar_coeffmod(const std::uint64_t *a, std::size_t n, std::uint64_t scalar, const Modulus &m,
                                          std::uint64_t *r) {
  detail::poly_op(3, a, nullptr, barrett_reduce_64(scalar, m), n, m, r);
}
inline void modulo_poly_coeffs(const std::uint64_t *a, std::size_t n, const Modulus &m, std::uint64_t *r) {
  detail::poly_op(4, a, nullptr, 0, n, m, r);
}
// ntt.h.  The lazy variant may leave values in [0, 4q) in SEAL; every consumer reduces, so the reduced result of
// the device transform is interchangeable.
inline void ntt_negacyclic_harvey(std::uint64_t *operand, const NTTTables &t) { detail::ntt(operand, t, false); }
inline void ntt_negacyclic_harvey_lazy(std::uint64_t *operand, const NTTTables &t) { detail::ntt(operand, t, false); }
inline void inverse_ntt_negacyclic_harvey(std::uint64_t *operand, const NTTTables &t) { detail::ntt(operand, t, true); }
inline void inverse_ntt_negacyclic_harvey_lazy(std::uint64_t *operand, const NTTTables &t) { detail::ntt(operand, t, true); }

// rlwe.h: samplers writing RNS form [L][N] for the moduli of `parms`
namespace detail {
inline void signed_to_rns(const std::int64_t *v, const EncryptionParameters &parms, std::uint64_t *dst) {
  const std::size_t N = parms.poly_modulus_degree();
  const auto &q = parms.coeff_modulus();
  for (std::size_t l = 0; l < q.size(); ++l)
    for (std::size_t i = 0; i < N; ++i) dst[l * N + i] = v[i] >= 0 ? (std::uint64_t)v[i] : q[l].value() - (std::uint64_t)(-v[i]);
}
inline void sample_signed(bool ternary, std::shared_ptr<UniformRandomGenerator> prng, const EncryptionParameters &parms,
                          std::uint64_t *dst) {
  const std::size_t N = parms.poly_modulus_degree();
  std::uint32_t key[8];
  std::uint64_t sid;
  (prng ? prng : parms.random_generator()->create())->sampler_key(key, &sid);
  facade::DBuf d(N);
  facade::check(ternary ? mmh_sample_ternary(key, sid, (std::int64_t *)d.get(), N, nullptr)
                        : mmh_sample_noise(key, sid, (std::int64_t *)d.get(), N, nullptr),
                "sample_poly");
  std::vector<std::int64_t> v(N);
  d.down(v.data(), N);
  signed_to_rns(v.data(), parms, dst);
}
}  // namespace detail
inline void sample_poly_ternary(std::shared_ptr<UniformRandomGenerator> prng, const EncryptionParameters &parms, std::uint64_t *dst) {
  detail::sample_signed(true, std::move(prng), parms, dst);
}
inline void sample_poly_cbd(std::shared_ptr<UniformRandomGenerator> prng, const EncryptionParameters &parms, std::uint64_t *dst) {
  detail::sample_signed(false, std::move(prng), parms, dst);
}
#define SEAL_NOISE_SAMPLER sample_poly_cbd
// uniform in [0, q_l) per limb.  Needs the level's device context, which a ContextData's parms do not carry; the
// moduli identify it: rejection sampling from the keystream on the device, one call per limb.
inline void sample_poly_uniform(std::shared_ptr<UniformRandomGenerator> prng, const EncryptionParameters &parms, std::uint64_t *dst) {
  const std::size_t N = parms.poly_modulus_degree();
  const auto &q = parms.coeff_modulus();
  auto g = prng ? prng : parms.random_generator()->create();
  // 64-bit words reduced by rejection: a word is kept when below the largest multiple of q_l (as SEAL does)
  std::vector<std::uint64_t> words(N);
  for (std::size_t l = 0; l < q.size(); ++l) {
    const std::uint64_t m = q[l].value(), limit = ~std::uint64_t(0) - (~std::uint64_t(0) % m) - 1;
    std::size_t have = 0;
    while (have < N) {
      g->generate((N - have) * 8, reinterpret_cast<seal_byte *>(words.data()));
      for (std::size_t i = 0; i < N - have; ++i)
        if (words[i] <= limit) dst[l * N + have++] = words[i];
      // (values are reduced below on the device)
    }
    modulo_poly_coeffs(dst + l * N, N, q[l], dst + l * N);
  }
}
}  // namespace util

// ------------------------------------------------------------------------------------------ KeyGenerator
class KeyGenerator {
 public:
  explicit KeyGenerator(const SEALContext &context) : context_(context) {
    auto cd = context_.key_context_data();
    const auto &parms = cd->parms();
    const std::size_t N = parms.poly_modulus_degree(), L = parms.coeff_modulus().size();
    std::uint32_t key[8];
    std::uint64_t sid;
    parms.random_generator()->create()->sampler_key(key, &sid);
    facade::DBuf s(N), sk(L * N);
    facade::check(mmh_sample_ternary(key, sid, (std::int64_t *)s.get(), N, nullptr), "KeyGenerator");
    facade::check(mmh_signed_to_ntt(cd->backend()->ctx, 0, (const std::int64_t *)s.get(), 1, sk.get(), nullptr), "KeyGenerator");
    sk_.data().resize(L * N);
    sk.down(sk_.data().data(), L * N);
    sk_.parms_id() = cd->parms_id();
  }
  const SecretKey &secret_key() const { return sk_; }
  Serializable<PublicKey> create_public_key() const {
    PublicKey pk;
    create_public_key(pk);
    return Serializable<PublicKey>(std::move(pk));
  }
  void create_public_key(PublicKey &pk) const {
    auto cd = context_.key_context_data();
    const auto &parms = cd->parms();
    const std::size_t N = parms.poly_modulus_degree(), L = parms.coeff_modulus().size();
    auto prng = parms.random_generator()->create();
    std::uint32_t key[8];
    std::uint64_t sid;
    facade::DBuf sk(L * N), a(L * N), e(N), out(2 * L * N);
    sk.up(sk_.data().data(), L * N);
    prng->sampler_key(key, &sid);
    facade::check(mmh_sample_uniform(cd->backend()->ctx, 0, key, sid, a.get(), 1, nullptr), "create_public_key");
    prng->sampler_key(key, &sid);
    facade::check(mmh_sample_noise(key, sid, (std::int64_t *)e.get(), N, nullptr), "create_public_key");
    facade::check(mmh_make_public_key(cd->backend()->ctx, sk.get(), a.get(), (const std::int64_t *)e.get(), out.get(), nullptr),
                  "create_public_key");
    pk.data().resize(context_, cd->parms_id(), 2);
    out.down(pk.data().data(), 2 * L * N);
    pk.data().is_ntt_form() = true;
  }

 private:
  SEALContext context_;
  SecretKey sk_;
};

// ------------------------------------------------------------------------------------------ Encryptor / Decryptor
class Encryptor {
 public:
  Encryptor(const SEALContext &context, const PublicKey &pk) : context_(context) { set_public_key(pk); }
  Encryptor(const SEALContext &context, const SecretKey &sk) : context_(context) { set_secret_key(sk); }
  Encryptor(const SEALContext &context, const PublicKey &pk, const SecretKey &sk) : context_(context) {
    set_public_key(pk), set_secret_key(sk);
  }
  void set_public_key(const PublicKey &pk) {
    if (!is_metadata_valid_for(pk, context_)) throw std::invalid_argument("public key is not valid for encryption parameters");
    pk_ = pk, has_pk_ = true;
  }
  void set_secret_key(const SecretKey &sk) {
    if (!is_metadata_valid_for(sk, context_)) throw std::invalid_argument("secret key is not valid for encryption parameters");
    sk_ = sk, has_sk_ = true;
  }
  Serializable<Ciphertext> encrypt_symmetric(const Plaintext &plain) const {
    Ciphertext ct;
    symmetric(&plain, ct, true);
    return Serializable<Ciphertext>(std::move(ct));
  }
  void encrypt_symmetric(const Plaintext &plain, Ciphertext &destination) const { symmetric(&plain, destination); }
  Serializable<Ciphertext> encrypt_zero_symmetric() const {
    Ciphertext ct;
    symmetric(nullptr, ct, true);
    return Serializable<Ciphertext>(std::move(ct));
  }
  // with a public key: asymmetric; an encryptor that only holds a secret key answers with a symmetric encryption
  // (the reference builds its "dummy" from such an encryptor, hom_conv2d_ss.cc:480,:512)
  Serializable<Ciphertext> encrypt_zero() const {
    Ciphertext ct;
    encrypt_zero(context_.first_parms_id(), ct);
    return Serializable<Ciphertext>(std::move(ct));
  }
  void encrypt_zero(Ciphertext &destination) const { encrypt_zero(context_.first_parms_id(), destination); }
  void encrypt_zero(const parms_id_type &parms_id, Ciphertext &destination) const {
    auto cd = context_.get_context_data(parms_id);
    if (!cd) throw std::invalid_argument("parms_id is not valid for encryption parameters");
    if (!has_pk_) {
      if (!has_sk_) throw std::logic_error("public key is not set");
      symmetric(nullptr, destination);
      if (cd->level() == 1 && context_.first_context_data()->level() == 0) switch_to_last(destination);
      return;
    }
    // Encryptor::encrypt_zero_internal: at a lower level the encryption is made one level up and divided by the
    // dropped prime (RNSTool::divide_and_round_q_last)
    auto top = context_.key_context_data();
    const auto &parms = top->parms();
    const std::size_t N = parms.poly_modulus_degree(), L = parms.coeff_modulus().size();
    auto prng = parms.random_generator()->create();
    std::uint32_t key[8];
    std::uint64_t sid;
    facade::DBuf pk(2 * L * N), u(N), e0(N), e1(N), out(2 * L * N);
    pk.up(pk_.data().data(), 2 * L * N);
    prng->sampler_key(key, &sid);
    facade::check(mmh_sample_ternary(key, sid, (std::int64_t *)u.get(), N, nullptr), "encrypt_zero");
    prng->sampler_key(key, &sid);
    facade::check(mmh_sample_noise(key, sid, (std::int64_t *)e0.get(), N, nullptr), "encrypt_zero");
    prng->sampler_key(key, &sid);
    facade::check(mmh_sample_noise(key, sid, (std::int64_t *)e1.get(), N, nullptr), "encrypt_zero");
    facade::check(mmh_encrypt_zero_asym(top->backend()->ctx, 0, pk.get(), (const std::int64_t *)u.get(),
                                        (const std::int64_t *)e0.get(), (const std::int64_t *)e1.get(), 1, out.get(), nullptr),
                  "encrypt_zero");
    destination.resize(context_, top->parms_id(), 2);
    out.down(destination.data(), 2 * L * N);
    destination.is_ntt_form() = false;
    destination.scale() = 1.0;
    if (cd->level() == 1 && top->level() == 0) switch_to_last(destination);
  }

 private:
  // save_seed: the Serializable form.  c1 = a is then expanded on the host from a fresh 64-byte seed exactly as a
  // SEAL loader will expand it again (wire::Blake2xbPRNG + sample_poly_uniform), so that save() can ship c0 + seed.
  void symmetric(const Plaintext *plain, Ciphertext &destination, bool save_seed = false) const {
    if (!has_sk_) throw std::logic_error("secret key is not set");
    auto cd = context_.first_context_data();
    const auto &parms = cd->parms();
    const std::size_t N = parms.poly_modulus_degree(), L = parms.coeff_modulus().size();
    if (plain && (plain->is_ntt_form() || plain->coeff_count() > N)) throw std::invalid_argument("plain is not valid for encryption parameters");
    auto prng = parms.random_generator()->create();
    std::uint32_t key[8];
    std::uint64_t sid;
    facade::DBuf sk(L * N), a(L * N), e(N), pt(N), out(2 * L * N);
    sk.up(sk_.data().data(), L * N);
    std::uint8_t seed[wire::kSeedBytes];
    if (save_seed) {
      prng->generate(sizeof(seed), reinterpret_cast<seal_byte *>(seed));
      std::vector<std::uint64_t> q, host_a(L * N);
      for (const auto &m : parms.coeff_modulus()) q.push_back(m.value());
      wire::Blake2xbPRNG expand(seed);
      wire::sample_poly_uniform(expand, q, N, host_a.data());
      a.up(host_a.data(), L * N);
    } else {
      prng->sampler_key(key, &sid);
      facade::check(mmh_sample_uniform(cd->backend()->ctx, 0, key, sid, a.get(), 1, nullptr), "encrypt_symmetric");
    }
    prng->sampler_key(key, &sid);
    facade::check(mmh_sample_noise(key, sid, (std::int64_t *)e.get(), N, nullptr), "encrypt_symmetric");
    if (plain) {
      pt.zero();
      pt.up(plain->data(), plain->coeff_count());
    }
    facade::check(mmh_encrypt_symmetric(cd->backend()->ctx, 0, sk.get(), a.get(), (const std::int64_t *)e.get(),
                                        plain ? pt.get() : nullptr, 1, out.get(), nullptr),
                  "encrypt_symmetric");
    destination.resize(context_, cd->parms_id(), 2);
    out.down(destination.data(), 2 * L * N);
    destination.is_ntt_form() = false;
    destination.scale() = 1.0;
    if (save_seed) destination.facade_set_seed(seed);
  }
  void switch_to_last(Ciphertext &ct) const {
    auto first = context_.first_context_data();
    const std::size_t N = first->parms().poly_modulus_degree(), L = first->parms().coeff_modulus().size();
    facade::DBuf in(2 * L * N), out(2 * N);
    in.up(ct.data(), 2 * L * N);
    facade::check(mmh_mod_switch_to_last(first->backend()->ctx, in.get(), 1, out.get(), nullptr), "mod_switch_to_inplace");
    ct.resize(context_, context_.last_parms_id(), 2);
    out.down(ct.data(), 2 * N);
  }
  SEALContext context_;
  PublicKey pk_;
  SecretKey sk_;
  bool has_pk_ = false, has_sk_ = false;
};

class Decryptor {
 public:
  Decryptor(const SEALContext &context, const SecretKey &sk) : context_(context), sk_(sk) {
    if (!is_metadata_valid_for(sk, context_)) throw std::invalid_argument("secret key is not valid for encryption parameters");
  }
  void decrypt(const Ciphertext &encrypted, Plaintext &destination) const {
    auto cd = context_.get_context_data(encrypted.parms_id());
    if (!cd || encrypted.size() != 2) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    if (encrypted.is_ntt_form()) throw std::invalid_argument("encrypted cannot be in NTT form");
    const std::size_t N = encrypted.poly_modulus_degree(), L = encrypted.coeff_modulus_size();
    facade::DBuf sk(L * N), ct(2 * L * N), pt(N);
    sk.up(sk_.data().data(), L * N);  // the first L limbs of the key-level secret: the key restricted to this level
    ct.up(encrypted.data(), 2 * L * N);
    facade::check(mmh_decrypt(cd->backend()->ctx, cd->level(), sk.get(), ct.get(), 1, pt.get(), nullptr), "decrypt");
    destination.parms_id() = parms_id_zero;
    destination.resize(N);
    pt.down(destination.data(), N);
    // Decryptor::bfv_decrypt trims leading zero coefficients (the reference guards with `idx >= n ? 0 : pt[idx]`,
    // hom_conv2d_ss.cc:965)
    destination.resize(std::max<std::size_t>(destination.significant_coeff_count(), 1));
  }

 private:
  SEALContext context_;
  SecretKey sk_;
};

// ------------------------------------------------------------------------------------------ Evaluator
class Evaluator {
 public:
  explicit Evaluator(const SEALContext &context) : context_(context) {}
  // facade: state a batched implementation of the protocol classes keeps with the evaluator it was given (stream,
  // device copies of the keys, layer plans; openmm_amd/dropin) -- lives and dies with the object that owns the evaluator
  std::shared_ptr<void> &facade_ext() const { return ext_; }

  void add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const { plain_op(encrypted, plain, false); }
  void add_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const {
    destination = encrypted;
    add_plain_inplace(destination, plain);
  }
  void sub_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const { plain_op(encrypted, plain, true); }
  void sub_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const {
    destination = encrypted;
    sub_plain_inplace(destination, plain);
  }

  void multiply_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const {
    auto cd = level_of(encrypted);
    if (encrypted.is_ntt_form() || plain.is_ntt_form()) throw std::invalid_argument("NTT form mismatch");
    const std::size_t N = encrypted.poly_modulus_degree(), L = encrypted.coeff_modulus_size();
    if (plain.coeff_count() > N) throw std::invalid_argument("plain is not valid for encryption parameters");
    if (plain.is_zero()) throw std::logic_error("result ciphertext is transparent");  // SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT
    facade::DBuf ct(2 * L * N), pt(N), out(2 * L * N);
    ct.up(encrypted.data(), 2 * L * N);
    pt.zero();
    pt.up(plain.data(), plain.coeff_count());
    // one term of the multiply-accumulate the layer calls batch: NTT(ct), lift + NTT(pt), dyadic product, inverse NTT
    facade::check(mmh_mul_plain_acc(cd->backend()->ctx, cd->level(), ct.get(), 1, pt.get(), 1, 1, 1, 1, 0, 0, 0, 0, 0, out.get(),
                                    nullptr),
                  "multiply_plain");
    out.down(encrypted.data(), 2 * L * N);
  }
  void multiply_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const {
    destination = encrypted;
    multiply_plain_inplace(destination, plain);
  }

  void add_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const {
    auto cd = level_of(encrypted1);
    if (encrypted1.parms_id() != encrypted2.parms_id() || encrypted1.is_ntt_form() != encrypted2.is_ntt_form() ||
        encrypted1.size() != 2 || encrypted2.size() != 2)
      throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    const std::size_t words = 2 * encrypted1.coeff_modulus_size() * encrypted1.poly_modulus_degree();
    facade::DBuf a(words), b(words);
    a.up(encrypted1.data(), words);
    b.up(encrypted2.data(), words);
    facade::check(mmh_ct_add(cd->backend()->ctx, cd->level(), a.get(), b.get(), 1, nullptr), "add");
    a.down(encrypted1.data(), words);
  }
  void add(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const {
    if (&destination == &encrypted2) {
      Ciphertext tmp = encrypted1;
      add_inplace(tmp, encrypted2);
      destination = std::move(tmp);
    } else {
      destination = encrypted1;
      add_inplace(destination, encrypted2);
    }
  }

  void mod_switch_to_inplace(Ciphertext &encrypted, const parms_id_type &parms_id) const {
    auto cd = level_of(encrypted);
    auto target = context_.get_context_data(parms_id);
    if (!target) throw std::invalid_argument("parms_id is not valid for encryption parameters");
    if (cd->level() > target->level() || (cd->chain_index() < target->chain_index()))
      throw std::invalid_argument("cannot switch to higher level modulus");
    if (cd->parms_id() == target->parms_id()) return;
    if (encrypted.is_ntt_form()) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
    const std::size_t N = encrypted.poly_modulus_degree(), L = encrypted.coeff_modulus_size();
    facade::DBuf in(2 * L * N), out(2 * N);
    in.up(encrypted.data(), 2 * L * N);
    facade::check(mmh_mod_switch_to_last(cd->backend()->ctx, in.get(), 1, out.get(), nullptr), "mod_switch_to_inplace");
    const bool ntt = encrypted.is_ntt_form();
    const double scale = encrypted.scale();
    encrypted.resize(context_, parms_id, 2);
    out.down(encrypted.data(), 2 * N);
    encrypted.is_ntt_form() = ntt;
    encrypted.scale() = scale;
  }
  void mod_switch_to(const Ciphertext &encrypted, const parms_id_type &parms_id, Ciphertext &destination) const {
    destination = encrypted;
    mod_switch_to_inplace(destination, parms_id);
  }

  void transform_to_ntt_inplace(Ciphertext &encrypted) const { transform(encrypted, false); }
  void transform_from_ntt_inplace(Ciphertext &encrypted_ntt) const { transform(encrypted_ntt, true); }

 private:
  std::shared_ptr<const SEALContext::ContextData> level_of(const Ciphertext &ct) const {
    auto cd = context_.get_context_data(ct.parms_id());
    if (!cd || ct.size() < 2) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    return cd;
  }
  void plain_op(Ciphertext &encrypted, const Plaintext &plain, bool sub) const {
    auto cd = level_of(encrypted);
    if (encrypted.is_ntt_form() || plain.is_ntt_form()) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
    const std::size_t N = encrypted.poly_modulus_degree(), L = encrypted.coeff_modulus_size();
    if (plain.coeff_count() > N) throw std::invalid_argument("plain is not valid for encryption parameters");
    facade::DBuf ct(2 * L * N), pt(N);
    ct.up(encrypted.data(), 2 * L * N);
    pt.zero();
    pt.up(plain.data(), plain.coeff_count());
    facade::check(mmh_add_plain(cd->backend()->ctx, cd->level(), ct.get(), pt.get(), 1, sub ? 1 : 0, ct.get(), nullptr),
                  sub ? "sub_plain" : "add_plain");
    ct.down(encrypted.data(), 2 * L * N);
  }
  void transform(Ciphertext &ct, bool inverse) const {
    auto cd = level_of(ct);
    if (ct.is_ntt_form() != inverse) throw std::invalid_argument(inverse ? "encrypted_ntt is not in NTT form" : "encrypted is already in NTT form");
    const std::size_t N = ct.poly_modulus_degree(), L = ct.coeff_modulus_size(), words = ct.size() * L * N;
    facade::DBuf d(words);
    d.up(ct.data(), words);
    facade::check(inverse ? mmh_ntt_inverse(cd->backend()->ctx, d.get(), ct.size(), L, nullptr)
                          : mmh_ntt_forward(cd->backend()->ctx, d.get(), ct.size(), L, nullptr),
                  "transform_ntt");
    d.down(ct.data(), words);
    ct.is_ntt_form() = !inverse;
  }
  SEALContext context_;
  mutable std::shared_ptr<void> ext_;
};

// ------------------------------------------------------------------------------------------ BatchEncoder
class BatchEncoder {
 public:
  explicit BatchEncoder(const SEALContext &context) : context_(context) {
    if (!context_.first_context_data()->qualifiers().using_batching)
      throw std::invalid_argument("encryption parameters are not valid for batching");
    slots_ = context_.first_context_data()->parms().poly_modulus_degree();
  }
  std::size_t slot_count() const { return slots_; }
  void encode(const std::vector<std::uint64_t> &values, Plaintext &destination) const {
    if (values.size() > slots_) throw std::invalid_argument("values has invalid size");
    auto cd = context_.first_context_data();
    facade::DBuf v(slots_), p(slots_);
    v.zero();
    v.up(values.data(), values.size());
    facade::check(mmh_batch_encode(cd->backend()->ctx, v.get(), 1, p.get(), nullptr), "BatchEncoder::encode");
    destination.parms_id() = parms_id_zero;
    destination.resize(slots_);
    p.down(destination.data(), slots_);
  }
  void decode(const Plaintext &plain, std::vector<std::uint64_t> &destination) const {
    if (plain.is_ntt_form()) throw std::invalid_argument("plain cannot be in NTT form");
    auto cd = context_.first_context_data();
    facade::DBuf p(slots_), v(slots_);
    p.zero();
    p.up(plain.data(), std::min(plain.coeff_count(), slots_));
    facade::check(mmh_batch_decode(cd->backend()->ctx, p.get(), 1, v.get(), nullptr), "BatchEncoder::decode");
    destination.resize(slots_);
    v.down(destination.data(), slots_);
  }

 private:
  SEALContext context_;
  std::size_t slots_ = 0;
};

}  // namespace seal
