// Internal helpers of the host-side class mirrors (hom_*_ss.cc): error plumbing in the style of the reference's
// ENSURE_OR_RETURN (include/gemini/core/common.h:10-22), a call-scoped device buffer, key upload.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <vector>

#include "mm_gemini/core.h"
#include "mm_gemini/he_types.h"
#include "mm_hip.h"

namespace gemini {
namespace {
#define MM_TRY(expr)                                  \
  do {                                                \
    const int rc_ = (expr);                           \
    if (rc_ != MMH_OK) return static_cast<Code>(rc_); \
  } while (0)
// several allocations checked at once: the first failure's code (codes are an enumeration, not flags)
inline int first_rc(std::initializer_list<int> rcs) {
  for (int rc : rcs)
    if (rc != MMH_OK) return rc;
  return MMH_OK;
}
#define MM_ENSURE(cond, code) \
  do {                        \
    if (!(cond)) return code; \
  } while (0)

// device buffer of 64-bit words for the span of one call
struct Dev {
  uint64_t *p = nullptr;
  int rc = MMH_OK;
  explicit Dev(size_t words) {
    void *q = nullptr;
    rc = mmh_malloc(&q, (words ? words : 1) * 8);
    p = static_cast<uint64_t *>(q);
  }
  ~Dev() { mmh_free(p); }
  Dev(const Dev &) = delete;
  int up(const uint64_t *src, size_t words, size_t at = 0) const {
    return words ? mmh_memcpy_h2d(p + at, src, words * 8, nullptr) : MMH_OK;
  }
  int down(uint64_t *dst, size_t words, size_t at = 0) const {
    const int r = words ? mmh_memcpy_d2h(dst, p + at, words * 8, nullptr) : MMH_OK;
    return r ? r : mmh_stream_sync(nullptr);
  }
};
bool seed_from_os(uint32_t key[8]) {
  FILE *f = std::fopen("/dev/urandom", "rb");
  const bool ok = f && std::fread(key, 1, 32, f) == 32;
  if (f) std::fclose(f);
  return ok;
}
// device copies of the key material a class needs: sk [L][N], its last-level restriction [N], pk at the last level
// [2][N] (limb 0 of both polynomials)
struct DeviceKeys {
  uint64_t *sk = nullptr, *sk_last = nullptr, *pk = nullptr, *pk_last = nullptr;
  void clear() {
    mmh_free(sk), mmh_free(sk_last), mmh_free(pk), mmh_free(pk_last);
    sk = sk_last = pk = pk_last = nullptr;
  }
  static int put(uint64_t **dst, const uint64_t *src, size_t words) {
    void *q = nullptr;
    int rc = mmh_malloc(&q, words * 8);
    if (!rc) rc = mmh_memcpy_h2d(q, src, words * 8, nullptr);
    if (!rc) rc = mmh_stream_sync(nullptr);
    *dst = static_cast<uint64_t *>(q);
    return rc;
  }
  int load(const mm::he::Context &c, const mm::he::SecretKey *s, const mm::he::PublicKey *p) {
    const size_t N = c.poly_degree(), L = c.coeff_modulus_size();
    clear();
    int rc = MMH_OK;
    if (s) {
      rc = put(&sk, s->data.data(), L * N);
      if (!rc) rc = put(&sk_last, s->data.data(), N);
    }
    if (p && !rc) {
      std::vector<uint64_t> last(2 * N);
      std::copy(p->data.begin(), p->data.begin() + N, last.begin());
      std::copy(p->data.begin() + L * N, p->data.begin() + L * N + N, last.begin() + N);
      rc = put(&pk, p->data.data(), 2 * L * N);
      if (!rc) rc = put(&pk_last, last.data(), 2 * N);
    }
    return rc;
  }
};
}  // namespace
}  // namespace gemini
