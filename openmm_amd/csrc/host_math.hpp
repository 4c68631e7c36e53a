// Host-side number theory used once per context: prime selection, roots of unity,
// twiddle tables, BFV/BEHZ constants.  Product code (independent of oracle/).
// Selection rules follow SEAL 3.7.2 so that the parameters equal the reference's
// (cheetah-api.cpp:95-105): CoeffModulus::Create, try_minimal_primitive_root,
// RNSTool::initialize (gamma).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mmh {
using u64 = uint64_t;
using u128 = unsigned __int128;

inline u64 mul_mod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }
inline u64 pow_mod(u64 a, u64 e, u64 m) {
  u64 r = 1 % m;
  for (a %= m; e; e >>= 1, a = mul_mod(a, a, m))
    if (e & 1) r = mul_mod(r, a, m);
  return r;
}
// modular inverse by extended Euclid; 0 when not invertible
inline u64 inv_mod(u64 a, u64 m) {
  __int128 s0 = 0, s1 = 1;
  u128 r0 = m, r1 = a % m;
  while (r1) {
    u128 k = r0 / r1, r2 = r0 - k * r1;
    __int128 s2 = s0 - (__int128)k * s1;
    r0 = r1, r1 = r2, s0 = s1, s1 = s2;
  }
  if (r0 != 1) return 0;
  return (u64)(s0 < 0 ? s0 + m : s0);
}
inline bool is_prime(u64 n) {
  if (n < 2) return false;
  for (u64 p : {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37}) {
    if (n == p) return true;
    if (n % p == 0) return false;
  }
  u64 d = n - 1;
  int r = 0;
  while (!(d & 1)) d >>= 1, ++r;
  for (u64 a : {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37}) {
    u64 x = pow_mod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool witness = true;
    for (int i = 1; i < r && witness; ++i) {
      x = mul_mod(x, x, n);
      if (x == n - 1) witness = false;
    }
    if (witness) return false;
  }
  return true;
}
// `count` largest primes = 1 (mod factor) below 2^bits, descending
inline std::vector<u64> primes_below(u64 factor, int bits, size_t count) {
  std::vector<u64> out;
  u64 v = ((u64(1) << bits) - 1) / factor * factor + 1, lo = u64(1) << (bits - 1);
  for (; out.size() < count && v > lo; v -= factor)
    if (is_prime(v)) out.push_back(v);
  return out;
}
// moduli for a list of bit sizes; equal sizes are handed out smallest first
inline std::vector<u64> coeff_modulus(size_t N, const int *bits, size_t n) {
  std::vector<u64> out;
  std::vector<std::vector<u64>> pool(64);
  for (size_t i = 0; i < n; ++i) {
    if (bits[i] < 2 || bits[i] > 61) return {};
    pool[bits[i]].push_back(0);
  }
  for (int b = 2; b < 64; ++b)
    if (!pool[b].empty()) {
      size_t k = pool[b].size();
      pool[b] = primes_below(2 * N, b, k);
      if (pool[b].size() != k) return {};
    }
  for (size_t i = 0; i < n; ++i) {
    out.push_back(pool[bits[i]].back());
    pool[bits[i]].pop_back();
  }
  return out;
}
// smallest primitive (2N)-th root of unity mod prime q (0 if none)
inline u64 min_primitive_root(u64 two_n, u64 q) {
  if ((q - 1) % two_n) return 0;
  u64 root = 0;
  for (u64 g = 2; g < q && !root; ++g) {
    u64 r = pow_mod(g, (q - 1) / two_n, q);
    if (pow_mod(r, two_n / 2, q) == q - 1) root = r;
  }
  u64 sq = mul_mod(root, root, q), cur = root, best = root;
  for (u64 i = 0; i < two_n / 2; ++i, cur = mul_mod(cur, sq, q))
    if (cur < best) best = cur;
  return best;
}
inline u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
inline uint32_t bit_reverse(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
  return r;
}
inline int log2_exact(size_t n) {
  int l = 0;
  while ((size_t(1) << l) < n) ++l;
  return (size_t(1) << l) == n ? l : -1;
}
inline int bit_count(u64 v) {
  int b = 0;
  for (; v; v >>= 1) ++b;
  return b;
}
}  // namespace mmh
