/* TEST INFRASTRUCTURE ONLY (see mm_oracle.h): an AVX-512 restatement of the oracle's lazy Harvey transforms, eight
 * 64-bit lanes per instruction, for the CPU baseline that bench.py times beside the GPU kernel.  The reference's own
 * NTT is Intel HEXL's AVX-512 code behind SEAL (SEAL_USE_INTEL_HEXL, SURVEY.md 8(c): absent from /root/reference);
 * this file restates the same butterflies (mm_oracle.c mmo_ntt_forward_fast / mmo_ntt_inverse_fast, which follow
 * seal/util/dwthandler.h) with vector instructions and is checked against them bit for bit in
 * tests/test_oracle.py.  64-bit high products are built from four 32x32 products (no IFMA: q0 has 60 bits,
 * past the 52-bit multiplier, which is also why HEXL takes its 64-bit path for this prime).
 * Built with per-function target attributes; callers ask mmo_have_avx512() first. */
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>

#include "mm_oracle.h"

#define V512 __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
#define F512 __attribute__((target("avx512f,avx512dq")))

int mmo_have_avx512(void) { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }

V512 __m512i mulhi64(__m512i a, __m512i b) {
  const __m512i mask = _mm512_set1_epi64(0xffffffffll);
  const __m512i ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
  const __m512i ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b),
                hh = _mm512_mul_epu32(ah, bh);
  const __m512i mid = _mm512_add_epi64(_mm512_add_epi64(_mm512_srli_epi64(ll, 32), _mm512_and_si512(lh, mask)),
                                       _mm512_and_si512(hl, mask));
  return _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(mid, 32)),
                          _mm512_add_epi64(_mm512_srli_epi64(lh, 32), _mm512_srli_epi64(hl, 32)));
}
/* x * w - floor(x * wsh / 2^64) * q, in [0, 2q) */
V512 __m512i shoup_lazy(__m512i x, __m512i w, __m512i wsh, __m512i q) {
  return _mm512_sub_epi64(_mm512_mullo_epi64(x, w), _mm512_mullo_epi64(mulhi64(x, wsh), q));
}
/* v >= m ? v - m : v */
V512 __m512i csub(__m512i v, __m512i m) { return _mm512_min_epu64(v, _mm512_sub_epi64(v, m)); }

/* The last three stages (gap 4, 2, 1) work on 16 consecutive coefficients: X and Y are picked out of two vectors,
 * the twiddles of the 2, 4 or 8 groups inside them spread over the lanes, and the results put back. */
static const long long kPickX[3][8] = {{0, 1, 2, 3, 8, 9, 10, 11}, {0, 1, 4, 5, 8, 9, 12, 13}, {0, 2, 4, 6, 8, 10, 12, 14}};
static const long long kPickY[3][8] = {{4, 5, 6, 7, 12, 13, 14, 15}, {2, 3, 6, 7, 10, 11, 14, 15}, {1, 3, 5, 7, 9, 11, 13, 15}};
static const long long kPutLo[3][8] = {{0, 1, 2, 3, 8, 9, 10, 11}, {0, 1, 8, 9, 2, 3, 10, 11}, {0, 8, 1, 9, 2, 10, 3, 11}};
static const long long kPutHi[3][8] = {{4, 5, 6, 7, 12, 13, 14, 15}, {4, 5, 12, 13, 6, 7, 14, 15}, {4, 12, 5, 13, 6, 14, 7, 15}};
static const long long kSpread[3][8] = {{0, 0, 0, 0, 1, 1, 1, 1}, {0, 0, 1, 1, 2, 2, 3, 3}, {0, 1, 2, 3, 4, 5, 6, 7}};

F512 void mmo_ntt_forward_avx512(uint64_t *a, size_t N, uint64_t q, const uint64_t *w, const uint64_t *wsh) {
  const __m512i vq = _mm512_set1_epi64((long long)q), vq2 = _mm512_set1_epi64((long long)(2 * q));
  size_t m = 1, gap = N >> 1;
  for (; gap >= 8; m <<= 1, gap >>= 1)
    for (size_t i = 0; i < m; i++) {
      const __m512i W = _mm512_set1_epi64((long long)w[m + i]), Ws = _mm512_set1_epi64((long long)wsh[m + i]);
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j += 8) {
        const __m512i X = csub(_mm512_loadu_si512(x + j), vq2);
        const __m512i Y = shoup_lazy(_mm512_loadu_si512(y + j), W, Ws, vq);
        _mm512_storeu_si512(x + j, _mm512_add_epi64(X, Y));
        _mm512_storeu_si512(y + j, _mm512_sub_epi64(_mm512_add_epi64(X, vq2), Y));
      }
    }
  for (int s = 0; gap >= 1; m <<= 1, gap >>= 1, s++) {  /* gap = 4, 2, 1; 8 / gap groups per 16 coefficients */
    const __m512i px = _mm512_loadu_si512(kPickX[s]), py = _mm512_loadu_si512(kPickY[s]),
                  lo = _mm512_loadu_si512(kPutLo[s]), hi = _mm512_loadu_si512(kPutHi[s]),
                  sp = _mm512_loadu_si512(kSpread[s]);
    const size_t groups = 8 / gap;
    const __mmask8 have = (__mmask8)((1u << groups) - 1);
    for (size_t b = 0; b < N / 16; b++) {
      uint64_t *p = a + 16 * b;
      const __m512i v0 = _mm512_loadu_si512(p), v1 = _mm512_loadu_si512(p + 8);
      const __m512i W = _mm512_permutexvar_epi64(sp, _mm512_maskz_loadu_epi64(have, w + m + b * groups));
      const __m512i Ws = _mm512_permutexvar_epi64(sp, _mm512_maskz_loadu_epi64(have, wsh + m + b * groups));
      const __m512i X = csub(_mm512_permutex2var_epi64(v0, px, v1), vq2);
      const __m512i Y = shoup_lazy(_mm512_permutex2var_epi64(v0, py, v1), W, Ws, vq);
      const __m512i A = _mm512_add_epi64(X, Y), B = _mm512_sub_epi64(_mm512_add_epi64(X, vq2), Y);
      _mm512_storeu_si512(p, _mm512_permutex2var_epi64(A, lo, B));
      _mm512_storeu_si512(p + 8, _mm512_permutex2var_epi64(A, hi, B));
    }
  }
  for (size_t j = 0; j < N; j += 8)
    _mm512_storeu_si512(a + j, csub(csub(_mm512_loadu_si512(a + j), vq2), vq));
}

F512 void mmo_ntt_inverse_avx512(uint64_t *a, size_t N, uint64_t q, const uint64_t *w, const uint64_t *wsh,
                                 uint64_t ninv, uint64_t ninv_sh) {
  const __m512i vq = _mm512_set1_epi64((long long)q), vq2 = _mm512_set1_epi64((long long)(2 * q));
  size_t m = N >> 1, gap = 1;
  for (int s = 2; gap < 8; m >>= 1, gap <<= 1, s--) {  /* gap = 1, 2, 4 */
    const __m512i px = _mm512_loadu_si512(kPickX[s]), py = _mm512_loadu_si512(kPickY[s]),
                  lo = _mm512_loadu_si512(kPutLo[s]), hi = _mm512_loadu_si512(kPutHi[s]),
                  sp = _mm512_loadu_si512(kSpread[s]);
    const size_t groups = 8 / gap;
    const __mmask8 have = (__mmask8)((1u << groups) - 1);
    for (size_t b = 0; b < N / 16; b++) {
      uint64_t *p = a + 16 * b;
      const __m512i v0 = _mm512_loadu_si512(p), v1 = _mm512_loadu_si512(p + 8);
      const __m512i W = _mm512_permutexvar_epi64(sp, _mm512_maskz_loadu_epi64(have, w + m + b * groups));
      const __m512i Ws = _mm512_permutexvar_epi64(sp, _mm512_maskz_loadu_epi64(have, wsh + m + b * groups));
      const __m512i X = _mm512_permutex2var_epi64(v0, px, v1), Y = _mm512_permutex2var_epi64(v0, py, v1);
      const __m512i A = csub(_mm512_add_epi64(X, Y), vq2);
      const __m512i B = shoup_lazy(_mm512_sub_epi64(_mm512_add_epi64(X, vq2), Y), W, Ws, vq);
      _mm512_storeu_si512(p, _mm512_permutex2var_epi64(A, lo, B));
      _mm512_storeu_si512(p + 8, _mm512_permutex2var_epi64(A, hi, B));
    }
  }
  for (; m >= 1; m >>= 1, gap <<= 1)
    for (size_t i = 0; i < m; i++) {
      const __m512i W = _mm512_set1_epi64((long long)w[m + i]), Ws = _mm512_set1_epi64((long long)wsh[m + i]);
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j += 8) {
        const __m512i X = _mm512_loadu_si512(x + j), Y = _mm512_loadu_si512(y + j);
        _mm512_storeu_si512(x + j, csub(_mm512_add_epi64(X, Y), vq2));
        _mm512_storeu_si512(y + j, shoup_lazy(_mm512_sub_epi64(_mm512_add_epi64(X, vq2), Y), W, Ws, vq));
      }
    }
  const __m512i vn = _mm512_set1_epi64((long long)ninv), vns = _mm512_set1_epi64((long long)ninv_sh);
  for (size_t j = 0; j < N; j += 8)
    _mm512_storeu_si512(a + j, csub(shoup_lazy(_mm512_loadu_si512(a + j), vn, vns, vq), vq));
}
