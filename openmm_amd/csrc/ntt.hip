// Batched negacyclic NTT / inverse NTT for N = 4096 and 8192 on gfx950.
//
// One 256-thread workgroup transforms one polynomial limb entirely on chip:
//   pass A  radix-2^(logN-8) butterflies on registers straight from coalesced HBM loads,
//   pass B  radix-16 on 16-element groups of stride 16 exchanged through LDS,
//   pass C  radix-16 on 16 consecutive elements, then a coalesced 16-B/lane store.
// 64-bit Harvey butterflies with Shoup twiddles (w, floor(w 2^64/q)) on VALU; values stay
// lazily reduced in [0,8q) and are corrected once at the end,
// so the visible result equals seal::util::ntt_negacyclic_harvey /
// inverse_ntt_negacyclic_harvey (hom_conv2d_ss.cc:306,315,333): bit-reversed order,
// fully reduced.
//
// LDS image: element j lives at j + (j >> 4) (one 8-byte pad per 16 elements), which makes
// every access pattern of the three passes bank-conflict free for ds_read_b64/ds_write_b64.
#include "common.hpp"

namespace mmh {

__device__ __forceinline__ int pad(int j) { return j + (j >> 4); }

// R radix-2 stages (forward, Cooley-Tukey) on 2^R register-resident elements.
// Element k of the group sits at index jhi*(G<<R) + k*G + jlo; stage u pairs k with
// k + 2^(R-1-u) and uses twiddle fwd[2^(S0+u) + (jhi<<u) + (k >> (R-u))].
template <int R, int S0>
__device__ __forceinline__ void fwd_radix(u64 (&x)[1 << R], const ulonglong2 *__restrict__ tw,
                                          u32 jhi, u64 nq, u64 q4) {
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int half = 1 << (R - 1 - u);
    const u32 base = (1u << (S0 + u)) + (jhi << u);
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) {
      const ulonglong2 w = tw[base + b];
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int k0 = b * 2 * half + i, k1 = k0 + half;
        u64 X = x[k0];
        X -= (X >= q4) ? q4 : 0;
        const u64 T = mul_shoup_lazy4(x[k1], w.x, w.y, nq);
        x[k0] = X + T;
        x[k1] = X + q4 - T;
      }
    }
  }
}

// The same stages run backwards (Gentleman-Sande) with the inverse twiddles.
template <int R, int S0>
__device__ __forceinline__ void inv_radix(u64 (&x)[1 << R], const ulonglong2 *__restrict__ tw,
                                          u32 jhi, u64 nq, u64 q4) {
#pragma unroll
  for (int u = R - 1; u >= 0; --u) {
    const int half = 1 << (R - 1 - u);
    const u32 base = (1u << (S0 + u)) + (jhi << u);
#pragma unroll
    for (int b = 0; b < (1 << u); ++b) {
      const ulonglong2 w = tw[base + b];
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int k0 = b * 2 * half + i, k1 = k0 + half;
        const u64 X = x[k0], Y = x[k1];
        u64 s = X + Y;
        s -= (s >= q4) ? q4 : 0;
        x[k0] = s;
        x[k1] = mul_shoup_lazy4(X + q4 - Y, w.x, w.y, nq);
      }
    }
  }
}

template <int LOGN>
__global__ __launch_bounds__(256) void ntt_fwd_kernel(u64 *__restrict__ data,
                                                      const DevMod *__restrict__ mods,
                                                      const ulonglong2 *__restrict__ tables,
                                                      int limbs, int mod_base) {
  constexpr int N = 1 << LOGN, RA = LOGN - 8, EA = 1 << RA, REPS = N / 4096;
  extern __shared__ __align__(16) u64 lds[];
  const int t = threadIdx.x;
  const DevMod &M = mods[mod_base + blockIdx.x % limbs];
  const u64 q = M.q, q2 = 2 * q, q4 = 4 * q, nq = M.nq;
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset;
  u64 *__restrict__ a = data + (size_t)blockIdx.x * N;

  {
    u64 x[EA];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = a[k * 256 + t];
    fwd_radix<RA, 0>(x, tw, 0, nq, q4);
#pragma unroll
    for (int k = 0; k < EA; ++k) lds[pad(k * 256 + t)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep, jhi = g >> 4, jlo = g & 15;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    fwd_radix<4, RA>(x, tw, jhi, nq, q4);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[17 * g + k];
    fwd_radix<4, RA + 4>(x, tw, g, nq, q4);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      u64 v = x[k];
      v -= (v >= q4) ? q4 : 0;
      v -= (v >= q2) ? q2 : 0;
      v -= (v >= q) ? q : 0;
      lds[17 * g + k] = v;
    }
  }
  __syncthreads();
  ulonglong2 *__restrict__ out = reinterpret_cast<ulonglong2 *>(a);
#pragma unroll
  for (int k = 0; k < N / 512; ++k) {
    const int j = 2 * (k * 256 + t);
    out[k * 256 + t] = make_ulonglong2(lds[pad(j)], lds[pad(j) + 1]);
  }
}

template <int LOGN>
__global__ __launch_bounds__(256) void ntt_inv_kernel(u64 *__restrict__ data,
                                                      const DevMod *__restrict__ mods,
                                                      const ulonglong2 *__restrict__ tables,
                                                      int limbs, int mod_base) {
  constexpr int N = 1 << LOGN, RA = LOGN - 8, EA = 1 << RA, REPS = N / 4096;
  extern __shared__ __align__(16) u64 lds[];
  const int t = threadIdx.x;
  const DevMod &M = mods[mod_base + blockIdx.x % limbs];
  const u64 q = M.q, q2 = 2 * q, q4 = 4 * q, nq = M.nq;
  const ulonglong2 *__restrict__ tw = tables + M.tw_offset + N;
  u64 *__restrict__ a = data + (size_t)blockIdx.x * N;

  const ulonglong2 *__restrict__ in = reinterpret_cast<const ulonglong2 *>(a);
#pragma unroll
  for (int k = 0; k < N / 512; ++k) {
    const int j = 2 * (k * 256 + t);
    const ulonglong2 v = in[k * 256 + t];
    lds[pad(j)] = v.x;
    lds[pad(j) + 1] = v.y;
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[17 * g + k];
    inv_radix<4, RA + 4>(x, tw, g, nq, q4);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[17 * g + k] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int rep = 0; rep < REPS; ++rep) {
    const int g = t + 256 * rep, jhi = g >> 4, jlo = g & 15;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[pad(jhi * 256 + k * 16 + jlo)];
    inv_radix<4, RA>(x, tw, jhi, nq, q4);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad(jhi * 256 + k * 16 + jlo)] = x[k];
  }
  __syncthreads();
  {
    u64 x[EA];
#pragma unroll
    for (int k = 0; k < EA; ++k) x[k] = lds[pad(k * 256 + t)];
    inv_radix<RA, 0>(x, tw, 0, nq, q4);
    const u64 ninv = M.ninv, ninv_sh = M.ninv_sh;
#pragma unroll
    for (int k = 0; k < EA; ++k) {
      u64 v = mul_shoup_lazy4(x[k], ninv, ninv_sh, nq);
      v -= (v >= q2) ? q2 : 0;
      v -= (v >= q) ? q : 0;
      a[k * 256 + t] = v;
    }
  }
}

template <int LOGN>
static constexpr size_t lds_bytes() {
  return ((size_t(1) << LOGN) + (size_t(1) << (LOGN - 4))) * sizeof(u64);
}

template <int LOGN>
static int launch(bool inverse, u64 *d, const DevMod *mods, const ulonglong2 *tables, size_t blocks,
                  int limbs, int mod_base, hipStream_t s) {
  auto kern = inverse ? ntt_inv_kernel<LOGN> : ntt_fwd_kernel<LOGN>;
  if (lds_bytes<LOGN>() > 48 * 1024)  // per device and cheap: set on every launch
    MMH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<LOGN>()));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes<LOGN>(), s, d, mods, tables,
                     limbs, mod_base);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

int ntt_dispatch(const mmh_ctx *ctx, bool inverse, u64 *d_polys, size_t n_polys, int limbs,
                 int mod_base, hipStream_t s) {
  const size_t blocks = n_polys * (size_t)limbs;
  if (blocks == 0) return MMH_OK;
  MMH_REQUIRE(blocks <= 0x7fffffffu, MMH_ERR_OUT_BOUND);
  switch (ctx->logN) {
    case 12: return launch<12>(inverse, d_polys, ctx->d_mods, ctx->d_tables, blocks, limbs, mod_base, s);
    case 13: return launch<13>(inverse, d_polys, ctx->d_mods, ctx->d_tables, blocks, limbs, mod_base, s);
    default: MMH_FAIL(MMH_ERR_CONFIG, "only N = 4096 and N = 8192 are built");
  }
}
}  // namespace mmh

using namespace mmh;

extern "C" int mmh_ntt_forward(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limbs,
                               void *stream) {
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(limbs >= 1 && limbs <= ctx->L, MMH_ERR_DIM_MISMATCH);
  return ntt_dispatch(ctx, false, d_polys, n_polys, (int)limbs, 0, (hipStream_t)stream);
}
extern "C" int mmh_ntt_inverse(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, size_t limbs,
                               void *stream) {
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(limbs >= 1 && limbs <= ctx->L, MMH_ERR_DIM_MISMATCH);
  return ntt_dispatch(ctx, true, d_polys, n_polys, (int)limbs, 0, (hipStream_t)stream);
}
extern "C" int mmh_ntt_forward_t(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, void *stream) {
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  return ntt_dispatch(ctx, false, d_polys, n_polys, 1, (int)ctx->L, (hipStream_t)stream);
}
extern "C" int mmh_ntt_inverse_t(const mmh_ctx *ctx, uint64_t *d_polys, size_t n_polys, void *stream) {
  MMH_REQUIRE(ctx && (d_polys || n_polys == 0), MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->batching, MMH_ERR_CONFIG);
  return ntt_dispatch(ctx, true, d_polys, n_polys, 1, (int)ctx->L, (hipStream_t)stream);
}
