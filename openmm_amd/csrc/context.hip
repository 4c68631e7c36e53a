// Context creation (host precomputation + upload), memory helpers, error text.
#include "common.hpp"
#include "host_math.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace mmh {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

// ---------------------------------------------------------------- scratch cache
// Layer calls need scratch (transformed operands, accumulators) for the length of the call.  hipMallocAsync /
// hipFreeAsync cost 15-22 us each on the host here (profiles/r01_hip_api_small_layer.csv), more than the
// kernels of a small layer, so freed blocks are kept in a process-wide list tagged with device and stream:
// a block is only handed out again on the stream it was last used on, where stream order already makes
// the reuse safe; no event and no synchronisation.  mmh_scratch_release() returns everything to the driver.
struct ScratchBlock {
  void *p;
  size_t bytes;
  int device;
  hipStream_t stream;
};
static std::mutex g_scratch_mu;
static std::vector<ScratchBlock> g_scratch_free;
static std::unordered_map<void *, ScratchBlock> g_scratch_live;

static void scratch_drop_free(int device) {  // caller holds the lock and has synchronised the device
  size_t w = 0;
  for (size_t i = 0; i < g_scratch_free.size(); ++i) {
    if (device < 0 || g_scratch_free[i].device == device) {
      (void)hipFree(g_scratch_free[i].p);
    } else {
      g_scratch_free[w++] = g_scratch_free[i];
    }
  }
  g_scratch_free.resize(w);
}

hipError_t scratch_alloc(void **p, size_t bytes, hipStream_t s) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  size_t best = g_scratch_free.size();
  for (size_t i = 0; i < g_scratch_free.size(); ++i) {
    const ScratchBlock &b = g_scratch_free[i];
    if (b.device != device || b.stream != s || b.bytes < bytes || b.bytes > 2 * bytes + (size_t(1) << 20)) continue;
    if (best == g_scratch_free.size() || b.bytes < g_scratch_free[best].bytes) best = i;
  }
  ScratchBlock blk;
  if (best < g_scratch_free.size()) {
    blk = g_scratch_free[best];
    g_scratch_free[best] = g_scratch_free.back();
    g_scratch_free.pop_back();
  } else {
    void *q = nullptr;
    e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {  // give the cached blocks back and try once more
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      scratch_drop_free(device);
      e = hipMalloc(&q, bytes);
      if (e != hipSuccess) return e;
    }
    blk = ScratchBlock{q, bytes, device, s};
  }
  blk.stream = s;
  g_scratch_live[blk.p] = blk;
  *p = blk.p;
  return hipSuccess;
}

hipError_t scratch_free(void *p, hipStream_t s) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto it = g_scratch_live.find(p);
  if (it == g_scratch_live.end()) return hipErrorInvalidValue;
  ScratchBlock blk = it->second;
  g_scratch_live.erase(it);
  blk.stream = s;  // kernels of this call may still be running: reusable in stream order only
  g_scratch_free.push_back(blk);
  return hipSuccess;
}

static int upload(mmh_ctx *ctx, const void *src, size_t bytes, void **dst) {
  MMH_HIP(hipMalloc(dst, bytes));
  ctx->allocs.push_back(*dst);
  MMH_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return MMH_OK;
}

// twiddle tables for modulus m with 2N-th root psi, appended to `tables`
static void build_mod(mmh_ctx *ctx, u64 m, u64 psi, DevMod *out, std::vector<ulonglong2> &tables) {
  const size_t N = ctx->N, base = tables.size();
  tables.resize(base + 2 * N);
  ulonglong2 *fwd = tables.data() + base, *inv = fwd + N;
  const u64 ipsi = inv_mod(psi, m);
  // MMH_NTT_F64=0 keeps the integer butterflies for every modulus (A/B measurements, tests)
  const char *env = getenv("MMH_NTT_F64");
  const bool f64 = m < (u64(1) << 49) && !(env && env[0] == '0');
  // integer moduli: (w, floor(w 2^64 / q)) pairs.  Moduli on the double-precision path need the twiddle as a double
  // only: a compact table of N doubles at the start of each half (the second N words of the half stay unused), so
  // that a lane's run of consecutive twiddles is contiguous and the transform pulls half the lines through L2.
  u64 p = 1, ip = 1;
  for (size_t k = 0; k < N; ++k) {
    const uint32_t r = bit_reverse((uint32_t)k, ctx->logN);
    if (f64) {
      const double d = (double)p, id = (double)ip;  // exact: below 2^49
      memcpy(reinterpret_cast<u64 *>(fwd) + r, &d, 8);
      memcpy(reinterpret_cast<u64 *>(inv) + r, &id, 8);
    } else {
      fwd[r] = make_ulonglong2(p, shoup(p, m));
      inv[r] = make_ulonglong2(ip, shoup(ip, m));
    }
    p = mul_mod(p, psi, m);
    ip = mul_mod(ip, ipsi, m);
  }
  out->q = m;
  out->nq = 0 - m;
  const u128 ratio = m & (m - 1) ? ~(u128)0 / m : (~(u128)0 / m) + 1;  // floor(2^128/m)
  out->barrett_hi = (u64)(ratio >> 64);
  out->barrett_lo = (u64)ratio;
  out->ninv = inv_mod((u64)N % m, m);
  out->ninv_sh = shoup(out->ninv, m);
  out->tw_offset = base;
  // pseudo-Mersenne shape used by the NTT's partial reductions
  const int b = bit_count(m);
  const u64 c = b >= 64 ? 0 : (u64(1) << b) - m;
  out->f64 = f64;
  out->qd = (double)m;
  out->qinv = 1.0 / (double)m;
  out->fold_c = out->fold_shift = out->fold_mask = 0;
  if (b >= 32 && b <= 60 && c < (u64(1) << 31)) {
    // a folded value is at most 2^b - 1 + (2^(64-b) - 1)*c and must stay below 2q
    const u128 worst = ((u128)1 << b) - 1 + (u128)((u64(1) << (64 - b)) - 1) * c;
    if (worst < 2 * (u128)m) {
      out->fold_c = (u32)c;
      out->fold_shift = (u32)(b - 32);
      out->fold_mask = (u32)((u64(1) << (b - 32)) - 1);
    }
  }
}
static ModC make_modc(u64 m) {
  const u128 ratio = m & (m - 1) ? ~(u128)0 / m : (~(u128)0 / m) + 1;
  return ModC{m, (u64)(ratio >> 64), (u64)ratio};
}

// SEAL ContextData / RNSTool constants for both levels; Q < 2^128 because L <= 2.
static int build_bfv(mmh_ctx *ctx) {
  BfvConsts &b = ctx->bfv;
  const u64 t = ctx->t;
  const size_t N = ctx->N;
  b.t = make_modc(t);
  b.gamma = make_modc(ctx->gamma);
  b.thr = (t + 1) >> 1;
  b.t_log2 = ctx->t_pow2 ? bit_count(t) - 1 : -1;
  b.t_inv64 = 0;
  if (t & 1) {  // Newton iteration for t^-1 mod 2^64
    u64 x = t;
    for (int i = 0; i < 6; ++i) x *= 2 - t * x;
    b.t_inv64 = x;
  }
  MMH_REQUIRE(ctx->t_pow2 || (t & 1), MMH_ERR_CONFIG);  // t = 2^k or odd, as in the reference
  b.inv_gamma_mod_t = inv_mod(ctx->gamma % t, t);
  MMH_REQUIRE(b.inv_gamma_mod_t != 0 || t == 1, MMH_ERR_CONFIG);
  b.inv_qlast_mod_q0 = ctx->L == 2 ? inv_mod(ctx->q[1] % ctx->q[0], ctx->q[0]) : 0;
  // truncate_for_decryption widths (hom_conv2d_ss.cc:418-436); (int)log2(12*N*sqrt(1/18)) = logN+1
  const int n_delta_bits = bit_count(ctx->q[0]) - bit_count(t);
  auto bits_mask = [](int z) {
    z = z < 0 ? 0 : (z > 63 ? 63 : z);
    return (~u64(0) >> z) << z;
  };
  b.mask0 = bits_mask(n_delta_bits - 1 - (ctx->t_pow2 ? 0 : 1));
  b.mask1 = bits_mask(n_delta_bits - (ctx->logN + 1));
  for (int lv = 0; lv < 2; ++lv) {
    LevelConsts &c = b.lv[lv];
    c.limbs = lv == 0 ? (int)ctx->L : 1;
    u128 Q = 1;
    for (int l = 0; l < c.limbs; ++l) Q *= ctx->q[l];
    c.q_mod_t = (u64)(Q % t);
    const u128 delta = Q / t;
    for (int l = 0; l < c.limbs; ++l) {
      const u64 ql = ctx->q[l];
      const u128 punct = Q / ql;
      c.delta_mod_q[l] = (u64)(delta % ql);
      c.prod_t_gamma_mod_q[l] = mul_mod(t % ql, ctx->gamma % ql, ql);
      c.inv_punct_mod_q[l] = inv_mod((u64)(punct % ql), ql);
      c.punct_mod_t[l] = (u64)(punct % t);
      c.punct_mod_gamma[l] = (u64)(punct % ctx->gamma);
    }
    const u64 it = inv_mod((u64)(Q % t), t), ig = inv_mod((u64)(Q % ctx->gamma), ctx->gamma);
    MMH_REQUIRE(it != 0 || t == 1, MMH_ERR_CONFIG);  // gcd(Q, t) = 1
    c.neg_inv_q_mod_t = (t - it) % t;
    c.neg_inv_q_mod_gamma = ctx->gamma - ig;
  }
  if (ctx->batching) {
    // BatchEncoder::populate_matrix_reps_index_map: powers of the generator 3 mod 2N
    std::vector<uint32_t> map(N);
    const size_t m = 2 * N, row = N / 2;
    u64 pos = 1;
    for (size_t i = 0; i < row; ++i) {
      map[i] = bit_reverse((uint32_t)((pos - 1) >> 1), ctx->logN);
      map[row + i] = bit_reverse((uint32_t)((m - pos - 1) >> 1), ctx->logN);
      pos = (pos * 3) & (m - 1);
    }
    void *d = nullptr;
    int rc = upload(ctx, map.data(), N * sizeof(uint32_t), &d);
    if (rc) return rc;
    ctx->d_slot_map = (uint32_t *)d;
  }
  return MMH_OK;
}
}  // namespace mmh

using namespace mmh;

extern "C" const char *mmh_last_error(void) { return g_last_error.c_str(); }

extern "C" int mmh_device_count(int *count) {
  MMH_REQUIRE(count, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipGetDeviceCount(count));
  return MMH_OK;
}

extern "C" int mmh_ctx_create(mmh_ctx **out, size_t N, const uint64_t *q, size_t L, uint64_t t,
                              int device) {
  MMH_REQUIRE(out && q, MMH_ERR_NULL_POINTER);
  *out = nullptr;
  const int logN = log2_exact(N);
  MMH_REQUIRE(logN == 12 || logN == 13, MMH_ERR_CONFIG);
  MMH_REQUIRE(L >= 1 && L <= (size_t)kMaxLimbs, MMH_ERR_CONFIG);
  MMH_REQUIRE(t == 0 || L <= 2, MMH_ERR_CONFIG);
  for (size_t l = 0; l < L; ++l) {
    // SEAL_USER_MOD_BIT_COUNT_MAX = 60; the NTT's lazy ranges use the headroom up to 16q < 2^64
    MMH_REQUIRE(bit_count(q[l]) <= 60 && is_prime(q[l]) && q[l] % (2 * N) == 1, MMH_ERR_INVALID_ARG);
    MMH_REQUIRE(t == 0 || q[l] > t, MMH_ERR_INVALID_ARG);  // fast plain lift needs q_l > t
    for (size_t k = 0; k < l; ++k) MMH_REQUIRE(q[k] != q[l], MMH_ERR_INVALID_ARG);
  }
  DeviceScope scope(device);  // tables are uploaded to `device`; the caller's current device is put back on return
  if (scope.rc) MMH_FAIL(scope.rc, "cannot make device " + std::to_string(device) + " current");
  mmh_ctx *ctx = new mmh_ctx();
  ctx->device = device;
  if (hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->n_cu <= 0)
    ctx->n_cu = 256;
  ctx->N = N;
  ctx->logN = logN;
  ctx->L = L;
  ctx->t = t;
  ctx->t_pow2 = t && !(t & (t - 1));
  int rc = MMH_OK;
  std::vector<ulonglong2> tables;
  for (size_t l = 0; l < L; ++l) {
    ctx->q[l] = q[l];
    ctx->psi[l] = min_primitive_root(2 * N, q[l]);
    build_mod(ctx, q[l], ctx->psi[l], &ctx->h_mods[l], tables);
  }
  size_t n_mods = L;
  if (t > 1 && is_prime(t) && t % (2 * N) == 1) {
    ctx->batching = true;
    ctx->t_psi = min_primitive_root(2 * N, t);
    build_mod(ctx, t, ctx->t_psi, &ctx->h_mods[L], tables);
    n_mods = L + 1;
  }
  {
    void *d = nullptr;
    rc = upload(ctx, tables.data(), tables.size() * sizeof(ulonglong2), &d);
    ctx->d_tables = (ulonglong2 *)d;
    for (size_t i = 0; i < n_mods && !rc; ++i) {
      ctx->h_mods[i].fwd = ctx->d_tables + ctx->h_mods[i].tw_offset;
      ctx->h_mods[i].inv = ctx->h_mods[i].fwd + N;
    }
  }
  if (!rc && t > 1) {
    auto g = primes_below(2 * N, 61, 2);  // RNSTool::initialize: {m_sk, gamma, ...}
    ctx->gamma = g.size() == 2 ? g[1] : 0;
    rc = build_bfv(ctx);
  }
  if (!rc) {
    void *d = nullptr;
    rc = upload(ctx, ctx->h_mods, n_mods * sizeof(DevMod), &d);
    ctx->d_mods = (DevMod *)d;
  }
  if (!rc && t > 1) {
    void *d = nullptr;
    rc = upload(ctx, &ctx->bfv, sizeof(BfvConsts), &d);
    ctx->d_bfv = (BfvConsts *)d;
  }
  if (rc) {
    mmh_ctx_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return MMH_OK;
}

extern "C" int mmh_ctx_create_bits(mmh_ctx **out, size_t N, const int *bits, size_t L, uint64_t t,
                                   int device) {
  MMH_REQUIRE(out && bits, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(L >= 1 && L <= (size_t)kMaxLimbs, MMH_ERR_CONFIG);
  auto q = coeff_modulus(N, bits, L);
  MMH_REQUIRE(q.size() == L, MMH_ERR_INVALID_ARG);
  return mmh_ctx_create(out, N, q.data(), L, t, device);
}

extern "C" int mmh_ctx_destroy(mmh_ctx *ctx) {
  if (!ctx) return MMH_OK;
  DeviceScope scope(ctx->device);
  for (void *p : ctx->allocs) (void)hipFree(p);
  delete ctx;
  return MMH_OK;
}

extern "C" int mmh_ctx_info(const mmh_ctx *ctx, size_t *N, size_t *L, uint64_t *q, uint64_t *psi,
                            uint64_t *t, uint64_t *gamma) {
  MMH_REQUIRE(ctx, MMH_ERR_NULL_POINTER);
  if (N) *N = ctx->N;
  if (L) *L = ctx->L;
  for (size_t l = 0; l < ctx->L; ++l) {
    if (q) q[l] = ctx->q[l];
    if (psi) psi[l] = ctx->psi[l];
  }
  if (t) *t = ctx->t;
  if (gamma) *gamma = ctx->gamma;
  return MMH_OK;
}

extern "C" int mmh_scratch_release(void) {
  MMH_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  int device = 0;
  MMH_HIP(hipGetDevice(&device));
  scratch_drop_free(device);
  return MMH_OK;
}

extern "C" int mmh_malloc(void **d_ptr, size_t bytes) {
  MMH_REQUIRE(d_ptr, MMH_ERR_NULL_POINTER);
  if (hipMalloc(d_ptr, bytes) != hipSuccess) MMH_FAIL(MMH_ERR_MEMORY, "hipMalloc failed");
  return MMH_OK;
}
extern "C" int mmh_free(void *d_ptr) {
  MMH_HIP(hipFree(d_ptr));
  return MMH_OK;
}
extern "C" int mmh_host_alloc(void **h_ptr, size_t bytes) {
  MMH_REQUIRE(h_ptr, MMH_ERR_NULL_POINTER);
  if (hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
    MMH_FAIL(MMH_ERR_MEMORY, "hipHostMalloc failed");
  return MMH_OK;
}
extern "C" int mmh_host_free(void *h_ptr) {
  MMH_HIP(hipHostFree(h_ptr));
  return MMH_OK;
}
extern "C" int mmh_get_device(int *device) {
  MMH_REQUIRE(device, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipGetDevice(device));
  return MMH_OK;
}
extern "C" int mmh_set_device(int device) {
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    MMH_FAIL(MMH_ERR_CONFIG, "hipSetDevice(" + std::to_string(device) + ") failed");
  }
  return MMH_OK;
}
extern "C" int mmh_stream_create(void **stream) {
  MMH_REQUIRE(stream, MMH_ERR_NULL_POINTER);
  hipStream_t s = nullptr;
  MMH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void *)s;
  return MMH_OK;
}
extern "C" int mmh_stream_destroy(void *stream) {
  if (stream) MMH_HIP(hipStreamDestroy((hipStream_t)stream));
  return MMH_OK;
}
// events for callers without a HIP toolchain: a point in a stream the host can wait for without draining the stream
extern "C" int mmh_event_create(void **event) {
  MMH_REQUIRE(event, MMH_ERR_NULL_POINTER);
  hipEvent_t e = nullptr;
  MMH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = (void *)e;
  return MMH_OK;
}
extern "C" int mmh_event_destroy(void *event) {
  if (event) MMH_HIP(hipEventDestroy((hipEvent_t)event));
  return MMH_OK;
}
extern "C" int mmh_event_record(void *event, void *stream) {
  MMH_REQUIRE(event, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return MMH_OK;
}
extern "C" int mmh_event_sync(void *event) {
  MMH_REQUIRE(event, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipEventSynchronize((hipEvent_t)event));
  return MMH_OK;
}
extern "C" int mmh_host_register(void *h_ptr, size_t bytes) {
  MMH_REQUIRE(h_ptr && bytes, MMH_ERR_NULL_POINTER);
  const hipError_t e = hipHostRegister(h_ptr, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    MMH_FAIL(MMH_ERR_MEMORY, std::string("hipHostRegister -> ") + hipGetErrorString(e));
  }
  return MMH_OK;
}
extern "C" int mmh_host_unregister(void *h_ptr) {
  MMH_REQUIRE(h_ptr, MMH_ERR_NULL_POINTER);
  const hipError_t e = hipHostUnregister(h_ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    MMH_FAIL(MMH_ERR_MEMORY, std::string("hipHostUnregister -> ") + hipGetErrorString(e));
  }
  return MMH_OK;
}
extern "C" int mmh_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *stream) {
  MMH_REQUIRE((d_dst && src) || bytes == 0, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return MMH_OK;
}
extern "C" int mmh_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *stream) {
  MMH_REQUIRE((dst && d_src) || bytes == 0, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return MMH_OK;
}
extern "C" int mmh_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream) {
  MMH_REQUIRE((d_dst && d_src) || bytes == 0, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMH_OK;
}
extern "C" int mmh_memset(void *d_dst, int value, size_t bytes, void *stream) {
  MMH_REQUIRE(d_dst || bytes == 0, MMH_ERR_NULL_POINTER);
  MMH_HIP(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
  return MMH_OK;
}
extern "C" int mmh_stream_sync(void *stream) {
  MMH_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MMH_OK;
}
