// HomFCSS (include/gemini/cheetah/hom_fc_ss.cc) and the SIMD half of HomBNSS
// (hom_bn_ss.cc:151-406: encryptVector / encodeVector / bn / bole / addMask /
// decryptToVector) as layer-level calls built from the batched kernels.
#include "bfv_device.hpp"
#include "common.hpp"
#include "convert_device.hpp"
#include "geometry.hpp"

namespace mmh {
int reshare_tail_approx(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_pk_ntt_last,
                        const int64_t *d_u, const int64_t *d_e1, const u64 *d_mask, const uint8_t *d_keep,
                        size_t keep_period, u64 *d_out, hipStream_t s, const int *d_share_dst, size_t share_plane,
                        u64 *d_share_out);
int ntt_dispatch(const mmh_ctx *ctx, bool inverse, u64 *d_polys, size_t n_polys, int limbs, int mod_base, hipStream_t s);
int ntt_inv_product_each_dispatch(const mmh_ctx *ctx, const u64 *d_pt_ntt, const u64 *d_ct_ntt, const u64 *d_pt2_ntt,
                                  const u64 *d_ct2_ntt, u64 *d_out, size_t n, int limbs, hipStream_t s);
int ntt_bn_front_dispatch(const mmh_ctx *ctx, const u64 *d_cts, const u64 *d_share_pts, const u64 *d_scale_pts, size_t n,
                          u64 *d_sum, u64 *d_pt_ntt, hipStream_t s);
int bn_alice_front_launch(const u64 *d_rcm_r2f, const u64 *d_rcm_f2r, const uint64_t delta[2], const uint64_t seed_r2f[2],
                          const uint64_t seed_f2r[2], u64 gid0, const u64 *d_x, size_t n, int bw, int shift, u64 p,
                          const u64 *d_scales, const uint32_t key[8], u64 sid, u64 *d_pts3, size_t n_ct, const u32 *d_map,
                          size_t sub_len, int logN, const ModC &t, u64 *d_share, u64 *d_out_ring, uint8_t *d_wire_r2f,
                          uint8_t *d_wire_f2r, hipStream_t s);
int cot_send_r2f_encode_launch(const u64 *d_rcm, const uint64_t delta[2], const uint64_t seed[2], u64 gid0, const u64 *d_x,
                               size_t n, int bw, u64 p, u64 *d_pts, const u32 *d_map, size_t sub_len, int logN,
                               const ModC &t, uint8_t *d_wire, hipStream_t s);
int launch_share_from_mask(const mmh_ctx *ctx, const u64 *r, u64 *out, size_t groups, size_t n_targets,
                           size_t n_spatial, size_t plane, const u32 *coef, const u32 *dst, const u32 *slice,
                           hipStream_t s);
int launch_gather_targets(const mmh_ctx *ctx, const u64 *pts, u64 *out, size_t groups, size_t n_targets,
                          size_t n_spatial, size_t plane, const u32 *coef, const u32 *dst, const u32 *slice,
                          hipStream_t s);
}  // namespace mmh

struct mmh_fc_plan {
  const mmh_ctx *ctx = nullptr;
  mmh_fc_meta meta{};
  long d0 = 0, d1 = 0;         // getSplit: rows x cols of one weight block
  size_t n_in = 0, n_out = 0;  // input chunks, output row blocks
  size_t n_targets = 0;
  uint32_t *d_tgt_coef = nullptr, *d_tgt_dst = nullptr, *d_tgt_slice = nullptr;
  uint8_t *d_keep = nullptr;   // [n_out][N]
  std::vector<void *> allocs;
};

namespace mmh {
__device__ __forceinline__ u64 red_t2(u64 v, const ModC &t) { return barrett128(0, v, t.q, t.barrett_hi, t.barrett_lo); }

// encodeInputVector / encryptInputVector (hom_fc_ss.cc:296-318, :239-252): chunk i as
// v0 - sum_{j>=1} v_j X^{N-j}; coefficient N-len of a shorter last chunk keeps what the previous
// chunk left in the reference's reused buffer, reproduced here.
__global__ __launch_bounds__(256) void fc_encode_input_kernel(const u64 *__restrict__ vec, u64 *__restrict__ polys,
                                                              size_t n_in, long len_total, long d1, int logN,
                                                              ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_in << logN)) return;
  const long N = 1L << logN, i = (long)(idx >> logN), j = (long)(idx & (N - 1));
  const long start = i * d1, len = (start + d1 <= len_total ? d1 : len_total - start);
  u64 v = 0;
  if (j == 0) {
    v = vec[start];
  } else {
    const long jj = N - j;  // this coefficient holds -v[start + jj]
    long src = -1;
    if (jj < len) src = start + jj;
    else if (jj == len && i > 0 && len < d1) src = start - d1 + jj;  // stale entry of chunk i-1
    if (src >= 0) {
      const u64 u = vec[src];
      v = u > 0 ? t.q - u : 0;
    }
  }
  polys[idx] = red_t2(v, t);
}

// encodeWeightMatrix (hom_fc_ss.cc:346-385): block (rb, cb), entry (r, c) at coefficient r*d1 + c
__global__ __launch_bounds__(256) void fc_encode_weights_kernel(const u64 *__restrict__ W, u64 *__restrict__ polys,
                                                                size_t n_polys, long rows, long cols, long d0,
                                                                long d1, long n_cb, int logN, ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_polys << logN)) return;
  const long N = 1L << logN, k = (long)(idx >> logN), j = (long)(idx & (N - 1));
  const long rb = k / n_cb, cb = k % n_cb, r = j / d1, c = j % d1;
  const long row = rb * d0 + r, col = cb * d1 + c;
  u64 v = 0;
  if (r < d0 && row < rows && col < cols) v = red_t2(W[row * cols + col], t);
  polys[idx] = v;
}

// HomBNSS chunking (hom_bn_ss.cc:171-178): chunk k of a length-n vector, zero padded to N slots
__global__ __launch_bounds__(256) void chunk_pad_kernel(const u64 *__restrict__ vec, u64 *__restrict__ slots,
                                                        size_t n_ct, size_t n, size_t sub_len, int logN) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_ct << logN)) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const size_t src = k * sub_len + i;
  slots[idx] = (i < sub_len && src < n) ? vec[src] : 0;
}
__global__ __launch_bounds__(256) void chunk_unpad_kernel(const u64 *__restrict__ slots, u64 *__restrict__ vec,
                                                          size_t n, size_t sub_len, int logN) {
  const size_t src = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (src >= n) return;
  vec[src] = slots[((src / sub_len) << logN) + src % sub_len];
}
// A conversion step inside the neighbouring HomBNSS step (CheetahLinear::bn / ::BOLE convert their shares with
// ring_to_field, hand the field elements to encodeVector / encryptVector, and take decryptToVector's output to
// field_to_ring, cheetah-api.cpp:560-660): the element never makes the round trip through memory between the two.
// ring_to_field's post step -> chunking -> BatchEncoder's slot order (the inverse transform follows):
__global__ __launch_bounds__(256) void r2f_post_scatter_kernel(int party, const u64 *__restrict__ tmp,
                                                               const uint8_t *__restrict__ msb,
                                                               const u64 *__restrict__ y, u64 *__restrict__ pts,
                                                               size_t n_ct, size_t n, size_t sub_len, int logN,
                                                               const u32 *__restrict__ map, ModC t, u64 cv,
                                                               u64 big_mod_p) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_ct << logN)) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const size_t src = k * sub_len + i;
  u64 v = 0;
  if (i < sub_len && src < n) v = r2f_post_elem(party, tmp[src], msb[src], y[src], t.q, cv, big_mod_p);
  pts[(k << logN) + map[i]] = barrett128(0, v, t.q, t.barrett_hi, t.barrett_lo);   // BatchEncoder reduces mod t
}
// BatchEncoder's slot order after the forward transform -> un-chunking -> field_to_ring's pre step:
__global__ __launch_bounds__(256) void gather_f2r_pre_kernel(int party, const u64 *__restrict__ slots_ntt,
                                                             size_t n, size_t sub_len, int logN,
                                                             const u32 *__restrict__ map, u64 p, u64 corr_value,
                                                             u64 *__restrict__ tmp, uint8_t *__restrict__ msb,
                                                             u64 *__restrict__ corr) {
  const size_t src = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (src >= n) return;
  bool m;
  const u64 v = f2r_pre_elem(party, slots_ntt[((src / sub_len) << logN) + map[src % sub_len]], p, m);
  tmp[src] = v;
  msb[src] = m;
  if (party == kAlice && corr) corr[src] = m ? 0 : corr_value;
}
// HomBNSS chunking and BatchEncoder's slot order in one pass (encodeVector, hom_bn_ss.cc:203-233: chunk, zero pad,
// BatchEncoder::encode's scatter and reduction mod t; the inverse transform follows)
__global__ __launch_bounds__(256) void chunk_scatter_kernel(const u64 *__restrict__ vec, u64 *__restrict__ pts,
                                                            size_t n_ct, size_t n, size_t sub_len, int logN,
                                                            const u32 *__restrict__ map, ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_ct << logN)) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const size_t src = k * sub_len + i;
  const u64 v = (i < sub_len && src < n) ? vec[src] : 0;
  pts[(k << logN) + map[i]] = red_t2(v, t);
}
// addMask's r (hom_bn_ss.cc:333-340) straight into that slot order: r = random % p, share = -r mod p
__global__ __launch_bounds__(256) void bn_mask_scatter_kernel(const u64 *__restrict__ random, u64 *__restrict__ pts,
                                                              u64 *__restrict__ share, size_t n_ct, size_t n,
                                                              size_t sub_len, int logN, const u32 *__restrict__ map,
                                                              ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_ct << logN)) return;
  const size_t k = idx >> logN, i = idx & ((size_t(1) << logN) - 1);
  const size_t src = k * sub_len + i;
  u64 v = 0;
  if (i < sub_len && src < n) {
    v = red_t2(random[src], t);
    share[src] = v ? t.q - v : 0;
  }
  pts[(k << logN) + map[i]] = v;
}
// The rest of bn / bole per coefficient: mod_switch_to_inplace(last) of the product (hom_bn_ss.cc:269,:308), addMask's
// add_plain_inplace of encode(r) at the last level (:355) and truncate_for_decryption (:357), one pass.
// acc [n][2][2][N] coefficient form at the first level, rpts [n][N] mod t, out [n][2][N].
__global__ __launch_bounds__(256) void bn_finish_kernel(const u64 *__restrict__ acc, const u64 *__restrict__ rpts,
                                                        u64 *__restrict__ out, size_t total, int logN, BfvConsts b,
                                                        const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n*2][N]
  if (idx >= total) return;
  const size_t i = idx & ((size_t(1) << logN) - 1), p = idx >> logN;
  const DevMod &q0 = mods[0];
  const u64 q1 = mods[1].q, half = q1 >> 1;
  const u64 last = csub(acc[((p * 2 + 1) << logN) + i] + half, q1);
  const u64 tmp = sub_mod(reduce64(last, q0), reduce64(half, q0), q0.q);
  u64 v = mul_mod(sub_mod(acc[((p * 2) << logN) + i], tmp, q0.q), b.inv_qlast_mod_q0, q0);
  if ((p & 1) == 0) {
    const u64 m = rpts[((p >> 1) << logN) + i];
    if (m) v = add_mod(v, scaled_plain(m, scale_fix(m, b, b.lv[1]), b.lv[1].delta_mod_q[0], q0), q0.q);
    out[idx] = v & b.mask0;
  } else {
    out[idx] = v & b.mask1;
  }
}
// addMask (hom_bn_ss.cc:333-340): r = random % p in place, share = -r mod p
__global__ __launch_bounds__(256) void bn_mask_kernel(const u64 *__restrict__ random, u64 *__restrict__ r,
                                                      u64 *__restrict__ share, size_t n, ModC t) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 v = red_t2(random[i], t);
  r[i] = v;
  share[i] = v ? t.q - v : 0;
}

template <class P, class T>
static int upload_vec(P *p, const std::vector<T> &v, T **d) {
  MMH_HIP(hipMalloc((void **)d, std::max<size_t>(1, v.size()) * sizeof(T)));
  p->allocs.push_back(*d);
  if (!v.empty()) MMH_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return MMH_OK;
}
static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace mmh

using namespace mmh;
#define CHECK_RC(x)      \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)
#define KLAUNCH(kern, total, s, ...)                                                     \
  do {                                                                                   \
    if ((total) > 0) {                                                                   \
      hipLaunchKernelGGL(kern, dim3(nblk(total)), dim3(256), 0, (hipStream_t)(s), __VA_ARGS__); \
      MMH_HIP(hipGetLastError());                                                        \
    }                                                                                    \
  } while (0)

// ------------------------------------------------------------------------------ HomFCSS
extern "C" int mmh_fc_plan_destroy(mmh_fc_plan *p) {
  if (!p) return MMH_OK;
  for (void *a : p->allocs) (void)hipFree(a);
  delete p;
  return MMH_OK;
}

static int fc_plan_make(const mmh_ctx *ctx, const mmh_fc_meta *meta, const int64_t *split, mmh_fc_plan **out) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && meta && out, MMH_ERR_NULL_POINTER);
  *out = nullptr;
  MMH_REQUIRE(ctx->t >= 2, MMH_ERR_CONFIG);
  MMH_REQUIRE(meta->rows > 0 && meta->cols > 0, MMH_ERR_INVALID_ARG);
  const long N = (long)ctx->N;
  MMH_REQUIRE(!split || (split[0] >= 1 && split[1] >= 1 && split[0] * split[1] <= N), MMH_ERR_INVALID_ARG);
  mmh_fc_plan *p = new mmh_fc_plan();
  p->ctx = ctx;
  p->meta = *meta;
  if (split) p->d0 = (long)split[0], p->d1 = (long)split[1];
  else fc_split(meta->rows, meta->cols, N, p->d0, p->d1);
  p->n_in = (size_t)((meta->cols + p->d1 - 1) / p->d1);
  p->n_out = (size_t)((meta->rows + p->d0 - 1) / p->d0);
  // addRandomMask targets i*d1 (hom_fc_ss.cc:482-485, :504-510); removeUnusedCoeffs keeps
  // index < rows_in_block*d1 with index % d1 == 0 (:543-548); decryptToVector reads the same (:595-598)
  std::vector<u32> coef, dst, slice;
  std::vector<uint8_t> keep(p->n_out * (size_t)N, 0);
  for (size_t b = 0; b < p->n_out; ++b) {
    const long r0 = (long)b * p->d0, r1 = std::min<long>(r0 + p->d0, meta->rows);
    for (long r = r0; r < r1; ++r) {
      coef.push_back((u32)((r - r0) * p->d1));
      dst.push_back((u32)r);
      slice.push_back((u32)b);
      keep[b * N + (r - r0) * p->d1] = 1;
    }
  }
  p->n_targets = coef.size();
  int rc;
  if ((rc = upload_vec(p, coef, &p->d_tgt_coef)) || (rc = upload_vec(p, dst, &p->d_tgt_dst)) ||
      (rc = upload_vec(p, slice, &p->d_tgt_slice)) || (rc = upload_vec(p, keep, &p->d_keep))) {
    mmh_fc_plan_destroy(p);
    return rc;
  }
  *out = p;
  return MMH_OK;
}
extern "C" int mmh_fc_plan_create(const mmh_ctx *ctx, const mmh_fc_meta *meta, mmh_fc_plan **out) {
  return fc_plan_make(ctx, meta, nullptr, out);
}
extern "C" int mmh_fc_plan_create_split(const mmh_ctx *ctx, const mmh_fc_meta *meta, const int64_t split[2],
                                        mmh_fc_plan **out) {
  MMH_REQUIRE(split, MMH_ERR_NULL_POINTER);
  return fc_plan_make(ctx, meta, split, out);
}

extern "C" int mmh_fc_split(int64_t rows, int64_t cols, size_t N, int64_t split[2]) {
  MMH_REQUIRE(split, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(rows > 0 && cols > 0 && N > 0, MMH_ERR_INVALID_ARG);
  long d0 = 0, d1 = 0;
  fc_split((long)rows, (long)cols, (long)N, d0, d1);
  split[0] = d0, split[1] = d1;
  return MMH_OK;
}

extern "C" int mmh_fc_plan_info(const mmh_fc_plan *p, int64_t split[2], size_t *n_in_cts, size_t *n_out_cts) {
  MMH_REQUIRE(p, MMH_ERR_NULL_POINTER);
  if (split) split[0] = p->d0, split[1] = p->d1;
  if (n_in_cts) *n_in_cts = p->n_in;
  if (n_out_cts) *n_out_cts = p->n_out;
  return MMH_OK;
}

extern "C" int mmh_fc_encode_input(const mmh_fc_plan *p, const uint64_t *d_vec, uint64_t *d_pts, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_vec && d_pts, MMH_ERR_NULL_POINTER);
  KLAUNCH(fc_encode_input_kernel, p->n_in << p->ctx->logN, stream, d_vec, d_pts, p->n_in, (long)p->meta.cols, p->d1,
          p->ctx->logN, p->ctx->bfv.t);
  return MMH_OK;
}

extern "C" int mmh_fc_encode_weights(const mmh_fc_plan *p, const uint64_t *d_w, uint64_t *d_pts, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_w && d_pts, MMH_ERR_NULL_POINTER);
  const size_t n_polys = p->n_in * p->n_out;
  KLAUNCH(fc_encode_weights_kernel, n_polys << p->ctx->logN, stream, d_w, d_pts, n_polys, (long)p->meta.rows,
          (long)p->meta.cols, p->d0, p->d1, (long)p->n_in, p->ctx->logN, p->ctx->bfv.t);
  return MMH_OK;
}

extern "C" int mmh_fc_encrypt_input(const mmh_fc_plan *p, const uint64_t *d_sk_ntt, const uint64_t *d_vec,
                                    const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_sk_ntt && d_vec && d_a && d_e && d_cts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (p->n_in << p->ctx->logN) * 8, s));
  int rc = mmh_fc_encode_input(p, d_vec, (u64 *)pts, stream);
  if (!rc) rc = mmh_encrypt_symmetric(p->ctx, 0, d_sk_ntt, d_a, d_e, (const u64 *)pts, p->n_in, d_cts, stream);
  (void)scratch_free(pts, s);
  return rc;
}

static int fc_server_impl(const mmh_fc_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                          const uint64_t *d_weight_pts, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                          const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share,
                          void *stream, const uint64_t *d_flood, const int64_t *d_e0) {
  const bool exact = d_flood != nullptr;  // USE_APPROX_RESHARE=0 build
  MMH_REQUIRE(!exact || d_e0, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(p && d_in_cts && d_weight_pts && d_pk_ntt_last && d_u && d_e1 && d_mask && d_out_cts && d_out_share,
              MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!p->meta.is_shared_input || d_share_pts, MMH_ERR_DIM_MISMATCH);  // matVecMul :424-426
  const mmh_ctx *ctx = p->ctx;
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N;
  void *input = nullptr, *acc = nullptr, *zero = nullptr, *mask_t = nullptr;
  int rc = MMH_OK;
  const u64 *cts = d_in_cts;
  if (p->meta.is_shared_input) {  // :431-440
    MMH_HIP(scratch_alloc(&input, p->n_in * 4 * N * 8, s));
    rc = mmh_add_plain(ctx, 0, d_in_cts, d_share_pts, p->n_in, 0, (u64 *)input, stream);
    cts = (const u64 *)input;
  }
  if (!rc && scratch_alloc(&acc, p->n_out * 4 * N * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  // :443-457: out[j] = sum_i input[i] * matrix[j][i]
  if (!rc)
    rc = mmh_mul_plain_acc(ctx, 0, cts, p->n_in, d_weight_pts, p->n_out * p->n_in, p->n_out, 1, p->n_in, 1, 0,
                           p->n_in, 1, 0, (u64 *)acc, stream);
  // addRandomMask :487-517 (the output vector is sized once, not per worker as the reference does);
  // the two re-sharing builds differ as in HomConv2DSS (hom_conv2d_ss.cc:179-273, :323-328, :358-380)
  if (!rc && !exact) {
    // shipped build: the fused tail of bfv.hip (forward NTT of u, inverse NTT of u * pk, one finishing pass) gives
    // the bits of the step-by-step sequence below in 3 launches instead of 9; the share comes from the mask alone
    rc = reshare_tail_approx(ctx, (u64 *)acc, false, p->n_out, d_pk_ntt_last, d_u, d_e1, d_mask, p->d_keep, p->n_out,
                             d_out_cts, s, nullptr, 0, nullptr);
    if (!rc)
      rc = launch_share_from_mask(ctx, d_mask, d_out_share, 1, p->n_targets, p->n_out, (size_t)p->meta.rows,
                                  p->d_tgt_coef, p->d_tgt_dst, p->d_tgt_slice, s);
  }
  if (exact) {
    if (!rc) rc = mmh_flood_add(ctx, (u64 *)acc, d_flood, p->n_out, stream);
    if (!rc) rc = mmh_mod_switch_to_last(ctx, (const u64 *)acc, p->n_out, d_out_cts, stream);
    if (!rc && scratch_alloc(&zero, p->n_out * 2 * N * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
    if (!rc) rc = mmh_encrypt_zero_asym(ctx, 1, d_pk_ntt_last, d_u, d_e0, d_e1, p->n_out, (u64 *)zero, stream);
    if (!rc) rc = mmh_ct_add(ctx, 1, d_out_cts, (const u64 *)zero, p->n_out, stream);
    if (!rc && scratch_alloc(&mask_t, p->n_out * N * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
    if (!rc) rc = mmh_reduce_mod_t(ctx, d_mask, (u64 *)mask_t, p->n_out * N, stream);
    if (!rc) rc = mmh_add_plain(ctx, 1, d_out_cts, (const u64 *)mask_t, p->n_out, 1, d_out_cts, stream);
    if (!rc)
      rc = launch_gather_targets(ctx, (const u64 *)mask_t, d_out_share, 1, p->n_targets, p->n_out,
                                 (size_t)p->meta.rows, p->d_tgt_coef, p->d_tgt_dst, p->d_tgt_slice, s);
    // truncate_for_decryption :462-466 + removeUnusedCoeffs :525-561
    if (!rc) rc = mmh_truncate_and_zero(ctx, d_out_cts, p->n_out, p->d_keep, p->n_out, stream);
  }
  if (input) (void)scratch_free(input, s);
  if (acc) (void)scratch_free(acc, s);
  if (zero) (void)scratch_free(zero, s);
  if (mask_t) (void)scratch_free(mask_t, s);
  return rc;
}

extern "C" int mmh_fc_server(const mmh_fc_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                             const uint64_t *d_weight_pts, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                             const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                             uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  return fc_server_impl(p, d_in_cts, d_share_pts, d_weight_pts, d_pk_ntt_last, d_u, d_e1, d_mask, d_out_cts,
                        d_out_share, stream, nullptr, nullptr);
}

extern "C" int mmh_fc_server_exact(const mmh_fc_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                                   const uint64_t *d_weight_pts, const uint64_t *d_pk_ntt_last,
                                   const uint64_t *d_flood, const int64_t *d_u, const int64_t *d_e0,
                                   const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                                   uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(d_flood && d_e0, MMH_ERR_NULL_POINTER);
  return fc_server_impl(p, d_in_cts, d_share_pts, d_weight_pts, d_pk_ntt_last, d_u, d_e1, d_mask, d_out_cts,
                        d_out_share, stream, d_flood, d_e0);
}

extern "C" int mmh_fc_decrypt(const mmh_fc_plan *p, const uint64_t *d_sk_ntt, const uint64_t *d_cts,
                              uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_sk_ntt && d_cts && d_out, MMH_ERR_NULL_POINTER);
  const mmh_ctx *ctx = p->ctx;
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (p->n_out << ctx->logN) * 8, s));
  int rc = mmh_decrypt(ctx, 1, d_sk_ntt, d_cts, p->n_out, (u64 *)pts, stream);
  if (!rc)
    rc = launch_gather_targets(ctx, (const u64 *)pts, d_out, 1, p->n_targets, p->n_out, (size_t)p->meta.rows,
                               p->d_tgt_coef, p->d_tgt_dst, p->d_tgt_slice, s);
  (void)scratch_free(pts, s);
  return rc;
}

// ------------------------------------------------------------------- HomBNSS, SIMD path
static int bn_counts(const mmh_ctx *ctx, size_t n, size_t *sub_len, size_t *n_ct) {
  MMH_REQUIRE(ctx, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->batching, MMH_ERR_INVALID_ARG);  // setUp_Field: "invalid SIMD modulus"
  MMH_REQUIRE(n > 0, MMH_ERR_INVALID_ARG);
  *sub_len = std::min(n, ctx->N);
  *n_ct = (n + *sub_len - 1) / *sub_len;
  return MMH_OK;
}

extern "C" int mmh_bn_encode_vector(const mmh_ctx *ctx, const uint64_t *d_vec, size_t n, uint64_t *d_pts,
                                    void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_vec && d_pts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(chunk_scatter_kernel, dim3(nblk(n_ct << ctx->logN)), dim3(256), 0, s, d_vec, d_pts, n_ct, n, sub,
                     ctx->logN, ctx->d_slot_map, ctx->bfv.t);
  MMH_HIP(hipGetLastError());
  return ntt_dispatch(ctx, true, d_pts, n_ct, 1, (int)ctx->L, s);   // BatchEncoder::encode's inverse transform mod t
}

extern "C" int mmh_bn_encrypt_vector(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_vec, size_t n,
                                     const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_sk_ntt && d_vec && d_a && d_e && d_cts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (n_ct << ctx->logN) * 8, s));
  int rc = mmh_bn_encode_vector(ctx, d_vec, n, (u64 *)pts, stream);
  if (!rc) rc = mmh_encrypt_symmetric(ctx, 0, d_sk_ntt, d_a, d_e, (const u64 *)pts, n_ct, d_cts, stream);
  (void)scratch_free(pts, s);
  return rc;
}

// shared tail of bn / bole: mod-switch, addMask, truncate
static int bn_tail(const mmh_ctx *ctx, const u64 *acc, size_t n, size_t n_ct, size_t sub, const u64 *d_random,
                   u64 *d_out_cts, u64 *d_out_share, hipStream_t s) {
  // three launches: r and its share straight into slot order, BatchEncoder's inverse transform of r, and one pass that
  // mod-switches the product, adds the scaled r and truncates (the step-by-step entry points give the same bits)
  void *rpts = nullptr;
  int rc = MMH_OK;
  if (scratch_alloc(&rpts, (n_ct << ctx->logN) * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  if (!rc) {
    hipLaunchKernelGGL(bn_mask_scatter_kernel, dim3(nblk(n_ct << ctx->logN)), dim3(256), 0, s, d_random, (u64 *)rpts,
                       d_out_share, n_ct, n, sub, ctx->logN, ctx->d_slot_map, ctx->bfv.t);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  if (!rc) rc = ntt_dispatch(ctx, true, (u64 *)rpts, n_ct, 1, (int)ctx->L, s);
  if (!rc) {
    const size_t total = n_ct * 2 * ctx->N;
    hipLaunchKernelGGL(bn_finish_kernel, dim3(nblk(total)), dim3(256), 0, s, acc, (const u64 *)rpts, d_out_cts, total,
                       ctx->logN, ctx->bfv, ctx->d_mods);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  if (rpts) (void)scratch_free(rpts, s);
  return rc;
}

extern "C" int mmh_bn_server(const mmh_ctx *ctx, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                             const uint64_t *d_scale_pts, size_t n, const uint64_t *d_random, uint64_t *d_out_cts,
                             uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_in_cts && d_scale_pts && d_random && d_out_cts && d_out_share, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N;
  void *sum = nullptr, *acc = nullptr, *pt_ntt = nullptr;
  int rc = MMH_OK;
  // hom_bn_ss.cc:263-265: the server's share added (out of place), or a working copy of the client's ciphertexts
  MMH_HIP(scratch_alloc(&sum, n_ct * 4 * N * 8, s));
  if (d_share_pts)
    rc = mmh_add_plain(ctx, 0, d_in_cts, d_share_pts, n_ct, 0, (u64 *)sum, stream);
  else if (hipMemcpyAsync(sum, d_in_cts, n_ct * 4 * N * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
    rc = MMH_ERR_INTERNAL;
  if (!rc && scratch_alloc(&acc, n_ct * 4 * N * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  if (!rc && scratch_alloc(&pt_ntt, n_ct * 2 * N * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  // :265-267: out[i] = cts[i] * scales[i] (multiply_plain): both operands to NTT form, the dyadic product formed while
  // the inverse transform loads (same canonical values as mmh_mul_plain_acc's separate product pass)
  if (!rc) rc = ntt_dispatch(ctx, false, (u64 *)sum, n_ct * 2, 2, 0, s);
  if (!rc) rc = mmh_plain_lift_ntt(ctx, 0, d_scale_pts, n_ct, (u64 *)pt_ntt, stream);
  if (!rc)
    rc = ntt_inv_product_each_dispatch(ctx, (const u64 *)pt_ntt, (const u64 *)sum, nullptr, nullptr, (u64 *)acc, n_ct, 2, s);
  if (!rc) rc = bn_tail(ctx, (const u64 *)acc, n, n_ct, sub, d_random, d_out_cts, d_out_share, s);
  for (void *b : {sum, acc, pt_ntt})
    if (b) (void)scratch_free(b, s);
  return rc;
}

// ALICE's side of a batch-norm layer in two calls that meet at the client's ciphertexts (see include/mm_hip.h)
extern "C" int mmh_bn_alice_prepare(const mmh_ctx *ctx, const uint64_t *d_rcm_r2f, const uint64_t *d_rcm_f2r,
                                    const uint64_t delta[2], const uint64_t seed_r2f[2], const uint64_t seed_f2r[2],
                                    uint64_t gid0, const uint64_t *d_x, size_t n, int bw, int shift,
                                    const uint64_t *d_scales, const uint32_t key[8], uint64_t stream_id,
                                    uint64_t *d_pts3, uint64_t *d_out_share, uint64_t *d_out_ring, uint8_t *d_wire_r2f,
                                    uint8_t *d_wire_f2r, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_rcm_r2f && d_rcm_f2r && delta && seed_r2f && seed_f2r && d_x && d_scales && key && d_pts3 && d_out_ring &&
                  d_wire_r2f && d_wire_f2r,
              MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 2 && bw <= 64 && shift < 64 && ctx->t >= 4 && n <= ((size_t)1 << 40), MMH_ERR_INVALID_ARG);
  hipStream_t s = (hipStream_t)stream;
  CHECK_RC(bn_alice_front_launch(d_rcm_r2f, d_rcm_f2r, delta, seed_r2f, seed_f2r, gid0, d_x, n, bw, shift, ctx->t, d_scales,
                                 key, stream_id, d_pts3, n_ct, ctx->d_slot_map, sub, ctx->logN, ctx->bfv.t, d_out_share,
                                 d_out_ring, d_wire_r2f, d_wire_f2r, s));
  return ntt_dispatch(ctx, true, d_pts3, 3 * n_ct, 1, (int)ctx->L, s);   // BatchEncoder::encode's inverse transform, all three
}

extern "C" int mmh_bn_alice_finish(const mmh_ctx *ctx, const uint64_t *d_in_cts, const uint64_t *d_pts3, size_t n,
                                   uint64_t *d_out_cts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_in_cts && d_pts3 && d_out_cts, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N;
  void *sum = nullptr, *acc = nullptr, *pt_ntt = nullptr;
  int rc = MMH_OK;
  auto grab = [&](void **b, size_t bytes) {
    if (!rc && scratch_alloc(b, bytes, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  };
  grab(&sum, n_ct * 4 * N * 8);
  grab(&acc, n_ct * 4 * N * 8);
  grab(&pt_ntt, n_ct * 2 * N * 8);
  // hom_bn_ss.cc:263-267: add_plain of her share on load, both operands of multiply_plain to NTT form (one launch), the
  // dyadic product while the inverse transform loads (one), then mod-switch, add_plain of r and truncation (one)
  if (!rc) rc = ntt_bn_front_dispatch(ctx, d_in_cts, d_pts3, d_pts3 + n_ct * N, n_ct, (u64 *)sum, (u64 *)pt_ntt, s);
  if (!rc)
    rc = ntt_inv_product_each_dispatch(ctx, (const u64 *)pt_ntt, (const u64 *)sum, nullptr, nullptr, (u64 *)acc, n_ct, 2, s);
  if (!rc) {
    const size_t total = n_ct * 2 * N;
    hipLaunchKernelGGL(bn_finish_kernel, dim3(nblk(total)), dim3(256), 0, s, (const u64 *)acc, d_pts3 + 2 * n_ct * N,
                       d_out_cts, total, ctx->logN, ctx->bfv, ctx->d_mods);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  for (void *b : {sum, acc, pt_ntt})
    if (b) (void)scratch_free(b, s);
  return rc;
}

extern "C" int mmh_bole_server(const mmh_ctx *ctx, const uint64_t *d_enc_xy, const uint64_t *d_pts_yx, size_t n,
                               const uint64_t *d_random, uint64_t *d_out_cts, uint64_t *d_out_share,
                               void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_enc_xy && d_pts_yx && d_random && d_out_cts && d_out_share, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N;
  void *acc = nullptr, *ct_ntt = nullptr, *pt_ntt = nullptr;
  int rc = MMH_OK;
  auto grab = [&](void **b, size_t bytes) {
    if (!rc && scratch_alloc(b, bytes, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  };
  grab(&acc, n_ct * 4 * N * 8);
  grab(&ct_ntt, 2 * n_ct * 4 * N * 8);
  grab(&pt_ntt, 2 * n_ct * 2 * N * 8);
  // hom_bn_ss.cc:304-306: out[i] = ex[i]*py[i] + ey[i]*px[i]; operands are [ex; ey] and [py; px].  Both to NTT form,
  // the two dyadic products and their sum formed while the inverse transform loads.
  if (!rc && hipMemcpyAsync(ct_ntt, d_enc_xy, 2 * n_ct * 4 * N * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
    rc = MMH_ERR_INTERNAL;
  if (!rc) rc = ntt_dispatch(ctx, false, (u64 *)ct_ntt, 2 * n_ct * 2, 2, 0, s);
  if (!rc) rc = mmh_plain_lift_ntt(ctx, 0, d_pts_yx, 2 * n_ct, (u64 *)pt_ntt, stream);
  if (!rc)
    rc = ntt_inv_product_each_dispatch(ctx, (const u64 *)pt_ntt, (const u64 *)ct_ntt, (const u64 *)pt_ntt + n_ct * 2 * N,
                                       (const u64 *)ct_ntt + n_ct * 4 * N, (u64 *)acc, n_ct, 2, s);
  if (!rc) rc = bn_tail(ctx, (const u64 *)acc, n, n_ct, sub, d_random, d_out_cts, d_out_share, s);
  for (void *b : {acc, ct_ntt, pt_ntt})
    if (b) (void)scratch_free(b, s);
  return rc;
}

extern "C" int mmh_bn_decrypt_vector(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_cts, size_t n,
                                     uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_sk_ntt && d_cts && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr, *slots = nullptr;
  MMH_HIP(scratch_alloc(&pts, (n_ct << ctx->logN) * 8, s));
  int rc = MMH_OK;
  if (scratch_alloc(&slots, (n_ct << ctx->logN) * 8, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  if (!rc) rc = mmh_decrypt(ctx, 1, d_sk_ntt, d_cts, n_ct, (u64 *)pts, stream);  // hom_bn_ss.cc:389
  if (!rc) rc = mmh_batch_decode(ctx, (const u64 *)pts, n_ct, (u64 *)slots, stream);
  if (!rc) {
    hipLaunchKernelGGL(chunk_unpad_kernel, dim3(nblk(n)), dim3(256), 0, s, (const u64 *)slots, d_out, n, sub,
                       ctx->logN);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  (void)scratch_free(pts, s);
  if (slots) (void)scratch_free(slots, s);
  return rc;
}

// ---- conversions fused with the neighbouring vector step (same bits as the two calls they replace)
extern "C" int mmh_bn_encode_vector_r2f(const mmh_ctx *ctx, int party, const uint64_t *d_tmp, const uint8_t *d_msb,
                                        const uint64_t *d_y, size_t n, int bw, uint64_t *d_pts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_tmp && d_msb && d_y && d_pts, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE((party == 1 || party == 2) && bw >= 2 && bw <= 64, MMH_ERR_INVALID_ARG);
  hipStream_t s = (hipStream_t)stream;
  const u64 p = ctx->t, mask = bw >= 64 ? ~u64(0) : (u64(1) << bw) - 1;
  hipLaunchKernelGGL(r2f_post_scatter_kernel, dim3(nblk(n_ct << ctx->logN)), dim3(256), 0, s, party, d_tmp, d_msb, d_y,
                     d_pts, n_ct, n, sub, ctx->logN, ctx->d_slot_map, ctx->bfv.t, mask % p + 1, (u64(1) << (bw - 2)) % p);
  MMH_HIP(hipGetLastError());
  return ntt_dispatch(ctx, true, d_pts, n_ct, 1, (int)ctx->L, s);   // BatchEncoder::encode's inverse transform mod t
}
// ALICE's Ring2Field in front of CheetahLinear::bn as the COT sender (library_fixed_uniform_cheetah.cpp:567-600,
// truncation.cpp:518-581, silent_ot.h:129-163), through to the plaintexts bn adds to the client's ciphertexts: mmh_r2f_pre,
// mmh_cot_send (mod p) and mmh_bn_encode_vector_r2f in one launch plus BatchEncoder's inverse transform; same d_pts and
// the same wire bytes as the three calls.
extern "C" int mmh_cot_send_r2f_encode(const mmh_ctx *ctx, const uint64_t *d_rcm, const uint64_t delta[2],
                                       const uint64_t seed[2], uint64_t gid0, const uint64_t *d_x, size_t n, int bw,
                                       uint64_t *d_pts, uint8_t *d_wire, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_rcm && delta && seed && d_x && d_pts && d_wire, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(bw >= 2 && bw <= 64 && n <= ((size_t)1 << 40), MMH_ERR_INVALID_ARG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N;
  if (n != n_ct * N)   // the slots past the last element of a short last ciphertext hold zeros
    MMH_HIP(hipMemsetAsync(d_pts + (n_ct - 1) * N, 0, N * 8, s));
  CHECK_RC(cot_send_r2f_encode_launch(d_rcm, delta, seed, gid0, d_x, n, bw, ctx->t, d_pts, ctx->d_slot_map, sub, ctx->logN,
                                      ctx->bfv.t, d_wire, s));
  return ntt_dispatch(ctx, true, d_pts, n_ct, 1, (int)ctx->L, s);   // BatchEncoder::encode's inverse transform mod t
}
extern "C" int mmh_bn_encrypt_vector_r2f(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, int party, const uint64_t *d_tmp,
                                         const uint8_t *d_msb, const uint64_t *d_y, size_t n, int bw,
                                         const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_sk_ntt && d_a && d_e && d_cts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (n_ct << ctx->logN) * 8, s));
  int rc = mmh_bn_encode_vector_r2f(ctx, party, d_tmp, d_msb, d_y, n, bw, (u64 *)pts, stream);
  if (!rc) rc = mmh_encrypt_symmetric(ctx, 0, d_sk_ntt, d_a, d_e, (const u64 *)pts, n_ct, d_cts, stream);
  (void)scratch_free(pts, s);
  return rc;
}
extern "C" int mmh_bn_decrypt_vector_f2r(const mmh_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_cts, size_t n,
                                         int party, int shift, uint64_t *d_tmp, uint8_t *d_msb, uint64_t *d_corr,
                                         void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  size_t sub, n_ct;
  CHECK_RC(bn_counts(ctx, n, &sub, &n_ct));
  MMH_REQUIRE(d_sk_ntt && d_cts && d_tmp && d_msb, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE((party == 1 || party == 2) && shift >= 0 && shift < 64, MMH_ERR_INVALID_ARG);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (n_ct << ctx->logN) * 8, s));
  int rc = mmh_decrypt(ctx, 1, d_sk_ntt, d_cts, n_ct, (u64 *)pts, stream);  // hom_bn_ss.cc:389
  // BatchEncoder::decode's forward transform mod t in place, then the gather in slot order straight into the pre step
  if (!rc) rc = ntt_dispatch(ctx, false, (u64 *)pts, n_ct, 1, (int)ctx->L, s);
  if (!rc) {
    hipLaunchKernelGGL(gather_f2r_pre_kernel, dim3(nblk(n)), dim3(256), 0, s, party, (const u64 *)pts, n, sub, ctx->logN,
                       ctx->d_slot_map, ctx->t, ctx->t >> shift, d_tmp, d_msb, d_corr);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  (void)scratch_free(pts, s);
  return rc;
}

// ------------------------------------------------------------------- HomBNSS, direct path
// Coefficient-packed per-channel scaling (hom_bn_ss.cc:471-790).  getSplitBN :61-67 never splits
// channels: tile (c, hb, wb) of floor(sqrt(N))-sized squares holds entry (h, w) at coefficient
// h*sw + w, zero where the tile sticks out of the tensor.  The re-sharing and decrypt sides index
// the *valid* entries of a tile densely instead (0 .. vh*vw-1, :666-678 and :769-781); both agree
// with the encoder only when no tile is ragged in w.  Restated as written.
namespace mmh {
struct BnTiles {
  long C, H, W, sh, sw, dH, dW;
  size_t n_ct;
};

__global__ __launch_bounds__(256) void bn_tensor_encode_kernel(const u64 *__restrict__ tensor,
                                                               u64 *__restrict__ polys, size_t total, int logN,
                                                               BnTiles g, ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long cid = (long)(idx >> logN), i = (long)(idx & ((size_t(1) << logN) - 1));
  u64 v = 0;
  if (i < g.sh * g.sw) {
    const long c = cid / (g.dH * g.dW), hh = (cid / g.dW) % g.dH * g.sh + i / g.sw, ww = cid % g.dW * g.sw + i % g.sw;
    if (hh < g.H && ww < g.W) v = barrett128(0, tensor[(c * g.H + hh) * g.W + ww], t.q, t.barrett_hi, t.barrett_lo);
  }
  polys[idx] = v;
}

// bn_direct :613-645: every polynomial of channel c times (+/-)(|scales[c]| mod t) mod q_l
__global__ __launch_bounds__(256) void bn_scale_kernel(u64 *__restrict__ cts, const u64 *__restrict__ scales,
                                                       size_t total, int logN, int limbs, long tiles_per_channel,
                                                       ModC t, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n_ct][2][limbs][N]
  if (idx >= total) return;
  const size_t poly = idx >> logN;
  const int l = (int)(poly % limbs);
  const long c = (long)(poly / (2 * limbs)) / tiles_per_channel;
  const int64_t s = (int64_t)scales[c];
  const u64 mag = s < 0 ? 0 - (u64)s : (u64)s;
  const DevMod &q = mods[l];
  u64 scalar = reduce64(barrett128(0, mag, t.q, t.barrett_hi, t.barrett_lo), q);
  if (s < 0 && scalar) scalar = q.q - scalar;
  cts[idx] = mul_mod(cts[idx], scalar, q);
}

// keep[(hb, wb)][i] = i < vh*vw: the "used" coefficients of bn_direct :666-681
__global__ __launch_bounds__(256) void bn_keep_kernel(uint8_t *__restrict__ keep, size_t total, int logN, BnTiles g) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long tile = (long)(idx >> logN), i = (long)(idx & ((size_t(1) << logN) - 1));
  const long vh = min(g.sh, g.H - tile / g.dW * g.sh), vw = min(g.sw, g.W - tile % g.dW * g.sw);
  keep[idx] = i < vh * vw;
}

// decryptToTensor :769-781: valid entries of a tile are read from consecutive coefficients
__global__ __launch_bounds__(256) void bn_tensor_gather_kernel(const u64 *__restrict__ pts, u64 *__restrict__ out,
                                                               size_t total, int logN, BnTiles g) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [C][H][W]
  if (idx >= total) return;
  const long ww = (long)(idx % g.W), hh = (long)(idx / g.W % g.H), c = (long)(idx / (g.W * g.H));
  const long hb = hh / g.sh, wb = ww / g.sw, vw = min(g.sw, g.W - wb * g.sw);
  const size_t cid = (size_t)((c * g.dH + hb) * g.dW + wb);
  out[idx] = pts[(cid << logN) + (hh % g.sh) * vw + ww % g.sw];
}
}  // namespace mmh

static int bn_tiles(const mmh_ctx *ctx, const mmh_bn_tensor_meta *m, BnTiles *g) {
  MMH_REQUIRE(ctx && m, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(m->c > 0 && m->h > 0 && m->w > 0, MMH_ERR_CONFIG);
  long n = (long)std::sqrt((double)ctx->N);  // getSplitBN :63
  while (n * n > (long)ctx->N) --n;
  while ((n + 1) * (n + 1) <= (long)ctx->N) ++n;
  g->C = m->c, g->H = m->h, g->W = m->w;
  g->sh = std::min(n, g->H), g->sw = std::min(n, g->W);
  g->dH = (g->H + g->sh - 1) / g->sh, g->dW = (g->W + g->sw - 1) / g->sw;
  g->n_ct = (size_t)(g->C * g->dH * g->dW);
  return MMH_OK;
}

extern "C" int mmh_bn_tensor_count(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, size_t *n_cts,
                                   int64_t split[3]) {
  BnTiles g;
  CHECK_RC(bn_tiles(ctx, meta, &g));
  if (n_cts) *n_cts = g.n_ct;
  if (split) split[0] = 1, split[1] = g.sh, split[2] = g.sw;
  return MMH_OK;
}

extern "C" int mmh_bn_encode_tensor(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_tensor,
                                    uint64_t *d_pts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  BnTiles g;
  CHECK_RC(bn_tiles(ctx, meta, &g));
  MMH_REQUIRE(ctx->t >= 2, MMH_ERR_CONFIG);
  MMH_REQUIRE(d_tensor && d_pts, MMH_ERR_NULL_POINTER);
  const size_t total = g.n_ct << ctx->logN;
  hipLaunchKernelGGL(bn_tensor_encode_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, d_tensor, d_pts,
                     total, ctx->logN, g, ctx->bfv.t);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_bn_encrypt_tensor(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_sk_ntt,
                                     const uint64_t *d_tensor, const uint64_t *d_a, const int64_t *d_e,
                                     uint64_t *d_cts, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  BnTiles g;
  CHECK_RC(bn_tiles(ctx, meta, &g));
  MMH_REQUIRE(d_sk_ntt && d_tensor && d_a && d_e && d_cts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (g.n_ct << ctx->logN) * 8, s));
  int rc = mmh_bn_encode_tensor(ctx, meta, d_tensor, (u64 *)pts, stream);
  if (!rc) rc = mmh_encrypt_symmetric(ctx, 0, d_sk_ntt, d_a, d_e, (const u64 *)pts, g.n_ct, d_cts, stream);
  (void)scratch_free(pts, s);
  return rc;
}

extern "C" int mmh_bn_direct_server(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_in_cts,
                                    const uint64_t *d_share_pts, const uint64_t *d_scales,
                                    const uint64_t *d_pk_ntt, const int64_t *d_u, const int64_t *d_e0,
                                    const int64_t *d_e1, const uint64_t *d_share1, uint64_t *d_out_cts,
                                    void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  BnTiles g;
  CHECK_RC(bn_tiles(ctx, meta, &g));
  MMH_REQUIRE(d_in_cts && d_scales && d_pk_ntt && d_u && d_e0 && d_e1 && d_share1 && d_out_cts, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!meta->is_shared_input || d_share_pts, MMH_ERR_DIM_MISMATCH);
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N, n = g.n_ct, S = (size_t)(g.dH * g.dW);
  void *acc = nullptr, *zero = nullptr, *zero_low = nullptr, *rnd = nullptr, *keep = nullptr;
  int rc = MMH_OK;
  auto grab = [&](void **b, size_t bytes) {
    if (!rc && scratch_alloc(b, bytes, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  };
  grab(&acc, n * 4 * N * 8);
  grab(&zero, n * 4 * N * 8);
  grab(&zero_low, n * 2 * N * 8);
  grab(&rnd, n * N * 8);
  grab(&keep, S * N);
  // step 1 (:598-610): reconstruct the encrypted tensor
  if (!rc && meta->is_shared_input) rc = mmh_add_plain(ctx, 0, d_in_cts, d_share_pts, n, 0, (u64 *)acc, stream);
  if (!rc && !meta->is_shared_input &&
      hipMemcpyAsync(acc, d_in_cts, n * 4 * N * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
    rc = MMH_ERR_INTERNAL;
  // step 2 (:613-645): channel-wise scalar
  if (!rc) {
    const size_t total = n * 4 * N;
    hipLaunchKernelGGL(bn_scale_kernel, dim3(nblk(total)), dim3(256), 0, s, (u64 *)acc, d_scales, total, ctx->logN, 2,
                       g.dH * g.dW, ctx->bfv.t, ctx->d_mods);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  // step 3 (:648-687): mod-switch, minus Delta*encode(share1), plus a public-key encryption of zero
  // (Encryptor::encrypt_zero at the last level = encrypt at the level above, then divide_and_round_q_last)
  if (!rc) rc = mmh_mod_switch_to_last(ctx, (const u64 *)acc, n, d_out_cts, stream);
  if (!rc) rc = mmh_bn_encode_tensor(ctx, meta, d_share1, (u64 *)rnd, stream);
  if (!rc) rc = mmh_add_plain(ctx, 1, d_out_cts, (const u64 *)rnd, n, 1, d_out_cts, stream);
  if (!rc) rc = mmh_encrypt_zero_asym(ctx, 0, d_pk_ntt, d_u, d_e0, d_e1, n, (u64 *)zero, stream);
  if (!rc) rc = mmh_mod_switch_to_last(ctx, (const u64 *)zero, n, (u64 *)zero_low, stream);
  if (!rc) rc = mmh_ct_add(ctx, 1, d_out_cts, (const u64 *)zero_low, n, stream);
  if (!rc) {
    hipLaunchKernelGGL(bn_keep_kernel, dim3(nblk(S * N)), dim3(256), 0, s, (uint8_t *)keep, S * N, ctx->logN, g);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  if (!rc) rc = mmh_truncate_and_zero(ctx, d_out_cts, n, (const uint8_t *)keep, S, stream);
  for (void *b : {acc, zero, zero_low, rnd, keep})
    if (b) (void)scratch_free(b, s);
  return rc;
}

extern "C" int mmh_bn_decrypt_tensor(const mmh_ctx *ctx, const mmh_bn_tensor_meta *meta, const uint64_t *d_sk_ntt,
                                     const uint64_t *d_cts, uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(ctx);
  BnTiles g;
  CHECK_RC(bn_tiles(ctx, meta, &g));
  MMH_REQUIRE(d_sk_ntt && d_cts && d_out, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (g.n_ct << ctx->logN) * 8, s));
  int rc = mmh_decrypt(ctx, 1, d_sk_ntt, d_cts, g.n_ct, (u64 *)pts, stream);
  if (!rc) {
    const size_t total = (size_t)(g.C * g.H * g.W);
    hipLaunchKernelGGL(bn_tensor_gather_kernel, dim3(nblk(total)), dim3(256), 0, s, (const u64 *)pts, d_out, total,
                       ctx->logN, g);
    if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
  }
  (void)scratch_free(pts, s);
  return rc;
}
