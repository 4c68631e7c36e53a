// HomConv2DSS on device: tensor <-> polynomial encoders and the fused server / client layer
// calls (include/gemini/cheetah/hom_conv2d_ss.cc, tensor_encoder.cc).  A plan holds the
// host-computed geometry (geometry.hpp) as small device tables; every layer call is then a
// short chain of batched kernels with no per-ciphertext host loop.
#include "bfv_device.hpp"
#include "common.hpp"
#include "geometry.hpp"
#include "host_math.hpp"
#include "ntt_device.hpp"

#include <cstdio>
#include <cstdlib>

namespace mmh {
int ntt_dispatch(const mmh_ctx *ctx, bool inverse, u64 *d_polys, size_t n_polys, int limbs,
                 int mod_base, hipStream_t s);
int ntt_lift_dispatch(const mmh_ctx *ctx, const u64 *d_src, u64 *d_out, size_t n_polys, int limbs,
                      hipStream_t s);
int reshare_tail_approx(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_pk_ntt_last, const int64_t *d_u,
                        const int64_t *d_e1, const u64 *d_mask, const uint8_t *d_keep, size_t keep_period,
                        u64 *d_out, hipStream_t s, const int *d_share_dst, size_t share_plane, u64 *d_share_out);
// (d_u_ntt: u already in NTT form at the last level, or null)
int reshare_tail_approx_u(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_pk_ntt_last,
                          const int64_t *d_u, const u64 *d_u_ntt, const int64_t *d_e1, const u64 *d_mask,
                          const uint8_t *d_keep, size_t keep_period, u64 *d_out, hipStream_t s, const int *d_share_dst,
                          size_t share_plane, u64 *d_share_out, bool acc_switched = false);
int reshare_tail_online(const mmh_ctx *ctx, u64 *d_acc, bool acc_ntt_form, size_t n, const u64 *d_zero,
                        const int64_t *d_e1, const u64 *d_mask, const uint8_t *d_keep, size_t keep_period, u64 *d_out,
                        hipStream_t s, const int *d_share_dst, size_t share_plane, u64 *d_share_out,
                        bool acc_switched = false);
size_t reshare_split_max();   // output ciphertexts up to which the tail takes its split form (bfv.hip)
int ntt_filter_dispatch(const mmh_ctx *ctx, const u64 *d_filters, u64 *d_out, size_t n_filters, int C, int fh,
                        int fw, int wc, int n_cslices, long origin, long plane, long row, int limbs,
                        hipStream_t s);

// one input window (image polynomial): where it reads the strided image and its zero margins
struct ImgWindow {
  int c0, y0, x0;          // first covered entry (strided-image coordinates)
  int nc, nh, nw;          // window size before mocking
  int lo_y, hi_y, lo_x, hi_x;
};
struct PwChannel { int row, e; };  // direct route: filter tap -> input ciphertext row (channel slice) and rotation
}  // namespace mmh

struct mmh_conv_plan {
  const mmh_ctx *ctx = nullptr;
  mmh_conv_meta meta{};
  mmh::Dim3 in, f;
  mmh::ConvSplit split;
  int sub_h = 1, sub_w = 1;             // Strided3DTensor strides
  mmh::AxisCut cut_c, cut_h, cut_w;      // cuts of the image that is encoded
  size_t n_in = 0, n_cslices = 0, n_spatial = 0, n_out = 0;
  long out_h = 0, out_w = 0;             // result tensor is n_filters x out_h x out_w
  long filt_origin = 0;                  // FilterIndexer::index_begin for a full channel window
  size_t n_targets = 0;                  // targets summed over the spatial windows
  // device tables
  mmh::ImgWindow *d_windows = nullptr;   // [n_in]
  uint32_t *d_tgt_coef = nullptr;        // [n_targets] coefficient index
  uint32_t *d_tgt_dst = nullptr;         // [n_targets] offset inside one out_h x out_w plane
  uint32_t *d_tgt_slice = nullptr;       // [n_targets] spatial window the target belongs to
  uint8_t *d_keep = nullptr;             // [n_spatial][N] coefficients that survive remove_unused_coeffs
  int *d_share_dst = nullptr;            // [n_spatial][N] offset of a target coefficient in the result plane, else -1
  // compact wire form of the result ciphertexts (mmh_conv_result_pack): per spatial window the kept coefficients of
  // c0 in ascending order, and where a coefficient sits in that list
  uint32_t *d_kept_idx = nullptr;        // [sum of kept] coefficient index
  uint32_t *d_kept_off = nullptr;        // [n_spatial + 1] first entry of a window's list
  int *d_kept_rank = nullptr;            // [n_spatial][N] position in the window's list, else -1
  size_t max_kept = 0;                   // longest list
  // 1x1 filters with few channels per polynomial: multiply-accumulate as a mod-q matrix product over
  // channels (pointwise_mac below) instead of transforming M*Cs sparse filter polynomials
  bool direct_capable = false;           // tables below are built
  bool direct_fresh = false;             // the direct product is the cheaper route when the filters arrive with the call
  bool direct_prepared = false;          // ... and when they were prepared (no filter transforms on the other route)
  mmh::PwChannel *d_pw_chan = nullptr;   // [C rounded up to 64]: input row (channel slice) and rotation of a channel
  int *d_pw_live = nullptr;              // [pw_live_tiles] the 64-column tiles the product computes (index = column / 64)
  size_t pw_live_tiles = 0;
  size_t pw_live_tiles_limb[2] = {0, 0};  // of those, the tiles of limb 0 / limb 1 (mmh_conv_plan_work)
  size_t pw_taps_padded = 0;             // taps rounded up to the product kernel's 64-tap step
  std::vector<void *> allocs;
};

// filters of one layer as the server call consumes them (mmh_conv_filters_prepare)
struct mmh_conv_filters {
  const mmh_conv_plan *plan = nullptr;
  uint64_t *f_ntt = nullptr;                 // transform route: [M*Cs][2][N] lifted, NTT form
  uint32_t *wp = nullptr, *flag = nullptr;   // 1x1 route: biased weights, "needs the general form" flag
  uint64_t *wl = nullptr;                    //            lifted 64-bit weights
  uint8_t *w8 = nullptr;                     //            signed byte digits in matrix-core fragment order
  int form = -1;                             //            0 fast, 1 general (read back once at prepare), -1 unknown
};

namespace mmh {
__device__ __forceinline__ u64 red_t(u64 v, const ModC &t) { return barrett128(0, v, t.q, t.barrett_hi, t.barrett_lo); }

// EncodeImageShare: polynomial w, coefficient (c, y, x) of the mocked window
__global__ __launch_bounds__(256) void encode_image_kernel(const u64 *__restrict__ img, u64 *__restrict__ polys,
                                                           const ImgWindow *__restrict__ wins, size_t n_in, int logN,
                                                           int wc, int wh, int ww, int H, int W, int sub_h, int sub_w,
                                                           ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (n_in << logN)) return;
  const size_t k = idx >> logN;
  const int i = (int)(idx & ((size_t(1) << logN) - 1));
  u64 v = 0;
  if (i < wc * wh * ww) {
    const ImgWindow g = wins[k];
    const int c = i / (wh * ww), y = (i / ww) % wh, x = i % ww;
    if (c < g.nc && y >= g.lo_y && y + g.hi_y < g.nh && x >= g.lo_x && x + g.hi_x < g.nw) {
      const size_t cc = c + g.c0, yy = (size_t)(y - g.lo_y + g.y0) * sub_h, xx = (size_t)(x - g.lo_x + g.x0) * sub_w;
      v = red_t(img[(cc * H + yy) * W + xx], t);
    }
  }
  polys[idx] = v;
}

// EncodeImageShare fused with add_plain (conv2DSS :655-671 straight from the share tensor): out = ct + Delta * m
// with m the coefficient encode_image_kernel would have written; both polynomials of every ciphertext are
// written, so no copy and no plaintext buffer is needed.  Also clears `n_words` flag words of the call.
struct ImageShareArgs {
  const u64 *in, *img;
  u64 *out;
  const ImgWindow *wins;
  size_t n_in;
  int logN, wc, wh, ww, H, W, sub_h, sub_w;
};
__device__ __forceinline__ void image_plus_share_body(size_t idx, const ImageShareArgs &a, const BfvConsts &b,
                                                      const DevMod *__restrict__ mods) {
  if (idx >= (a.n_in << a.logN)) return;
  const int logN = a.logN;
  const size_t k = idx >> logN;
  const int i = (int)(idx & ((size_t(1) << logN) - 1));
  u64 m = 0;
  if (a.img && i < a.wc * a.wh * a.ww) {
    const ImgWindow g = a.wins[k];
    const int c = i / (a.wh * a.ww), y = (i / a.ww) % a.wh, x = i % a.ww;
    if (c < g.nc && y >= g.lo_y && y + g.hi_y < g.nh && x >= g.lo_x && x + g.hi_x < g.nw) {
      const size_t cc = c + g.c0, yy = (size_t)(y - g.lo_y + g.y0) * a.sub_h, xx = (size_t)(x - g.lo_x + g.x0) * a.sub_w;
      m = red_t(a.img[(cc * a.H + yy) * a.W + xx], b.t);
    }
  }
  const LevelConsts &c0 = b.lv[0];
  const u64 fix = m ? scale_fix(m, b, c0) : 0;
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const DevMod &q = mods[l];
    const size_t o0 = (((k * 2) * 2 + l) << logN) + i, o1 = (((k * 2 + 1) * 2 + l) << logN) + i;
    u64 v = a.in[o0];
    if (m) v = add_mod(v, scaled_plain(m, fix, c0.delta_mod_q[l], q), q.q);
    a.out[o0] = v;
    a.out[o1] = a.in[o1];
  }
}
__global__ __launch_bounds__(256) void image_plus_share_kernel(ImageShareArgs a, BfvConsts b,
                                                               const DevMod *__restrict__ mods,
                                                               u32 *__restrict__ words, int n_words,
                                                               u32 *__restrict__ one_more) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (size_t)n_words) words[idx] = 0;
  if (idx == 0 && one_more) *one_more = 0;
  image_plus_share_body(idx, a, b, mods);
}

// EncodeFilter: scatter filter m, channel c, tap (r, col) to its reversed coefficient
__global__ __launch_bounds__(256) void encode_filter_kernel(const u64 *__restrict__ filt, u64 *__restrict__ polys,
                                                            size_t total, int logN, int C, int fh, int fw, int wc,
                                                            int n_cslices, long origin, long plane, long row, ModC t) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [M][C][fh][fw]
  if (idx >= total) return;
  const int col = (int)(idx % fw), r = (int)((idx / fw) % fh);
  const size_t mc = idx / ((size_t)fw * fh);
  const int c = (int)(mc % C);
  const size_t m = mc / C;
  const int slice = c / wc, cc = c % wc;
  const long coef = origin - cc * plane - r * row - col;
  polys[((m * n_cslices + slice) << logN) + coef] = red_t(filt[idx], t);
}

// conv2DOneFilter :573-580 rejects a filter whose polynomials are all zero, i.e. whose weights are all
// 0 mod t: one workgroup per filter raises *flag when it finds such a filter
__global__ __launch_bounds__(256) void zero_filter_kernel(const u64 *__restrict__ filt, size_t per_filter, ModC t,
                                                          u32 *__restrict__ flag) {
  const u64 *__restrict__ f = filt + (size_t)blockIdx.x * per_filter;
  int nz = 0;
  for (size_t i = threadIdx.x; i < per_filter; i += 256) nz |= red_t(f[i], t) != 0;
  if (!__syncthreads_or(nz) && threadIdx.x == 0) *flag = 1u;   // every writer stores the same value: no atomic needed
}

// share tensor from the RNS mask: out[m][dst] = scale_and_round(r[m*S + slice][coef])
__global__ __launch_bounds__(256) void share_from_mask_kernel(const u64 *__restrict__ r, u64 *__restrict__ out,
                                                              size_t n_filters, size_t n_targets, size_t n_spatial,
                                                              size_t plane, int logN, const u32 *__restrict__ coef,
                                                              const u32 *__restrict__ dst, const u32 *__restrict__ slice,
                                                              BfvConsts b, const DevMod *__restrict__ mods) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_filters * n_targets) return;
  const size_t m = idx / n_targets, g = idx % n_targets;
  const u64 x[2] = {r[((m * n_spatial + slice[g]) << logN) + coef[g]], 0};  // last level: one limb
  out[m * plane + dst[g]] = behz_round(x, b, b.lv[1], mods);
}
// gather decrypted coefficients: out[m][dst] = pts[m*S + slice][coef]
__global__ __launch_bounds__(256) void gather_targets_kernel(const u64 *__restrict__ pts, u64 *__restrict__ out,
                                                             size_t n_filters, size_t n_targets, size_t n_spatial,
                                                             size_t plane, int logN, const u32 *__restrict__ coef,
                                                             const u32 *__restrict__ dst,
                                                             const u32 *__restrict__ slice) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_filters * n_targets) return;
  const size_t m = idx / n_targets, g = idx % n_targets;
  out[m * plane + dst[g]] = pts[((m * n_spatial + slice[g]) << logN) + coef[g]];
}

int launch_share_from_mask(const mmh_ctx *ctx, const u64 *r, u64 *out, size_t groups, size_t n_targets,
                           size_t n_spatial, size_t plane, const u32 *coef, const u32 *dst, const u32 *slice,
                           hipStream_t s) {
  const size_t total = groups * n_targets;
  if (!total) return MMH_OK;
  hipLaunchKernelGGL(share_from_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, r, out, groups,
                     n_targets, n_spatial, plane, ctx->logN, coef, dst, slice, ctx->bfv, ctx->d_mods);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
int launch_gather_targets(const mmh_ctx *ctx, const u64 *pts, u64 *out, size_t groups, size_t n_targets,
                          size_t n_spatial, size_t plane, const u32 *coef, const u32 *dst, const u32 *slice,
                          hipStream_t s) {
  const size_t total = groups * n_targets;
  if (!total) return MMH_OK;
  hipLaunchKernelGGL(gather_targets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pts, out, groups,
                     n_targets, n_spatial, plane, ctx->logN, coef, dst, slice);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}


// ------------------------------------------------------------------ 1x1 filters, few channels per polynomial
// EncodeFilter gives a 1x1 filter polynomial one monomial per channel of its slice:
//   f[m][s] = sum_c w[m][s*wc + c] X^(origin - c*plane),
// so  sum_s ct[s] * f[m][s]  =  sum_ch w[m][ch] * (X^e(ch) * ct[s(ch)])  over all channels, and X^e * ct is
// the ciphertext rotated by e coefficients with the wrapped ones negated (X^N = -1).  The layer is then the
// mod-q matrix product W[M x C] * U[C x columns] with scalar weights on the coefficient form: no filter
// polynomial is built, transformed or streamed (M*Cs*64 KiB in the dyadic form), the input needs no
// forward and the result no inverse transform, and of c0 only the coefficients that survive
// removeUnusedCoeffs are computed (c1 keeps all N: hom_conv2d_ss.cc:828 walks ct.data(0) only).
// Everything is exact mod q_l, so the kept coefficients equal the per-term multiply_plain schedule bit
// for bit.  Cheaper than the transform route while wc is small (per output column C scalar MACs against
// Cs * (2 NTTs + 4 dyadic MACs)); measured break-even near 28 channels per polynomial.
constexpr int kPwBiasLog = 20;  // fast form: weights in (-2^20, 2^20) stored as w + 2^20

// weights [M][K] (mod t; K = C*fh*fw taps) -> biased 32-bit form wp[Cpad][Mpad], lifted 64-bit form
// wl[2][Cpad][Mpad], signed byte digits w8 in matrix-core fragment order (pw_mfma_kernel), and a flag word: bit 0
// when some weight does not fit the fast forms (|w| >= 2^20), bit 1 when some weight needs a third byte digit.
// One thread per filter and group of 16 taps: a 128-byte run of the raw tensor in, one 16-byte fragment per digit
// out.  nzmap (optional, [M], cleared by the caller): word m becomes 1 when filter m has a weight that is not 0
// mod t (the conv2DOneFilter :573-580 rule is evaluated from it by pw_expand_kernel).
// FROM_PTS: read the weights back from encoded filter plaintexts instead of the raw tensor.
// w8 layout: [m / 128][k / 64][digit 3][(m % 128) / 16][lane = ((k % 64) / 16) << 4 | m % 16][k % 16].
struct PwWeightArgs {
  const u64 *src;
  u32 *wp;
  u64 *wl;
  uint4 *w8;
  int M, Mpad, Mpad128, C, Cpad, Cpad64;
  const PwChannel *chan;
  int n_cslices, logN;
  ModC t;
  u64 thr, q0, q1;
};
// one thread's filter and group of 16 taps; returns the flag bits in bits 0-1 and "has a weight != 0 mod t" in bit 2
template <bool FROM_PTS>
__device__ __forceinline__ u32 pw_weights_body(size_t gid, const PwWeightArgs &a) {
  if (gid >= (size_t)(a.Cpad64 / 16) * a.Mpad128) return 0;
  const int M = a.M, Mpad = a.Mpad, C = a.C, Cpad = a.Cpad;
  const ModC &t = a.t;
  const int m = (int)(gid % a.Mpad128), kg = (int)(gid / a.Mpad128);
  u32 dig[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  u32 bits = 0;
  bool any = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int ch = kg * 16 + j;
    u64 v = 0;  // filters and taps beyond the real ones weigh nothing
    if (m < M && ch < C) {
      if (FROM_PTS) {
        const PwChannel pc = a.chan[ch];   // the tap's coefficient in the polynomial of its channel slice
        v = a.src[(((size_t)m * a.n_cslices + pc.row) << a.logN) + pc.e];
      } else {
        v = red_t(a.src[(size_t)m * C + ch], t);
      }
    }
    any |= v != 0;
    const bool neg = v >= a.thr;
    const u64 mag = neg ? t.q - v : v;
    const bool fits = mag < (u64(1) << kPwBiasLog);
    // balanced byte digits of the signed weight: w = d0 + 256 d1 + 65536 d2, d0 and d1 in [-128, 127]
    const int w = fits ? (neg ? -(int)mag : (int)mag) : 0;
    const int wb = w + 0x8080;
    const int d2 = wb >> 16;
    if (!fits) bits |= 1u;
    if (d2 != 0) bits |= 2u;
    const int sh = 8 * (j & 3);
    dig[0][j >> 2] |= (u32)((wb & 0xff) ^ 0x80) << sh;
    dig[1][j >> 2] |= (u32)(((wb >> 8) & 0xff) ^ 0x80) << sh;
    dig[2][j >> 2] |= (u32)(d2 & 0xff) << sh;
    if (m < Mpad && ch < Cpad) {   // (a call that multiplies on the matrix cores leaves both forms out: wp, wl null)
      const size_t idx = (size_t)ch * Mpad + m;
      if (a.wp) a.wp[idx] = (u32)((u64(1) << kPwBiasLog) + (neg ? 0 - mag : mag));
      if (a.wl) {
        a.wl[idx] = neg ? v + (a.q0 - t.q) : v;                            // fast plain lift, limb 0
        a.wl[(size_t)Cpad * Mpad + idx] = neg ? v + (a.q1 - t.q) : v;      // limb 1
      }
    }
  }
  const int kc = kg >> 2;
  const size_t unit = ((((size_t)(m >> 7) * (a.Cpad64 >> 6) + kc) * 3) * 8 + ((m & 127) >> 4)) * 64 + (((kg & 3) << 4) | (m & 15));
#pragma unroll
  for (int d = 0; d < 3; ++d) a.w8[unit + (size_t)d * 8 * 64] = make_uint4(dig[d][0], dig[d][1], dig[d][2], dig[d][3]);
  return bits | (any && m < M ? 4u : 0u);
}
template <bool FROM_PTS>
__global__ __launch_bounds__(256) void pw_weights_kernel(PwWeightArgs a, u32 *__restrict__ flag,
                                                         u32 *__restrict__ nzmap) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const u32 r = pw_weights_body<FROM_PTS>(gid, a);
  if (r & 3u) atomicOr(flag, r & 3u);
  if (nzmap && (r & 4u)) {
    const int m = (int)(gid % a.Mpad128);
    if (nzmap[m] == 0) atomicOr(nzmap + m, 1u);
  }
}

// The front of a direct-route layer call from the raw tensors as ONE launch (CheetahLinear::conv2d's server half up
// to the products, cheetah-api.cpp:417-423): workgroups [0, n_img) build the image (encodeImage fused with
// add_plain), the others the weight forms.  Nothing here needs a cleared word: every weights workgroup leaves its flag
// bits in a word of its own (wflag) and every thread its "has a non-zero weight" byte (anymap, [taps / 16][Mpad128]);
// pw_expand_kernel, the next launch, folds them into the one flag word and the zero-filter verdict.
// Workgroups [0, n_u) beside them transform the ternary u of the re-sharing tail (encrypt_zero's first step,
// hom_conv2d_ss.cc:302-306), which depends on nothing else in the call: for the small layers of a network its launch
// was a tenth of the call.
// A call that draws its own randomness (mmh_conv_server_tensors_sampled) has no sampling launch in front of this one:
// the u workgroups draw the ternary polynomial they transform (SRC_TERNARY), and two more ranges of workgroups behind
// the weights draw e1 and the mask for the tail, each from the stream and tile mmh_sample_reshare would use.  Where the
// tail takes its split form (few output ciphertexts), the u workgroups -- two per ciphertext then -- go on to the
// product with pk_j and the inverse transform (ntt_zero_unit12), so the tail's own transform launch falls away too.
struct FrontDraw {
  int mode;              // 0: u from memory; 1: u drawn in place -> u_ntt; 2: drawn, -> the encryption of zero [n][2][N]
  const u64 *pk;         // mode 2: pk at the last level [2][N], NTT form
  u64 *zero;
  int64_t *e1;           // modes 1, 2: e1 and the mask [n][N], drawn by the last n_e + n_m workgroups
  u64 *mask;
  unsigned n_e, n_m, lo_e, lo_m;   // tiles of the noise / uniform streams that meet [first, first + total)
  size_t first, total;
};
template <int LOGN>
__global__ __launch_bounds__(256) void conv_front_kernel(ImageShareArgs ia, BfvConsts b, const DevMod *__restrict__ mods,
                                                         unsigned n_u, unsigned n_img, PwWeightArgs wa,
                                                         u32 *__restrict__ wflag, uint8_t *__restrict__ anymap,
                                                         u32 *__restrict__ zero_filter, u64 *__restrict__ u_ntt,
                                                         NttSource usrc, const ulonglong2 *__restrict__ tables,
                                                         unsigned n_w, FrontDraw dr) {
  extern __shared__ __align__(16) u64 front_lds[];
  if (blockIdx.x < n_u) {
    if (dr.mode == 2) {
      if (LOGN == 12)
        ntt_zero_unit12<true>(blockIdx.x >> 1, mods[0], tables, usrc, dr.pk + ((size_t)(blockIdx.x & 1) << LOGN),
                              dr.zero + ((size_t)blockIdx.x << LOGN), front_lds);
    } else if (dr.mode == 1) {
      ntt_fwd_unit<LOGN, true, SRC_TERNARY>(blockIdx.x, u_ntt, mods, tables, 1, 0, usrc, front_lds);
    } else {
      ntt_fwd_unit<LOGN, true, SRC_SIGNED>(blockIdx.x, u_ntt, mods, tables, 1, 0, usrc, front_lds);
    }
    return;
  }
  if (blockIdx.x >= n_u + n_img + n_w) {   // the tail's e1 and mask (streams sid + 1, sid + 2 of the draw)
    const unsigned d = blockIdx.x - n_u - n_img - n_w;
    if (d < dr.n_e) noise_block(usrc.key, usrc.sid + 1, 0, dr.lo_e + d, threadIdx.x, dr.e1, dr.total, dr.first);
    else uniform_block(usrc.key, usrc.sid + 2, 0, dr.lo_m + (d - dr.n_e), threadIdx.x, dr.mask, dr.total, LOGN, 1, mods, dr.first);
    return;
  }
  if (blockIdx.x < n_u + n_img) {
    const size_t idx = (size_t)(blockIdx.x - n_u) * 256 + threadIdx.x;
    if (idx == 0 && zero_filter) *zero_filter = 0;
    image_plus_share_body(idx, ia, b, mods);
    return;
  }
  const unsigned wb = blockIdx.x - n_u - n_img;
  const size_t gid = (size_t)wb * 256 + threadIdx.x;
  const u32 r = pw_weights_body<false>(gid, wa);
  if (anymap && gid < (size_t)(wa.Cpad64 / 16) * wa.Mpad128) anymap[gid] = (uint8_t)((r >> 2) & 1u);
  const int b0 = __syncthreads_or((int)(r & 1u)), b1 = __syncthreads_or((int)(r & 2u));
  if (threadIdx.x == 0) wflag[wb] = (b0 ? 1u : 0u) | (b1 ? 2u : 0u);
}

// Two filters against both halves of one column value: four independent v_mad_u64_u32 in one asm
// statement (hipcc pads every statement boundary).
__device__ __forceinline__ void mad_pair(u32 s_w0, u32 s_w1, u32 a0, u32 a1, u64 &x00, u64 &x01, u64 &x10,
                                         u64 &x11) {
  u64 carry;
  asm("v_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
      "v_mad_u64_u32 %1, %4, %5, %8, %1\n\t"
      "v_mad_u64_u32 %2, %4, %6, %7, %2\n\t"
      "v_mad_u64_u32 %3, %4, %6, %8, %3"
      : "+v"(x00), "+v"(x01), "+v"(x10), "+v"(x11), "=&s"(carry)
      : "v"(s_w0), "v"(s_w1), "v"(a0), "v"(a1));
}
// the same with the two weights in SGPRs (wave-uniform)
__device__ __forceinline__ void mad_pair_s(u32 s_w0, u32 s_w1, u32 a0, u32 a1, u64 &x00, u64 &x01, u64 &x10,
                                           u64 &x11) {
  u64 carry;
  asm("v_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
      "v_mad_u64_u32 %1, %4, %5, %8, %1\n\t"
      "v_mad_u64_u32 %2, %4, %6, %7, %2\n\t"
      "v_mad_u64_u32 %3, %4, %6, %8, %3"
      : "+v"(x00), "+v"(x01), "+v"(x10), "+v"(x11), "=&s"(carry)
      : "s"(s_w0), "s"(s_w1), "v"(a0), "v"(a1));
}
template <int TM>
struct alignas(4 * TM) PwWeights { u32 v[TM]; };

// acc[m*S + sp][p][l][i] = sum_ch W[m][ch] * (X^e(ch) * image[s(ch)*S + sp])[p][l][i] mod q_l.
// A workgroup owns a tile of 64 columns (sp, p, l, i..i+63) and 64 filters: its 256 threads stage the
// column values of 16 channels at a time in LDS (rotation and sign applied once, the loads of the next
// chunk in flight during the products of this one); every wave then multiplies the staged columns into
// its own 16 filters, whose weights are wave-uniform and arrive through scalar loads in bursts of four
// channels.  (Weights read as LDS broadcasts cost 1 KiB of LDS bandwidth per ds_read_b128 and made the
// loop LDS-bound: 36 LDS cycles against 32 VALU cycles per channel and wave.)
// Tiles of c0 without a kept coefficient write zeros and leave.
// Fast form: u = a0 + a1*2^30, acc0 += w'*a0, acc1 += w'*a1 with w' = w + 2^20 < 2^21: every sum of
// <= 4096 terms stays below 2^63, one v_mad_u64_u32 each and no carries; the bias is removed at the end
// with the column sums B0, B1:  sum w*u = (acc0 + acc1*2^30) - 2^20*(B0 + B1*2^30).
constexpr int kPwChunk = 16;  // channels staged per step

// Two instantiations are launched back to back; the device-side flag tells each whether it is the one to
// run (the general form needs 172 VGPRs, which would halve the occupancy of the fast form).
// TM filters per wave (4*TM per workgroup): 16 when that still fills the chip, else 8.
template <bool FAST, int TM>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const u64 *__restrict__ image, const u32 *__restrict__ wp,
                                                      const u64 *__restrict__ wl, const u32 *__restrict__ flag,
                                                      const uint8_t *__restrict__ keep,
                                                      const PwChannel *__restrict__ chan, u64 *__restrict__ acc, int M,
                                                      int Mpad, int Cpad, size_t S, int logN,
                                                      const DevMod *__restrict__ mods) {
  constexpr int PER = kPwChunk / 4, TF = 4 * TM;  // channels per wave per staging step, filters per workgroup
  if (((*flag & 1u) == 0) != FAST) return;
  constexpr bool fast = FAST;
  __shared__ u64 tile[2][kPwChunk][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = 1 << logN;
  const size_t col = (size_t)blockIdx.x * 64 + lane, cols = S * 4 * N;
  const int i = (int)(col & (N - 1));
  const int pl = (int)((((size_t)blockIdx.x * 64) >> logN) & 3), l = pl & 1;   // uniform over the tile
  const size_t sp = ((size_t)blockIdx.x * 64) >> (logN + 2);
  const int mbase = blockIdx.y * TF + wave * TM;
  u64 res[TM];
  const bool live = pl >= 2 || __ballot(keep[(sp << logN) + i] != 0) != 0;
  if (!live) {
#pragma unroll
    for (int j = 0; j < TM; ++j) res[j] = 0;
  } else {
    const DevMod q = mods[l];
    const u64 *__restrict__ base = image + ((sp * 4 + pl) << logN);
    const size_t row_step = S * 4 * N;
    // this thread's share of a chunk: channels wave, wave + 4, ... of the chunk, column `lane`
    u64 stage[PER];
    int stage_e[PER];
    auto issue = [&](int chunk) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const PwChannel pc = chan[chunk * kPwChunk + k * 4 + wave];
        stage_e[k] = pc.e;
        stage[k] = base[(size_t)pc.row * row_step + ((i - pc.e) & (N - 1))];
      }
    };
    auto commit = [&](int buf) {  // X^N = -1: coefficients that wrapped around come back negated
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const u64 v = stage[k];
        tile[buf][k * 4 + wave][lane] = i < stage_e[k] ? (v ? q.q - v : 0) : v;
      }
    };
    const int n_chunks = Cpad / kPwChunk;
    const u64 *__restrict__ wg = wl ? wl + (size_t)l * Cpad * Mpad + mbase : nullptr;
    u64 x0[TM], x1[TM], B0 = 0, B1 = 0;  // fast: the two 30-bit partial sums; general: 128-bit lo/hi
#pragma unroll
    for (int j = 0; j < TM; ++j) x0[j] = x1[j] = 0;
    issue(0);
    commit(0);
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
      const int buf = chunk & 1;
      if (chunk + 1 < n_chunks) issue(chunk + 1);
      if (mbase < M) {
        if (fast) {
          // weights of 4 channels at a time for this wave's filters: scalar loads issued as one burst
          // (scalar returns are unordered, so a wait drains them all: one wait per 4 channels)
          const PwWeights<TM> *__restrict__ wrow =
              reinterpret_cast<const PwWeights<TM> *>(wp + (size_t)chunk * kPwChunk * Mpad + mbase);
          const size_t wstep = (size_t)Mpad / TM;
          for (int k0 = 0; k0 < kPwChunk; k0 += 4) {
            PwWeights<TM> w4[4];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) w4[j4] = wrow[(size_t)(k0 + j4) * wstep];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
              const u64 u = tile[buf][k0 + j4][lane];
              const u32 a0 = (u32)u & 0x3fffffffu, a1 = (u32)(u >> 30);
              B0 += a0, B1 += a1;
#pragma unroll
              for (int j = 0; j < TM; j += 2)
                mad_pair_s(w4[j4].v[j], w4[j4].v[j + 1], a0, a1, x0[j], x1[j], x0[j + 1], x1[j + 1]);
            }
          }
        } else {
          // general weights: lifted 64-bit values, lazy 128-bit sums reduced every 8 terms (8 q^2 < q 2^64)
          for (int k = 0; k < kPwChunk; ++k) {
            const u64 u = tile[buf][k][lane];
            const u64 *__restrict__ wr = wg + (size_t)(chunk * kPwChunk + k) * Mpad;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              const u64 wj = wr[j], p = u * wj;
              x0[j] += p;
              x1[j] += mulhi64(u, wj) + (x0[j] < p);
            }
            if ((k & 7) == 7) {
#pragma unroll
              for (int j = 0; j < TM; ++j) x0[j] = reduce128_wide(x1[j], x0[j], q), x1[j] = 0;
            }
          }
        }
      }
      if (chunk + 1 < n_chunks) commit(buf ^ 1);
      __syncthreads();
    }
    if (fast) {
      // bias term 2^20*(B0 + B1*2^30) mod q
      const u64 blo = B0 + (B1 << 30), bhi = (B1 >> 34) + (blo < B0);
      const u64 bias = reduce128_wide((bhi << kPwBiasLog) | (blo >> (64 - kPwBiasLog)), blo << kPwBiasLog, q);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const u64 lo = x0[j] + (x1[j] << 30), hi = (x1[j] >> 34) + (lo < x0[j]);
        res[j] = sub_mod(reduce128_wide(hi, lo, q), bias, q.q);
      }
    } else {
#pragma unroll
      for (int j = 0; j < TM; ++j) res[j] = x0[j];  // reduced after the last group of 8
    }
  }
#pragma unroll
  for (int j = 0; j < TM; ++j)
    if (mbase + j < M) acc[(size_t)(mbase + j) * cols + col] = res[j];
}

// ------------------------------------------------------------------ the fast form on the matrix cores
// With |w| < 2^20 the product W * U mod q is an exact integer matrix product of small digits:
//   w = sum_a wd_a 256^a (2 or 3 balanced byte digits),  u = sum_j ud_j 256^j (8 balanced byte digits of the
//   column value, or of its negation for a wrapped coefficient: no mod-q negation is needed),
//   sum_ch w u = sum_d 256^d * sum_{a+j=d} (sum_ch wd_a ud_j),
// every inner sum an i8 x i8 -> i32 matrix product (v_mfma_i32_16x16x64_i8; |sum| < 3 * 4096 * 2^14 < 2^28).
// pw_expand_kernel writes the digits of U once per layer in fragment order, pw_mfma_kernel multiplies.
// Digit image of a live 64-column tile: [slot][k / 64][digit 8][column / 16][lane = ((k % 64) / 16) << 4 |
// column % 16][k % 16]: the 16 bytes of a lane are the B operand of one instruction, a 64-channel step of one
// tile is 32 KiB contiguous.  The byte order inside a lane only has to agree between the two operands.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr u64 kDigitBias = 0x0080808080808080ull;

struct PwFold {   // what conv_front_kernel left for this launch to fold (all null on the other paths)
  const u32 *wflag;        // one word of flag bits per weights workgroup
  int n_wflag;
  u32 *flag_out;           // the one flag word the product kernels read
  const uint8_t *anymap;   // [KG][Mpad128] bytes
  int KG, Mpad128;
};
__global__ __launch_bounds__(256) void pw_expand_kernel(const u64 *__restrict__ image,
                                                        const PwChannel *__restrict__ chan,
                                                        const int *__restrict__ live, uint4 *__restrict__ u8,
                                                        const u32 *__restrict__ flag, int KC, size_t S, int logN,
                                                        const u32 *__restrict__ nzmap, int M,
                                                        u32 *__restrict__ zero_filter, PwFold fold, int skip7) {
  // conv2DOneFilter :573-580: some filter all zero mod t?  One workgroup reads the map pw_weights_kernel left.
  if (nzmap && blockIdx.x == 0 && blockIdx.y == 0) {
    int bad = 0;
    for (int m = threadIdx.x; m < M; m += 256) bad |= nzmap[m] == 0;
    if (bad) *zero_filter = 1u;
  }
  if (fold.wflag) {
    // (the launch after conv_front_kernel: the form is not known yet, so the digits are always written)
    const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.x * gridDim.y;
    if (b == 0) {
      u32 bits = 0;
      for (int i = threadIdx.x; i < fold.n_wflag; i += 256) bits |= fold.wflag[i];
      const int b0 = __syncthreads_or((int)(bits & 1u)), b1 = __syncthreads_or((int)(bits & 2u));
      if (threadIdx.x == 0) *fold.flag_out = (b0 ? 1u : 0u) | (b1 ? 2u : 0u);
    }
    if (fold.anymap) {   // the verdict, filters dealt round robin to the workgroups; every writer stores the same 1
      for (unsigned m = b; m < (unsigned)M; m += nb) {
        int any = 0;
        for (int kg = threadIdx.x; kg < fold.KG; kg += 256) any |= fold.anymap[(size_t)kg * fold.Mpad128 + m];
        if (!__syncthreads_or(any) && threadIdx.x == 0) *zero_filter = 1u;
      }
    }
  } else if (*flag & 1u) {
    return;
  }
  const size_t c0 = (size_t)live[blockIdx.x] * 64;   // first column of the tile: (sp, c0/c1, limb, i)
  const int kc = blockIdx.y, N = 1 << logN;
  const int col = threadIdx.x >> 2, kg = threadIdx.x & 3;
  const int i = (int)(c0 & (N - 1)) + col;
  const int pl = (int)((c0 >> logN) & 3);
  const size_t sp = c0 >> (logN + 2);
  const u64 *__restrict__ base = image + ((sp * 4 + pl) << logN);
  const size_t row_step = S * 4 * N;
  u32 lo[16], hi[16];
  PwChannel pc[16];   // padding taps read row 0 against zero weights
#pragma unroll
  for (int k = 0; k < 16; ++k) pc[k] = chan[kc * 64 + kg * 16 + k];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    u64 v = base[(size_t)pc[k].row * row_step + ((i - pc[k].e) & (N - 1))];
    if (i < pc[k].e) v = 0 - v;  // X^N = -1: wrapped coefficients enter negated, as plain negative integers
    const u64 d = (v + kDigitBias) ^ kDigitBias;
    lo[k] = (u32)d, hi[k] = (u32)(d >> 32);
  }
  // (modulus below 2^54: the top digit of every value and of every negation is 0, the product kernel leaves it out)
  const int nplanes = (skip7 >> (pl & 1)) & 1 ? 7 : 8;
  // byte j of the 16 values -> 16 bytes: 4x4 byte transposes
  uint4 *__restrict__ out = u8 + (((size_t)blockIdx.x * KC + kc) * 8 * 4 + (col >> 4)) * 64 + ((kg << 4) | (col & 15));
  auto planes = [&](const u32 (&x)[16], int j0) {
    u32 o[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32 a = __builtin_amdgcn_perm(x[4 * g + 1], x[4 * g], 0x05010400u);
      const u32 b = __builtin_amdgcn_perm(x[4 * g + 1], x[4 * g], 0x07030602u);
      const u32 c = __builtin_amdgcn_perm(x[4 * g + 3], x[4 * g + 2], 0x05010400u);
      const u32 d = __builtin_amdgcn_perm(x[4 * g + 3], x[4 * g + 2], 0x07030602u);
      o[0][g] = __builtin_amdgcn_perm(c, a, 0x05040100u);
      o[1][g] = __builtin_amdgcn_perm(c, a, 0x07060302u);
      o[2][g] = __builtin_amdgcn_perm(d, b, 0x05040100u);
      o[3][g] = __builtin_amdgcn_perm(d, b, 0x07060302u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j0 + j < nplanes) out[(size_t)(j0 + j) * 4 * 64] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
  };
  planes(lo, 0);
  planes(hi, 4);
}

// A workgroup of 8 waves works on an item of (at most) 128 filters x 64 columns at a time (PwItems below); wave
// (wm, wn) owns half of the item's filters (4, 2 or 1 tiles of 16) and columns 16 wn .. +15.  Per 64-channel step the 32 KiB of column digits and 16/24 KiB of weight digits are
// copied global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: no staging registers; both images are
// already in fragment order, lane l reads unit l, conflict free) into one of two buffers while the products of
// earlier steps run: three buffers, two steps in flight, one raw barrier per step (two buffers and one step in
// flight when the weights need a third digit).  Ten i32 accumulator tiles per filter tile, one per power of
// 256, started at 2^28 so that the sums stay positive.
// (Giving each XCD whole column tiles, all of whose filter blocks then stream the same digits through one L2, measured
// within 3 % of the plain order: the digit image sits in the Infinity Cache either way.)
// Where the time goes, 256 -> 256 3x3 at 14x14 (36 steps, 88 us): with the copies switched off 73 us, with the
// products switched off 54 us (48 KiB per step and compute unit = 10 TB/s over the chip), with both 8 us; one step
// takes 2.0 us = 3,700 cycles at the 2.03-2.11 GHz the kernel holds, against ~3,400 for its 120 products at the
// rate v_mfma_i32_16x16x64_i8 sustains back to back (tools/mfma_i8_probe.hip; 1,920 by the nominal 16 cycles).
// The rare form inside the product kernel: some weight of the layer is 2^20 or more in magnitude, so the byte-digit
// product does not apply and the workgroup computes its 128 filters x 64 columns on the vector ALU from the lifted
// 64-bit weights, exactly as pw_gemm_kernel<false> does (lazy 128-bit sums reduced every 8 taps, canonical results).
// A branch here instead of a second launch that nearly always exits at once (4-5 us per layer).  Not tuned: wave w
// takes filters 16 w .. 16 w + 15, every lane one column, the rotated column value re-read per tap.
struct PwLift {   // where the lifted weights come from when no wl array was written: the raw filter tensor [M][C]
  const u64 *raw;
  int C;
  ModC t;
  u64 thr;
};
// What a paired item (below) needs of the modulus switch: q1^-1 mod q0 with its Shoup quotient, floor(q1 / 2) and
// that value modulo q0 (divide_and_round_q_last, the first step of the re-sharing tail).
struct PwSwitch {
  u64 inv_qlast, inv_qlast_sh, half1, half0;
};
// divide_and_round_q_last of one coefficient from its two residues: last = [a1 + q1/2]_q1 read as an integer
// (q1 < q0), result ((a0 - last + half0) q1^-1) mod q0, canonical -- the value the tail's switched_plus adds to the
// encryption of zero (ntt.hip), so the tail reads one word where it read two.
__device__ __forceinline__ u64 pw_switch_last(u64 a1, u64 q1, const PwSwitch &sw) { return csub(a1 + sw.half1, q1); }
__device__ __forceinline__ u64 pw_switch_finish(u64 a0_minus_last, u64 q0, const PwSwitch &sw) {
  const u64 y = shoup_mad_uniform(0, a0_minus_last + sw.half0, sw.inv_qlast, sw.inv_qlast_sh, 0 - q0);   // [0, 3 q0)
  return csub(csub(y, 2 * q0), q0);
}
__device__ __noinline__ bool pw_general_core(const u64 *__restrict__ image, const PwChannel *__restrict__ chan,
                                             const u64 *__restrict__ wl, int M, int Mpad, int Cpad, size_t S, int logN,
                                             const DevMod *__restrict__ mods, size_t c0, int mt, const PwLift &lift,
                                             u64 (&x0)[16]) {
  const int N = 1 << logN, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t col = c0 + lane;
  const int i = (int)(col & (N - 1)), pl = (int)((c0 >> logN) & 3), l = pl & 1;
  const size_t sp = c0 >> (logN + 2);
  const int mbase = mt * 128 + wave * 16;
  if (mbase >= M) return false;
  const DevMod q = mods[l];
  const u64 *__restrict__ base = image + ((sp * 4 + pl) << logN);
  const size_t row_step = S * 4 * N;
  const u64 *__restrict__ wg = wl + (size_t)l * Cpad * Mpad + mbase;
  const int nf = min(16, Mpad - mbase);
  u64 x1[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x0[j] = x1[j] = 0;
  for (int k = 0; k < Cpad; ++k) {
    const PwChannel pc = chan[k];
    const u64 v = base[(size_t)pc.row * row_step + ((i - pc.e) & (N - 1))];
    const u64 u = i < pc.e ? (v ? q.q - v : 0) : v;   // X^N = -1
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      u64 wj = 0;
      if (wl) {
        wj = j < nf ? wg[(size_t)k * Mpad + j] : 0;
      } else if (mbase + j < M && k < lift.C) {   // multiply_plain's fast lift of the weight, as pw_weights_body writes it
        const u64 v = red_t(lift.raw[(size_t)(mbase + j) * lift.C + k], lift.t);
        wj = v >= lift.thr ? v + (q.q - lift.t.q) : v;
      }
      const u64 p = u * wj;
      x0[j] += p;
      x1[j] += mulhi64(u, wj) + (x0[j] < p);
    }
    if ((k & 7) == 7) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x0[j] = reduce128_wide(x1[j], x0[j], q), x1[j] = 0;
    }
  }
  return true;
}
__device__ __forceinline__ void pw_general_tile(const u64 *__restrict__ image, const PwChannel *__restrict__ chan,
                                                const u64 *__restrict__ wl, u64 *__restrict__ acc, int M, int Mpad,
                                                int Cpad, size_t S, int logN, const DevMod *__restrict__ mods, size_t c0,
                                                int mt, const PwLift &lift) {
  const size_t cols = S * 4 * (size_t(1) << logN), col = c0 + (threadIdx.x & 63);
  const int mbase = mt * 128 + (int)(threadIdx.x >> 6) * 16;
  u64 x0[16];
  if (!pw_general_core(image, chan, wl, M, Mpad, Cpad, S, logN, mods, c0, mt, lift, x0)) return;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (mbase + j < M) acc[(size_t)(mbase + j) * cols + col] = x0[j];
}
// The same for a paired item: both limbs of the tile at c0 (the limb-0 tile; its partner sits N columns on), the
// mod-switched value written in the one-limb layout [M][S][2][N].
__device__ __forceinline__ void pw_general_pair(const u64 *__restrict__ image, const PwChannel *__restrict__ chan,
                                                const u64 *__restrict__ wl, u64 *__restrict__ acc, int M, int Mpad,
                                                int Cpad, size_t S, int logN, const DevMod *__restrict__ mods, size_t c0,
                                                int mt, const PwLift &lift, const PwSwitch &sw) {
  const size_t N = size_t(1) << logN, cols = S * 2 * N;
  const size_t sp = c0 >> (logN + 2), k = (c0 >> (logN + 1)) & 1, i = (c0 & (N - 1)) + (threadIdx.x & 63);
  const size_t col = ((sp * 2 + k) << logN) + i;
  const int mbase = mt * 128 + (int)(threadIdx.x >> 6) * 16;
  u64 a1[16], a0[16];
  if (!pw_general_core(image, chan, wl, M, Mpad, Cpad, S, logN, mods, c0 + N, mt, lift, a1)) return;
  (void)pw_general_core(image, chan, wl, M, Mpad, Cpad, S, logN, mods, c0, mt, lift, a0);
  const u64 q0 = mods[0].q, q1 = mods[1].q;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (mbase + j < M) {
      const u64 last = pw_switch_last(a1[j], q1, sw);
      acc[(size_t)(mbase + j) * cols + col] = pw_switch_finish(a0[j] >= last ? a0[j] - last : a0[j] + q0 - last, q0, sw);
    }
}

// The work of one launch as a list of items, item = one live 64-column tile x a block of filters.  The launch has one
// workgroup per compute unit (at most), workgroup b takes items b, b + G, b + 2G, ...: the copies of an item's first
// steps are issued while the previous item's last products and its epilogue run, and no workgroup is started or retired
// between items (against one workgroup per item: 1024 -> 256 1x1 at 14x14 58 -> 45 us, 256 -> 256 3x3 114 -> 88 us,
// 512 -> 2048 1x1 at 7x7 176 -> 161 us; layers of few steps and many items, 256 -> 1024 1x1, stay where they were).
// Main items hold F filters (128; 64 or 32 when the layer has no more, so that both wave rows work); the items left
// over after the last full round of G are cut into s = 2 or 4 pieces of F / s filters when that many more workgroups
// find a free compute unit, which shortens the last round instead of leaving most of the chip idle in it.
struct PwItems {
  int n_main;    // items of F filters: index -> (slot = index / MTm, filter block index % MTm)
  int n_items;   // n_main + (left over) << s_log
  int MTm;       // filter blocks of F per column tile
  int F_log;     // 7, 6 or 5
  int s_log;     // 0, 1 or 2
  int skip7;     // bit l: modulus l is below 2^54, so the eighth column digit is 0 throughout and its products are left out
  // An item owns BOTH limbs of its 64 columns (live tiles are listed in pairs: limb 1, then limb 0): the steps of limb
  // 1 first, its residues mod-switched against limb 0's on chip, one word written per coefficient (PwSwitch).  The
  // column-tile index of an item then counts pairs.
  int paired;
};
static PwItems pw_items(int n_live, int M, int n_cu) {
  PwItems it;
  it.F_log = M <= 32 ? 5 : M <= 64 ? 6 : 7;
  it.MTm = (M + (1 << it.F_log) - 1) >> it.F_log;
  const int items = n_live * it.MTm;
  const int rem = items < n_cu ? items : items % n_cu;
  it.n_main = items - rem;
  it.s_log = 0;
  while (it.s_log < 2 && it.F_log - (it.s_log + 1) >= 5 && (rem << (it.s_log + 1)) <= n_cu) ++it.s_log;
  it.n_items = it.n_main + (rem << it.s_log);
  it.skip7 = 0;
  it.paired = 0;
  return it;
}

// One output of the product from its ten biased digit sums: value = sum_d (p[d] - 2^28) 256^d, signed, below 2^100 in
// magnitude.  The biased digits are assembled into 128 bits, k*q - bias is added (k*q in [2^101, 2^102)) to land on a
// non-negative representative below 2^103, then reduced: two folds of q = 2^b - c where the modulus has that shape,
// else Barrett.  Canonical result.
// (written on 32-bit words with v_mad_u64_u32 and funnel shifts; hipcc's 64-bit shifts by a register amount and its
// 128-bit adds made the epilogue a tenth of the kernel)
struct PwRed {
  const DevMod &q;
  u32 o0, o1, o2, o3, fc, fs, fmask;
  bool folds;
  __device__ __forceinline__ explicit PwRed(const DevMod &q_) : q(q_) {
    const u32 qbits = 64 - (u32)__builtin_clzll(q.q), sh = 102 - qbits;   // q << sh in [2^101, 2^102)
    const u64 kq_lo = q.q << sh, kq_hi = q.q >> (64 - sh);
    const u64 off_lo = kq_lo - 0x1010101010000000ull, off_hi = kq_hi - 0x1010101010ull - (kq_lo < 0x1010101010000000ull);
    const u32 fbits = q.fold_shift + 32;
    fc = q.fold_c, fs = q.fold_shift, fmask = q.fold_mask;
    // fold bounds: x < 2^103 -> x1 < 2^b + 2^(103-b) c (bits1 bits) -> x2 < 2^b + (x1 >> b) c (bits2 bits, must fit a word)
    folds = fc != 0 && fbits >= 40 && fbits <= 62;
    if (folds) {
      const u32 cb = 32 - (u32)__builtin_clz(fc);
      const u32 bits1 = max(fbits, 103 - fbits + cb) + 1;
      const u32 bits2 = max(fbits, bits1 - fbits + cb) + 1;
      folds = bits1 <= 64 + fbits && bits2 <= 64;
    }
    o0 = lo32(off_lo), o1 = hi32(off_lo), o2 = lo32(off_hi), o3 = hi32(off_hi);
  }
  __device__ __forceinline__ u64 operator()(const u32 (&p)[10]) const {
    const u64 lo = mad32(p[3], 1u << 24, mad32(p[2], 1u << 16, mad32(p[1], 1u << 8, p[0])));
    const u64 mid = mad32(p[7], 1u << 24, mad32(p[6], 1u << 16, mad32(p[5], 1u << 8, p[4])));
    const u64 top = mad32(p[9], 1u << 8, p[8]);
    // lo + mid 2^32 + top 2^64 + offset, word by word
    const u64 a0 = (u64)lo32(lo) + o0;
    const u64 a1 = (u64)hi32(lo) + lo32(mid) + o1 + hi32(a0);
    const u64 a2 = (u64)hi32(mid) + lo32(top) + o2 + hi32(a1);
    u32 w0 = lo32(a0), w1 = lo32(a1), w2 = lo32(a2);
    const u32 w3 = hi32(top) + o3 + hi32(a2);
    if (folds) {
      fold_words(w0, w1, w2, w3, fc, fs, fmask);
      fold_words(w0, w1, w2, 0, fc, fs, fmask);
      return csub(fold_mod(make64(w0, w1), fc, fs, fmask), q.q);
    }
    return reduce128_wide(make64(w2, w3), make64(w0, w1), q);
  }
};

constexpr int kPwLdsBytes = 3 * (2048 + 1024) * 16;   // 144 KiB: three buffers of 2 weight digits (two of 3 fit too)
// STAMPS (diagnostic instantiation, MMH_PW_STAMPS=<file>): lane 0 of every wave of workgroup 0 records the shader clock
// at five points of each of its first kPwStampSteps steps -- 0 top of the step, 1 past the wait and the barrier (this
// step's digits have landed for every wave), 2 the next step's copies issued, 3 the fragments read (first product can
// issue), 4 the last product issued -- into stamps[wave][step][5].  Reading the clock drains the scalar counter, so the
// stamped kernel runs some per cent slower than the plain one; the table shows where a step's cycles go.
constexpr int kPwStampSteps = 96;
template <bool STAMPS>
__global__ __launch_bounds__(512) void pw_mfma_kernel(const uint4 *__restrict__ u8, const uint4 *__restrict__ w8,
                                                      const u32 *__restrict__ flag, const int *__restrict__ live,
                                                      u64 *__restrict__ acc, int M, int KC, int n_live, size_t S,
                                                      int logN, const DevMod *__restrict__ mods,
                                                      const u64 *__restrict__ image, const PwChannel *__restrict__ chan,
                                                      const u64 *__restrict__ wl, int Mpad, int Cpad, PwItems items, PwLift lift,
                                                      PwSwitch sw, u64 *__restrict__ stamps) {
  extern __shared__ uint4 pw_lds[];
  int stamp_step = 0;
  auto stamp = [&](int k) {
    if (STAMPS && blockIdx.x == 0 && stamp_step < kPwStampSteps) {
      const u64 t = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) stamps[((threadIdx.x >> 6) * kPwStampSteps + stamp_step) * 5 + k] = t;
    }
  };
  const u32 fl = *flag;
  const int G = (int)gridDim.x;
  if (fl & 1u) {                        // weights of 2^20 and more: the lifted form (rare), 128 filters per item
    const int MT = (M + 127) >> 7;
    for (int idx = blockIdx.x; idx < n_live * MT; idx += G)
      if (items.paired)   // n_live counts pairs; the pair's second tile is the limb-0 one
        pw_general_pair(image, chan, wl, acc, M, Mpad, Cpad, S, logN, mods, (size_t)live[2 * (idx / MT) + 1] * 64,
                        idx % MT, lift, sw);
      else
        pw_general_tile(image, chan, wl, acc, M, Mpad, Cpad, S, logN, mods, (size_t)live[idx / MT] * 64, idx % MT, lift);
    return;
  }
  const int nA = (fl & 2u) ? 3 : 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int N = 1 << logN;
  const size_t cols = items.paired ? S * 2 * N : S * 4 * N;
  const int KCB = items.paired ? 2 * KC : KC;   // steps of an item's column digits (paired: limb 1, then limb 0)
  const int stride = 2048 + 512 * nA, nbuf = nA == 2 ? 3 : 2;   // uint4 per buffer: column digits, weight digits
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) void *)pw_lds;
  // item index -> column-tile slot and first filter; false past the layer's last filter (pieces of the last block)
  const int F = 1 << items.F_log, f_sub = F >> items.s_log;
  auto decode = [&](int it, int &slot, int &fb, int &f) {
    int idx = it, part = 0;
    f = F;
    if (it >= items.n_main) {
      const int j = it - items.n_main;
      idx = items.n_main + (j >> items.s_log), part = j & ((1 << items.s_log) - 1), f = f_sub;
    }
    slot = idx / items.MTm;
    fb = (idx % items.MTm) * F + part * f;
    return fb < M;
  };
  auto next_item = [&](int it) {
    int slot, fb, f;
    while (it < items.n_items && !decode(it, slot, fb, f)) it += G;
    return it;
  };
  // One 1 KiB copy global -> LDS: lane l's 16 bytes land at dst + 16 l.  Inline asm so that hipcc neither counts
  // the copy nor drains it before the next LDS read; the waits below count them.
  auto glds16 = [&](const uint4 *src, u32 dst_unit) {
    u32 keep;
    const u32 dst = __builtin_amdgcn_readfirstlane(lds0 + dst_unit * 1024);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
  };
  // The copies run ahead of the products over item boundaries: (s_it, s_kc) is the next step to copy, `staged` counts
  // the steps copied so far and `done` the steps multiplied (both by this workgroup; buffer = count mod nbuf).
  int s_it = next_item((int)blockIdx.x), s_kc = 0, s_buf = 0, staged = 0, done = 0;
  const uint4 *gB = u8 + lane, *gA = w8 + lane;
  auto stager_item = [&]() {
    int slot, fb, f;
    if (s_it < items.n_items && decode(s_it, slot, fb, f)) {
      gB = u8 + (size_t)slot * KCB * 2048 + lane;   // (a pair's two tiles are adjacent slots of the digit image)
      gA = w8 + (size_t)(fb >> 7) * KC * 1536 + lane;
    }
  };
  stager_item();
  auto stage_next = [&]() {   // this wave's share of one step: 4 KiB of column digits, 1 KiB per weight digit
    if (s_it >= items.n_items) return;
    const u32 unit0 = (u32)(s_buf * stride) / 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) glds16(gB + (size_t)s_kc * 2048 + (wave * 4 + r) * 64, unit0 + wave * 4 + r);
    const int a_kc = s_kc >= KC ? s_kc - KC : s_kc;   // the weights are the same for both limbs
#pragma unroll
    for (int a = 0; a < 3; ++a)
      if (a < nA) glds16(gA + (size_t)a_kc * 1536 + (a * 8 + wave) * 64, unit0 + 32 + a * 8 + wave);
    ++staged;
    s_buf = s_buf + 1 == nbuf ? 0 : s_buf + 1;
    if (++s_kc == KCB) {
      s_kc = 0, s_it = next_item(s_it + G);
      stager_item();
    }
  };
  for (int i = 0; i + 1 < nbuf; ++i) stage_next();
  int cur = 0;                     // buffer of the step being multiplied
  for (int c_it = next_item((int)blockIdx.x); c_it < items.n_items; c_it = next_item(c_it + G)) {
    int slot, fb, f;
    decode(c_it, slot, fb, f);
    const int tile = live[items.paired ? 2 * slot + 1 : slot];     // paired: the limb-0 tile, which names the output
    const int msn = f >> 5;                                       // 16-filter tiles per wave: 4, 2 or 1
    const int wfb = fb + wm * (f >> 1);                           // this wave's first filter
    const int u0 = __builtin_amdgcn_readfirstlane(((fb & 127) >> 4) + wm * msn);   // ... its unit in the 128-filter block
    // tiles of this wave that hold a filter at all (wave-uniform; fewer only in the layer's last block)
    const int ms_used = __builtin_amdgcn_readfirstlane(min(msn, max(0, (M - wfb + 15) / 16)));
    int nj = (items.skip7 >> (items.paired ? 1 : (int)(((size_t)tile * 64 >> logN) & 1))) & 1 ? 7 : 8;   // column digits with products
    v4i sum[4][10];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int d = 0; d < 10; ++d) sum[ms][d] = v4i{1 << 28, 1 << 28, 1 << 28, 1 << 28};
    for (int kc = 0; kc < KCB; ++kc) {
      if (items.paired && kc == KC) {
        // Limb 1 is complete: last = [a1 + q1/2]_q1 of every output, and limb 0's sums start at -last, digit by
        // digit (they are plain integers), so that the second half accumulates a0 - last: nothing is parked.
        const PwRed red(mods[1]);
        const u64 q1 = mods[1].q;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (ms >= ms_used) continue;
            u32 p[10];
#pragma unroll
            for (int d = 0; d < 10; ++d) p[d] = (u32)sum[ms][d][r];
            const u64 last = pw_switch_last(red(p), q1, sw);
#pragma unroll
            for (int d = 0; d < 10; ++d)
              sum[ms][d][r] = (1 << 28) - (d < 8 ? (int)((last >> (8 * d)) & 255) : 0);
          }
        nj = items.skip7 & 1 ? 7 : 8;
      }
      stamp(0);
      // This step has landed (this wave's copies; the barrier covers the others'), a later step may stay in flight:
      // when one was copied after it, at least 4 + nA operations are younger than this step's copies (an epilogue's
      // stores may be too, which only makes the wait longer).  Past the barrier every wave is done with the buffer of
      // the step before, which the new copies overwrite.  A raw barrier: __syncthreads() would drain the copies.
      if (nbuf == 3 && staged - done >= 2)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // 4 + 2 copies per wave and step
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stamp(1);
      stage_next();   // (issuing the pieces between the groups of products instead measured the same)
      stamp(2);
      const uint4 *ldsB = pw_lds + cur * stride, *ldsA = ldsB + 2048;
      cur = cur + 1 == nbuf ? 0 : cur + 1;
      ++done;
      v4i b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint4 t = ldsB[(j * 4 + wn) * 64 + lane];
        b[j] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
      }
      if (STAMPS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(3);
      }
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        if (ms >= ms_used) break;   // filter tiles past the wave's share or the layer's last filter: no products
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (a < nA) {
            const uint4 t = ldsA[(a * 8 + u0 + ms) * 64 + lane];
            const v4i av = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < 7 || nj == 8)
                sum[ms][a + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[j], sum[ms][a + j], 0, 0, 0);
          }
        }
      }
      stamp(4);
      if (STAMPS) ++stamp_step;
    }
    const size_t c0 = (size_t)tile * 64;
    const int mrow = wfb + (lane >> 4) * 4;   // + 16*ms + r
    const DevMod &q = mods[(int)((c0 >> logN) & 1)];
    const PwRed red(q);
    // paired: the limb-0 tile's (window, component, coefficient) in the one-limb layout [S][2][N]
    const size_t col = (items.paired ? (((c0 >> (logN + 2)) * 2 + ((c0 >> (logN + 1)) & 1)) << logN) + (c0 & (size_t)(N - 1)) : c0) +
                       wn * 16 + (lane & 15);
    u64 *__restrict__ out = acc + (size_t)mrow * cols + col;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (ms >= ms_used) continue;
        u32 p[10];
#pragma unroll
        for (int d = 0; d < 10; ++d) p[d] = (u32)sum[ms][d][r];
        u64 v = red(p);
        if (items.paired) v = pw_switch_finish(v, q.q, sw);   // v was (a0 - last) mod q0
        if (mrow + 16 * ms + r < M) out[(size_t)(16 * ms + r) * cols] = v;
      }
  }
}

// ------------------------------------------------------------------ layers of at most 32 filters: no digit image
// DenseNet121's 58 growth layers (128 -> 32, 3x3) and SqueezeNet's squeeze layers have one block of filters, so the
// digit image of the rotated columns would be written once and read once: pw_expand_kernel's pass and the image's two
// trips through memory bought nothing.  Here the product's own workgroup builds the digits of a 64-tap step in LDS
// (the expand kernel's gather and byte transposes, into the same fragment order) while the previous step's products
// run: one workgroup of eight waves per live 64-column tile; every wave gathers 8 of the step's 64 taps for the 64
// columns, wave (wm, wn) multiplies filter tile wm by columns 16 wn .. + 15; two LDS buffers of column digits (32 KiB)
// and weight digits (2 or 3 planes of 2 KiB), the loads of steps k + 1 and k + 2 in registers during the products of
// step k.  Same digits, same i32 sums, same PwRed epilogue as pw_mfma_kernel: the same bits.
// The launch also does what the digit launch did for a call from raw tensors: folds the weight pass's flag words and
// evaluates the zero-filter rule (workgroup 0).
constexpr int kPwNarrowLds = 2 * (2048 + 3 * 128) * 16;   // 76 KiB
// The pk-encryptions of zero of a drawing call (ntt_zero_unit12: u drawn, transformed, multiplied by pk_j, transformed
// back) depend on nothing else in the call and only the tail reads them.  In the front launch they were its longest
// workgroups by far -- two transforms in a row, ~16 us, beside ~10 us of image and weight work.  Here they RIDE in the
// product launch instead: its first `wgs` workgroups make one ciphertext's two components each (lanes 0-255 component
// 0, lanes 256-511 component 1, each on its own half of the LDS allocation) beside products that take 20-40 us.
// (Tried and dropped: the same in the digit image's launch and in pw_mfma_kernel's -- those launches are shorter than
// the riders, or their persistent workgroups already take every compute unit: EXPERIMENTS.md.)
struct ZeroRide {
  unsigned wgs;            // output ciphertexts = riding workgroups (0: nothing rides)
  const u64 *pk;           // pk at the last level [2][N], NTT form
  u64 *zero;               // [wgs][2][N]
  const ulonglong2 *tables;
  ChaChaKey key;           // the ternary stream of the draw
  u64 sid, first_tile;
};
__global__ __launch_bounds__(512, 2) void pw_narrow_kernel(const u64 *__restrict__ image, const PwChannel *__restrict__ chan,
                                                        const int *__restrict__ live, const uint4 *__restrict__ w8,
                                                        const u32 *__restrict__ flag, u64 *__restrict__ acc, int M,
                                                        int KC, size_t S, int logN, const DevMod *__restrict__ mods,
                                                        const u64 *__restrict__ wl, int Mpad, int Cpad, PwLift lift,
                                                        PwFold fold, u32 *__restrict__ zero_filter, int skip7,
                                                        ZeroRide zr) {
  extern __shared__ uint4 pwn_lds[];
  __shared__ u32 s_flag;
  if (blockIdx.x < zr.wgs) {   // (uniform over the workgroup)
    const unsigned j = threadIdx.x >> 8;
    NttSource usrc{};
    usrc.key = zr.key, usrc.sid = zr.sid, usrc.first_tile = zr.first_tile;
    ntt_zero_unit12<true>(blockIdx.x, mods[0], zr.tables, usrc, zr.pk + ((size_t)j << 12),
                          zr.zero + (((size_t)blockIdx.x * 2 + j) << 12),
                          reinterpret_cast<u64 *>(pwn_lds) + j * (lds_bytes<12>() / 8), (int)(threadIdx.x & 255));
    return;
  }
  const unsigned bid = blockIdx.x - zr.wgs;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (fold.wflag) {   // (a call from raw tensors: the weight pass left one word of flag bits per workgroup)
    u32 bits = 0;
    for (int i = tid; i < fold.n_wflag; i += 512) bits |= fold.wflag[i];
    const int b0 = __syncthreads_or((int)(bits & 1u)), b1 = __syncthreads_or((int)(bits & 2u));
    if (tid == 0) {
      s_flag = (b0 ? 1u : 0u) | (b1 ? 2u : 0u);
      if (bid == 0) *fold.flag_out = s_flag;
    }
    if (fold.anymap && bid == 0)   // conv2DOneFilter :573-580: some filter all zero mod t?
      for (int m = 0; m < M; ++m) {
        int any = 0;
        for (int kg = tid; kg < fold.KG; kg += 512) any |= fold.anymap[(size_t)kg * fold.Mpad128 + m];
        if (!__syncthreads_or(any) && tid == 0) *zero_filter = 1u;
      }
  } else if (tid == 0) {
    s_flag = *flag;
  }
  __syncthreads();
  const u32 fl = s_flag;
  const size_t c0 = (size_t)live[bid] * 64;   // first column of the tile: (window, c0/c1, limb, i)
  const int N = 1 << logN;
  const size_t cols = S * 4 * N;
  const int pl = (int)((c0 >> logN) & 3);
  if (fl & 1u) {
    // Weights of 2^20 and more (rare): the lifted form on the vector ALU as in pw_general_core, but four filters per
    // wave (8 waves x 4 = the 32 filters at most) so that this branch does not set the kernel's register count.
    const int mbase = wave * 4;
    if (mbase >= M) return;
    const DevMod q = mods[pl & 1];
    const size_t ocol = c0 + lane;
    const int ii = (int)(ocol & (size_t)(N - 1));
    const u64 *__restrict__ gbase = image + (((c0 >> (logN + 2)) * 4 + pl) << logN);
    const size_t rstep = S * 4 * N;
    const u64 *__restrict__ wg = wl ? wl + (size_t)(pl & 1) * Cpad * Mpad + mbase : nullptr;
    u64 x0[4] = {0, 0, 0, 0}, x1[4] = {0, 0, 0, 0};
    for (int k = 0; k < Cpad; ++k) {
      const PwChannel pc = chan[k];
      const u64 v = gbase[(size_t)pc.row * rstep + ((ii - pc.e) & (N - 1))];
      const u64 u = ii < pc.e ? (v ? q.q - v : 0) : v;   // X^N = -1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u64 wj = 0;
        if (wg) {
          wj = mbase + j < Mpad ? wg[(size_t)k * Mpad + j] : 0;
        } else if (mbase + j < M && k < lift.C) {   // multiply_plain's fast lift of the weight, as pw_weights_body writes it
          const u64 r = red_t(lift.raw[(size_t)(mbase + j) * lift.C + k], lift.t);
          wj = r >= lift.thr ? r + (q.q - lift.t.q) : r;
        }
        const u64 p = u * wj;
        x0[j] += p;
        x1[j] += mulhi64(u, wj) + (x0[j] < p);
      }
      if ((k & 7) == 7) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x0[j] = reduce128_wide(x1[j], x0[j], q), x1[j] = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (mbase + j < M) acc[(size_t)(mbase + j) * cols + ocol] = x0[j];
    return;
  }
  const int nA = (fl & 2u) ? 3 : 2;
  const int nj = (skip7 >> (pl & 1)) & 1 ? 7 : 8;
  constexpr int kStride = 2048 + 3 * 128;   // uint4 per buffer
  // gather role: wave w takes taps 8 w .. + 7 of the step (its PwChannel entries are wave-uniform: scalar loads), lane =
  // column, so a tap's 64 rotated coefficients are one 512-byte run (pw_expand_kernel splits a wave over four rows)
  const int i = (int)(c0 & (size_t)(N - 1)) + lane;
  const u64 *__restrict__ base = image + (((c0 >> (logN + 2)) * 4 + pl) << logN);
  const size_t row_step = S * 4 * N;
  const int n_wunits = nA * 2;   // 1-KiB units of weight digits per step, unit u = (digit u >> 1, filter tile u & 1)
  struct Staged {   // one step's operands on their way from memory to LDS: nothing here reads a loaded value, so the
    u64 v[8];       // loads stay in flight until store_step
    u32 neg;        // bit k: tap k's coefficient wrapped around X^N = -1 and enters negated
    v4u w;          // (a native vector: an asm operand)
  };
  // The loads are issued from inline asm and counted by hand (kLoadsPerStep per lane and step, in order): hipcc's own
  // wait insertion loses track of them across the loop's back edge and drains every outstanding load before each
  // store_step, which leaves one step of lead instead of two (1.65 us per step measured, the latency of a round trip).
  constexpr int kLoadsPerStep = 9;
  // (the step's eight PwChannel entries are fetched a step ahead of the loads that need them -- scalar loads, whose
  // round trip would otherwise sit in front of every step's products)
  struct Taps {
    PwChannel pc[8];
  };
  auto fetch_taps = [&](int kc, Taps &tp) {
    if (kc >= KC) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) tp.pc[k] = chan[kc * 64 + wave * 8 + k];
  };
  auto load_step = [&](int kc, Staged &st, const Taps &tp) {
    u32 neg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const PwChannel pc = tp.pc[k];
      const u64 *addr = base + (size_t)pc.row * row_step + ((i - pc.e) & (N - 1));
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(st.v[k]) : "v"(addr) : "memory");
      neg |= (i < pc.e ? 1u : 0u) << k;
    }
    st.neg = neg;
    const int u = wave < n_wunits ? wave : 0;   // (every wave loads, so that the value lives in registers; two more store)
    const uint4 *waddr = w8 + (size_t)kc * 1536 + ((u >> 1) * 8 + (u & 1)) * 64 + lane;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st.w) : "v"(waddr) : "memory");
  };
  // st's loads have landed once at most `newer` younger loads are outstanding; the "+v" ties keep every use of st behind
  // the wait
  auto landed = [&](Staged &st, bool newer_step_in_flight) {
    if (newer_step_in_flight)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerStep) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(st.v[k]));
    asm volatile("" : "+v"(st.w));
  };
  // unit (16 bytes) of digit plane j for this lane's column and this wave's tap group; the wave fills one half of it
  u64 *const out_half = reinterpret_cast<u64 *>(pwn_lds + (lane >> 4) * 64 + (((wave >> 1) << 4) | (lane & 15))) + (wave & 1);
  auto store_step = [&](int buf, const Staged &st) {
    u32 lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u64 x = (st.neg >> k) & 1u ? 0 - st.v[k] : st.v[k];   // plain negative integers: no mod-q negation
      const u64 d = (x + kDigitBias) ^ kDigitBias;
      lo[k] = (u32)d, hi[k] = (u32)(d >> 32);
    }
    u64 *out = out_half + (size_t)buf * kStride * 2;
    auto planes = [&](const u32 (&x)[8], int j0) {   // byte j of the 8 values -> 8 bytes: two 4x4 byte transposes
      u32 o[4][2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const u32 a = __builtin_amdgcn_perm(x[4 * g + 1], x[4 * g], 0x05010400u);
        const u32 b = __builtin_amdgcn_perm(x[4 * g + 1], x[4 * g], 0x07030602u);
        const u32 c = __builtin_amdgcn_perm(x[4 * g + 3], x[4 * g + 2], 0x05010400u);
        const u32 d = __builtin_amdgcn_perm(x[4 * g + 3], x[4 * g + 2], 0x07030602u);
        o[0][g] = __builtin_amdgcn_perm(c, a, 0x05040100u);
        o[1][g] = __builtin_amdgcn_perm(c, a, 0x07060302u);
        o[2][g] = __builtin_amdgcn_perm(d, b, 0x05040100u);
        o[3][g] = __builtin_amdgcn_perm(d, b, 0x07060302u);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j0 + j < nj) out[(size_t)(j0 + j) * 4 * 64 * 2] = make64(o[j][0], o[j][1]);
    };
    planes(lo, 0);
    planes(hi, 4);
    if (wave < n_wunits) pwn_lds[buf * kStride + 2048 + wave * 64 + lane] = make_uint4(st.w.x, st.w.y, st.w.z, st.w.w);
  };
  // product role: wave (wm, wn) owns filter tile wm and columns 16 wn .. + 15
  const int wm = wave >> 2, wn = wave & 3;
  const bool has_tile = wm * 16 < M;
  v4i sum[10];
#pragma unroll
  for (int d = 0; d < 10; ++d) sum[d] = v4i{1 << 28, 1 << 28, 1 << 28, 1 << 28};
  auto products = [&](int buf) {
    if (!has_tile) return;
    const uint4 *ldsB = pwn_lds + buf * kStride, *ldsA = ldsB + 2048;
    v4i b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint4 t = ldsB[(j * 4 + wn) * 64 + lane];
      b[j] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (a < nA) {
        const uint4 t = ldsA[(a * 2 + wm) * 64 + lane];
        const v4i av = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < 7 || nj == 8) sum[a + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[j], sum[a + j], 0, 0, 0);
      }
    }
  };
  // Two steps of loads in flight: step k + 2 is requested before the products of step k, step k + 1 (requested an
  // iteration earlier) is turned into digits behind them.  Staging slots alternate, so the loop runs in pairs.
  // (a raw barrier behind a wait for this wave's LDS traffic only: __syncthreads() would also drain the global loads
  // that are meant to stay in flight across it)
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  Staged s0, s1;
  Taps t0, t1;
  fetch_taps(0, t0);
  fetch_taps(1, t1);
  load_step(0, s0, t0);
  fetch_taps(2, t0);
  if (KC > 1) load_step(1, s1, t1);
  fetch_taps(3, t1);
  landed(s0, KC > 1);
  store_step(0, s0);
  lds_barrier();
  for (int kc = 0; kc < KC; kc += 2) {
    if (kc + 2 < KC) load_step(kc + 2, s0, t0);
    fetch_taps(kc + 4, t0);
    products(0);
    if (kc + 1 < KC) {   // the other buffer: its readers finished at the last barrier
      landed(s1, kc + 2 < KC);
      store_step(1, s1);
    }
    lds_barrier();
    if (kc + 1 >= KC) break;
    if (kc + 3 < KC) load_step(kc + 3, s1, t1);
    fetch_taps(kc + 5, t1);
    products(1);
    if (kc + 2 < KC) {
      landed(s0, kc + 3 < KC);
      store_step(0, s0);
    }
    lds_barrier();
  }
  if (!has_tile) return;
  const size_t ocol = c0 + wn * 16 + (lane & 15);
  const int mrow = wm * 16 + (lane >> 4) * 4;   // + r
  const PwRed red(mods[pl & 1]);
  u64 *__restrict__ out = acc + (size_t)mrow * cols + ocol;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    u32 p[10];
#pragma unroll
    for (int d = 0; d < 10; ++d) p[d] = (u32)sum[d][r];
    const u64 val = red(p);
    if (mrow + r < M) out[(size_t)r * cols] = val;
  }
}
// MMH_PW_NARROW=0 keeps layers of at most 32 filters on the digit image, =1 sends all of them here (A/B measurements,
// tests of both forms).  By itself the library takes this kernel when its one workgroup per tile fills the chip or
// the tap loop is long enough to hide a workgroup's start: with fewer tiles than compute units and at most four steps
// the two-launch form's full occupancy wins (SqueezeNet's 128 / 256 -> 32 1x1 at 27x27, 150 tiles: 59 against 65 us
// per layer call; 64 / 128 -> 16 at 56x56, 250 tiles: 52 -> 48 and 60 -> 55 us the other way).
static bool pw_narrow_ok(const mmh_conv_plan *p) {
  const char *e = getenv("MMH_PW_NARROW");
  if (e && e[0] == '0') return false;
  if (p->meta.n_filters > 32) return false;
  if (e && e[0] == '1') return true;
  return (int)p->pw_live_tiles >= p->ctx->n_cu || p->pw_taps_padded >= 8 * 64;
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

struct PwSizes {
  int M, Mpad, Mpad128, C, Cpad, Cpad64, wc;
  size_t w8_bytes() const { return (size_t)Mpad128 * Cpad64 * 3; }
};
static bool pw_use_mfma() {  // MMH_CONV_PW_MFMA=0 keeps the fast form on the vector ALU (A/B measurements)
  const char *e = getenv("MMH_CONV_PW_MFMA");
  return !(e && e[0] == '0');
}
// (the product's epilogue needs moduli of at least 40 bits for its k*q offset; SEAL's are 49 and 60)
static bool pw_mfma_enabled(const mmh_ctx *ctx) {
  return pw_use_mfma() && (ctx->q[0] >> 39) != 0 && (ctx->q[1] >> 39) != 0;
}
// The product can hand the tail the mod-switched accumulator (paired items) when the last modulus is the smaller one
// (its residue is then its own representative modulo q0, as ReshareCoeff::switched_plus assumes for qlast_small).
// MMH_PW_PAIRED=0 keeps the two-limb accumulator (A/B measurements, tests of both forms).
static bool pw_switched_ok(const mmh_ctx *ctx) {
  const char *e = getenv("MMH_PW_PAIRED");
  return !(e && e[0] == '0') && pw_mfma_enabled(ctx) && ctx->L == 2 && ctx->q[1] < ctx->q[0];
}
// Paired items are twice as long and half as many, so a layer needs enough of them to keep filling the chip: with
// fewer than two rounds of n_cu paired items the product loses more to its last round (and to the smaller pieces that
// round is cut into) than the tail gains from reading one word instead of two.  Measured, paired against single-limb
// items, whole layer call: 256 -> 1024 1x1 at 14x14 (552 paired items) -3.7 %, 512 -> 1024 stride 2 -2.5 %,
// 512 -> 2048 1x1 at 7x7 -1.3 %; 256 -> 256 3x3 at 14x14 (138 paired items, one round of double length against 1.25
// rounds) +13.8 %, 64 -> 64 3x3 at 56x56 (125) +9 %.  In the kernels: the tail 87.7 -> 80.0 us per 1024 output
// ciphertexts, the product unchanged at 69-70 us although it writes half the bytes.
// MMH_PW_PAIRED=1 pairs every layer (tests), =0 none.
static bool pw_paired_pays(const mmh_conv_plan *p) {
  const char *e = getenv("MMH_PW_PAIRED");
  if (e && e[0] == '1') return true;
  const int n_live = (int)p->pw_live_tiles, M = (int)p->meta.n_filters, n_cu = p->ctx->n_cu;
  return pw_items(n_live / 2, M, n_cu).n_items >= 2 * n_cu;
}
static PwSizes pw_sizes(const mmh_conv_plan *p) {
  PwSizes z;
  z.M = (int)p->meta.n_filters, z.Mpad = (z.M + 63) / 64 * 64, z.C = (int)p->f.count(), z.wc = (int)p->split.win_c;
  z.Cpad = (z.C + kPwChunk - 1) / kPwChunk * kPwChunk;
  z.Mpad128 = (z.M + 127) / 128 * 128, z.Cpad64 = (z.C + 63) / 64 * 64;
  return z;
}
static PwWeightArgs pw_weight_args(const mmh_conv_plan *p, const u64 *src, u32 *wp, u64 *wl, uint8_t *w8) {
  const mmh_ctx *ctx = p->ctx;
  const PwSizes z = pw_sizes(p);
  return PwWeightArgs{src, wp, wl, (uint4 *)w8, z.M, z.Mpad, z.Mpad128, z.C, z.Cpad, z.Cpad64, p->d_pw_chan,
                      (int)p->n_cslices, ctx->logN, ctx->bfv.t, ctx->bfv.thr, ctx->q[0], ctx->q[1]};
}
// weights of the 1x1 route from the raw filters or from encoded plaintexts; flag must be zeroed first
static int pointwise_weights(const mmh_conv_plan *p, const u64 *d_filters_raw, const u64 *d_filter_pts, u32 *wp,
                             u64 *wl, uint8_t *w8, u32 *flag, hipStream_t s, bool flag_is_clear = false,
                             u32 *nzmap = nullptr) {
  const PwSizes z = pw_sizes(p);
  if (!flag_is_clear) MMH_HIP(hipMemsetAsync(flag, 0, 4, s));
  const unsigned blocks = nblk((size_t)(z.Cpad64 / 16) * z.Mpad128);
  if (d_filters_raw)
    hipLaunchKernelGGL(pw_weights_kernel<false>, dim3(blocks), dim3(256), 0, s, pw_weight_args(p, d_filters_raw, wp, wl, w8),
                       flag, nzmap);
  else
    hipLaunchKernelGGL(pw_weights_kernel<true>, dim3(blocks), dim3(256), 0, s, pw_weight_args(p, d_filter_pts, wp, wl, w8),
                       flag, nzmap);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
// acc[m*S + sp] ([2][2][N] each, coefficient form) for all filters from the coefficient-form image
// Whether a fresh front-launch call's encryptions of zero ride in the product launch: the one-launch narrow kernel at
// N = 4096 (pointwise_product takes it under the same condition), and only while every workgroup of that launch gets a
// compute unit of its own -- with more tiles than that the riders delay products (128 -> 32 3x3 at 56x56, 400 tiles:
// +11 %) where at 28x28 and below the call gains 12-16 %.  MMH_ZERO_RIDE=0 keeps them in the front launch, =1 lets
// them ride whatever the grid.
static bool pw_zero_rides(const mmh_conv_plan *p, bool switched) {
  const char *e = getenv("MMH_ZERO_RIDE");
  if (e && e[0] == '0') return false;
  static_assert(2 * lds_bytes<12>() <= (size_t)kPwNarrowLds, "two transform images in the narrow kernel's LDS");
  if (p->ctx->logN != 12 || !pw_mfma_enabled(p->ctx) || !pw_narrow_ok(p) || switched) return false;
  return e ? true : (int)(p->pw_live_tiles + p->n_out) <= p->ctx->n_cu;
}
static int pointwise_product(const mmh_conv_plan *p, const u64 *image, const u32 *wp, const u64 *wl, const uint8_t *w8,
                             const u32 *flag, u64 *acc, hipStream_t s, bool fill_dead, int form = -1,
                             const u32 *nzmap = nullptr, u32 *zero_filter = nullptr, const PwFold *fold = nullptr,
                             void *verdict_event = nullptr, const u64 *raw_filters = nullptr, bool switched = false,
                             const ZeroRide *ride = nullptr) {
  // switched: acc receives the MOD-SWITCHED product, [M][S][2][N] at the last level (pw_switched_ok() must hold)
  // ride: encryptions of zero to make in the narrow kernel's launch (the caller has checked pw_zero_rides())
  const mmh_ctx *ctx = p->ctx;
  const size_t N = ctx->N, S = p->n_spatial, cols = S * 4 * N;
  MMH_REQUIRE(!switched || (pw_switched_ok(ctx) && form != 1 && !fill_dead), MMH_ERR_INTERNAL);
  const PwSizes z = pw_sizes(p);
  // live column tiles: all of c1, of c0 only those with a kept coefficient (about n_targets / 64 per limb)
  const size_t live_tiles = S * (2 * N / 64) + 2 * (p->n_targets / 64 + S);
  const bool wide = live_tiles * (size_t)(z.Mpad / 64) >= 1024;   // 4 workgroups per CU with 16 filters per wave
  auto launch = [&](auto kern, int tf) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(cols / 64), (unsigned)(z.Mpad / tf)), dim3(256), 0, s, image, wp, wl, flag,
                       p->d_keep, p->d_pw_chan, acc, z.M, z.Mpad, z.Cpad, S, ctx->logN, ctx->d_mods);
  };
  // the device-side flag picks the kernel that runs; a caller that already knows the form skips the other
  const bool mfma = pw_mfma_enabled(ctx);
  if (mfma && form != 1) {
    // fast form on the matrix cores: byte digits of the rotated columns once, then the i8 products
    const int KC = z.Cpad64 / 64, n_live = (int)p->pw_live_tiles;
    const int skip7n = ((ctx->q[0] >> 54) == 0 ? 1 : 0) | ((ctx->q[1] >> 54) == 0 ? 2 : 0);
    if (pw_narrow_ok(p) && !nzmap && !switched) {   // one launch, no digit image (the non-front weight pass keeps the image)
      if (fill_dead) MMH_HIP(hipMemsetAsync(acc, 0, (size_t)z.M * cols * 8, s));
      const ZeroRide zr = ride ? *ride : ZeroRide{};
      hipLaunchKernelGGL(pw_narrow_kernel, dim3((unsigned)n_live + zr.wgs), dim3(512), kPwNarrowLds, s, image, p->d_pw_chan,
                         p->d_pw_live, (const uint4 *)w8, flag, acc, z.M, KC, S, ctx->logN, ctx->d_mods, wl, z.Mpad, z.Cpad,
                         PwLift{raw_filters, z.C, ctx->bfv.t, ctx->bfv.thr},
                         fold ? *fold : PwFold{nullptr, 0, nullptr, nullptr, 0, 0}, zero_filter, skip7n, zr);
      MMH_HIP(hipGetLastError());
      if (verdict_event) MMH_HIP(hipEventRecord((hipEvent_t)verdict_event, s));   // the zero-filter word is final here
      return MMH_OK;
    }
    void *u8 = nullptr;
    if (scratch_alloc(&u8, (size_t)n_live * KC * 32768, s) != hipSuccess) {
      set_error("pointwise_product: allocation failed");
      return MMH_ERR_MEMORY;
    }
    if (fill_dead) MMH_HIP(hipMemsetAsync(acc, 0, (size_t)z.M * cols * 8, s));
    const int skip7 = ((ctx->q[0] >> 54) == 0 ? 1 : 0) | ((ctx->q[1] >> 54) == 0 ? 2 : 0);
    hipLaunchKernelGGL(pw_expand_kernel, dim3((unsigned)n_live, (unsigned)KC), dim3(256), 0, s, image, p->d_pw_chan,
                       p->d_pw_live, (uint4 *)u8, flag, KC, S, ctx->logN, nzmap, z.M, zero_filter,
                       fold ? *fold : PwFold{nullptr, 0, nullptr, nullptr, 0, 0}, skip7);
    if (verdict_event) MMH_HIP(hipEventRecord((hipEvent_t)verdict_event, s));   // the zero-filter word is final here
    const int n_col = switched ? n_live / 2 : n_live;   // column tiles of the items: pairs or single tiles
    PwItems items = pw_items(n_col, z.M, ctx->n_cu);
    items.skip7 = skip7;
    items.paired = switched ? 1 : 0;
    const u64 q0 = ctx->q[0], q1 = ctx->q[1];
    const PwSwitch sw{ctx->bfv.inv_qlast_mod_q0, shoup(ctx->bfv.inv_qlast_mod_q0, q0), q1 >> 1, (q1 >> 1) % q0};
    const char *stamp_file = getenv("MMH_PW_STAMPS");   // diagnostic: the stamped instantiation, its table appended to that file
    if (stamp_file && stamp_file[0]) {
      const size_t words = (size_t)8 * kPwStampSteps * 5;
      u64 *d_st = nullptr;
      MMH_HIP(hipMalloc((void **)&d_st, words * 8));
      MMH_HIP(hipMemsetAsync(d_st, 0, words * 8, s));
      MMH_HIP(hipFuncSetAttribute((const void *)pw_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPwLdsBytes));
      hipLaunchKernelGGL(pw_mfma_kernel<true>, dim3((unsigned)std::min(items.n_items, ctx->n_cu)), dim3(512), kPwLdsBytes, s,
                         (const uint4 *)u8, (const uint4 *)w8, flag, p->d_pw_live, acc, z.M, KC, n_col, S, ctx->logN,
                         ctx->d_mods, image, p->d_pw_chan, wl, z.Mpad, z.Cpad, items,
                         PwLift{raw_filters, z.C, ctx->bfv.t, ctx->bfv.thr}, sw, d_st);
      std::vector<u64> h(words);
      MMH_HIP(hipMemcpyAsync(h.data(), d_st, words * 8, hipMemcpyDeviceToHost, s));
      MMH_HIP(hipStreamSynchronize(s));
      (void)hipFree(d_st);
      if (FILE *f = fopen(stamp_file, "a")) {
        fprintf(f, "# pw_mfma_kernel stamps: M=%d KC=%d items=%d paired=%d F=%d (workgroup 0; wave step t0 d1 d2 d3 d4: shader "
                   "cycles from t0 to barrier passed / copies issued / fragments read / last product issued)\n",
                z.M, KC, items.n_items, items.paired, 1 << items.F_log);
        for (int w = 0; w < 8; ++w)
          for (int st = 0; st < kPwStampSteps; ++st) {
            const u64 *r = &h[((size_t)w * kPwStampSteps + st) * 5];
            if (!r[0]) break;
            fprintf(f, "%d %d %llu %lld %lld %lld %lld\n", w, st, (unsigned long long)r[0], (long long)(r[1] - r[0]),
                    (long long)(r[2] - r[0]), (long long)(r[3] - r[0]), (long long)(r[4] - r[0]));
          }
        fclose(f);
      }
    } else
    hipLaunchKernelGGL(pw_mfma_kernel<false>, dim3((unsigned)std::min(items.n_items, ctx->n_cu)), dim3(512), kPwLdsBytes, s,
                       (const uint4 *)u8, (const uint4 *)w8, flag, p->d_pw_live, acc, z.M, KC, n_col, S, ctx->logN,
                       ctx->d_mods, image, p->d_pw_chan, wl, z.Mpad, z.Cpad, items,
                       PwLift{raw_filters, z.C, ctx->bfv.t, ctx->bfv.thr}, sw, (u64 *)nullptr);
    (void)scratch_free(u8, s);
  }
  // (on the matrix-core path the product kernel carries the lifted-weight form as a branch of its own)
  const bool fallback_launch = form != 0 && (!mfma || form == 1);
  if (wide) {
    if (!mfma && form != 1) launch(pw_gemm_kernel<true, 16>, 64);
    if (fallback_launch) launch(pw_gemm_kernel<false, 16>, 64);
  } else {
    if (!mfma && form != 1) launch(pw_gemm_kernel<true, 8>, 32);
    if (fallback_launch) launch(pw_gemm_kernel<false, 8>, 32);
  }
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
static int pointwise_mac(const mmh_conv_plan *p, const u64 *image, const u64 *d_filters_raw,
                         const u64 *d_filter_pts, u64 *acc, hipStream_t s, bool fill_dead, u32 *clear_flag = nullptr,
                         u32 *nzmap = nullptr, u32 *zero_filter = nullptr) {
  const PwSizes z = pw_sizes(p);
  void *wp = nullptr, *wl = nullptr, *w8 = nullptr, *flag = clear_flag;
  int rc = MMH_OK;
  auto grab = [&](void **b, size_t bytes) {
    if (!rc && scratch_alloc(b, bytes, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  };
  grab(&wp, (size_t)z.Cpad * z.Mpad * 4);
  grab(&wl, (size_t)z.Cpad * z.Mpad * 16);
  grab(&w8, z.w8_bytes());
  if (!clear_flag) grab(&flag, 4);
  if (rc) set_error("pointwise_mac: allocation failed");
  if (!rc)
    rc = pointwise_weights(p, d_filters_raw, d_filter_pts, (u32 *)wp, (u64 *)wl, (uint8_t *)w8, (u32 *)flag, s,
                           clear_flag != nullptr, nzmap);
  if (!rc)
    rc = pointwise_product(p, image, (const u32 *)wp, (const u64 *)wl, (const uint8_t *)w8, (const u32 *)flag, acc, s,
                           fill_dead, -1, nzmap, zero_filter);
  for (void *b : {wp, wl, w8, clear_flag ? nullptr : flag})
    if (b) (void)scratch_free(b, s);
  return rc;
}

template <class T>
static int upload_vec(mmh_conv_plan *p, const std::vector<T> &v, T **d) {
  MMH_HIP(hipMalloc((void **)d, std::max<size_t>(1, v.size()) * sizeof(T)));
  p->allocs.push_back(*d);
  if (!v.empty()) MMH_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return MMH_OK;
}
}  // namespace mmh

using namespace mmh;
#define CHECK_RC(x)      \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)

extern "C" int mmh_conv_plan_destroy(mmh_conv_plan *p) {
  if (!p) return MMH_OK;
  for (void *a : p->allocs) (void)hipFree(a);
  delete p;
  return MMH_OK;
}

extern "C" int mmh_conv_plan_create(const mmh_ctx *ctx, const mmh_conv_meta *meta, mmh_conv_plan **out) {
  MMH_DEVICE_SCOPE(ctx);
  MMH_REQUIRE(ctx && meta && out, MMH_ERR_NULL_POINTER);
  *out = nullptr;
  MMH_REQUIRE(ctx->t >= 2, MMH_ERR_CONFIG);
  const long N = (long)ctx->N;
  const Dim3 in{meta->in_c, meta->in_h, meta->in_w}, f{meta->in_c, meta->f_h, meta->f_w};
  MMH_REQUIRE(in.count() > 0 && f.count() > 0 && meta->n_filters > 0 && meta->stride >= 1, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(f.h * f.w <= N, MMH_ERR_INVALID_ARG);
  const bool same = meta->padding_same != 0;
  const long stride = meta->stride;
  const ConvSplit sp = conv_split(N, in, f, same, stride);
  MMH_REQUIRE(sp.ok, MMH_ERR_INVALID_ARG);                       // EncodeImageShare: shape_inference failed
  MMH_REQUIRE(sp.win_h >= f.h && sp.win_w >= f.h, MMH_ERR_INVALID_ARG);  // sliced_3d_tensor.h:148-149
  MMH_REQUIRE(sp.win_c * sp.win_h * sp.win_w <= N, MMH_ERR_OUT_BOUND);
  // result shape (shape_inference.cc:28-52)
  const long oh = (sp.padded_shape.h - f.h + stride) / stride, ow = (sp.padded_shape.w - f.w + stride) / stride;
  MMH_REQUIRE(oh > 0 && ow > 0, MMH_ERR_CONFIG);

  mmh_conv_plan *p = new mmh_conv_plan();
  p->ctx = ctx;
  p->meta = *meta;
  p->in = in;
  p->f = f;
  p->split = sp;
  p->out_h = oh;
  p->out_w = ow;
  // Strided3DTensor (tensor_encoder.cc:197-204): subsample where the split shrank the axis
  p->sub_h = sp.strided_shape.h < in.h ? (int)stride : 1;
  p->sub_w = sp.strided_shape.w < in.w ? (int)stride : 1;
  const Dim3 simg{in.c, (in.h + p->sub_h - 1) / p->sub_h, (in.w + p->sub_w - 1) / p->sub_w};
  p->cut_c.init(simg.c, 0, sp.win_c, f.c, false);
  p->cut_h.init(simg.h, sp.pad_h, sp.win_h, f.h, true);
  p->cut_w.init(simg.w, sp.pad_w, sp.win_w, f.w, true);
  p->n_cslices = (size_t)p->cut_c.count;
  p->n_spatial = (size_t)(p->cut_h.count * p->cut_w.count);
  p->n_in = p->n_cslices * p->n_spatial;
  p->n_out = (size_t)meta->n_filters * p->n_spatial;
  // FilterIndexer (tensor_encoder.cc:78-84): "h" is taken from the filter WIDTH
  p->filt_origin = sp.win_h * sp.win_w * (sp.win_c - 1) + sp.win_w * (f.w - 1) + f.w - 1;
  int rc = MMH_OK;
  auto fail = [&](int code, const char *msg) {
    mmh_conv_plan_destroy(p);
    set_error(std::string("mmh_conv_plan_create: ") + msg);
    return code;
  };
  if (p->filt_origin >= N || p->filt_origin - (sp.win_c - 1) * sp.win_h * sp.win_w - (f.h - 1) * sp.win_w - (f.w - 1) < 0)
    return fail(MMH_ERR_OUT_BOUND, "filter coefficient index out of range");

  std::vector<ImgWindow> wins(p->n_in);
  for (size_t i = 0; i < p->n_in; ++i) {
    const long kc = (long)(i / p->n_spatial), kh = (long)((i % p->n_spatial) / p->cut_w.count),
               kw = (long)(i % p->cut_w.count);
    long c0, y0, x0, lo, hi;
    ImgWindow g{};
    if (!p->cut_c.clip(kc, c0, lo, hi)) return fail(MMH_ERR_INTERNAL, "empty channel window");
    g.c0 = (int)c0;
    g.nc = (int)p->cut_c.size(kc);
    if (!p->cut_h.clip(kh, y0, lo, hi)) return fail(MMH_ERR_INTERNAL, "empty row window");
    g.y0 = (int)y0, g.lo_y = (int)lo, g.hi_y = (int)hi, g.nh = (int)p->cut_h.size(kh);
    if (!p->cut_w.clip(kw, x0, lo, hi)) return fail(MMH_ERR_INTERNAL, "empty column window");
    g.x0 = (int)x0, g.lo_x = (int)lo, g.hi_x = (int)hi, g.nw = (int)p->cut_w.size(kw);
    wins[i] = g;
  }
  // Output targets: the index calculator that addRandomMask / decryptToTensor build
  // (hom_conv2d_ss.cc:738-744, :874-880): on the subsampled shape with stride 1 when the input
  // was compressed, else on the original shape with the real stride.  Its cuts equal ours.
  const bool compressed = sp.strided_shape.count() < in.count();
  const long out_stride = compressed ? 1 : stride;
  std::vector<u32> coef, dst, slice;
  std::vector<uint8_t> keep(p->n_spatial * (size_t)N, 0);
  long hoff = 0;
  for (long kh = 0; kh < p->cut_h.count; ++kh) {
    long woff = 0, last_h = 0;
    for (long kw = 0; kw < p->cut_w.count; ++kw) {
      const size_t s = (size_t)(kh * p->cut_w.count + kw);
      const OutWindow o = out_window(p->cut_h, p->cut_w, kh, kw, f, out_stride);
      for (long y = 0; y < o.out_h; ++y)
        for (long x = 0; x < o.out_w; ++x) {
          const long r = y * out_stride + o.first_r, c = x * out_stride + o.first_c;
          if (r >= sp.win_h || c >= sp.win_w) return fail(MMH_ERR_OUT_BOUND, "target outside the window");
          if (hoff + y >= oh || woff + x >= ow) return fail(MMH_ERR_DIM_MISMATCH, "windows overrun the result");
          coef.push_back((u32)(p->filt_origin + r * sp.win_w + c));
          dst.push_back((u32)((hoff + y) * ow + woff + x));
          slice.push_back((u32)s);
        }
      // removeUnusedCoeffs (hom_conv2d_ss.cc:817-819) always indexes with stride 1, VALID, on the
      // padded strided shape: a superset of the targets when stride > 1 without compression
      const OutWindow k1 = out_window(p->cut_h, p->cut_w, kh, kw, f, 1);
      for (long y = 0; y < k1.out_h; ++y)
        for (long x = 0; x < k1.out_w; ++x) keep[s * N + p->filt_origin + y * sp.win_w + x] = 1;
      woff += o.out_w;
      last_h = o.out_h;
    }
    hoff += last_h;
  }
  p->n_targets = coef.size();
  std::vector<int> share_dst(p->n_spatial * (size_t)N, -1);
  for (size_t g = 0; g < coef.size(); ++g) share_dst[(size_t)slice[g] * N + coef[g]] = (int)dst[g];
  // Direct route (pointwise_mac): the layer as a mod-q matrix product over the filter taps instead of
  // transforming M*Cs sparse filter polynomials.  Tap k = (c, r, col) of a filter multiplies the input
  // polynomial of channel slice c / wc rotated by its coefficient index (EncodeFilter's, above).
  const long taps = f.count();
  // (at most 5120 taps: the i32 digit sums stay below 3 * taps * 2^14 < 2^28, the 64-bit ones of the vector form
  // below 5120 * 2^51 < 2^64)
  p->direct_capable = taps <= 5120 && ctx->L == 2;
  if (p->direct_capable) {
    std::vector<PwChannel> chan((size_t)(taps + 63) / 64 * 64, PwChannel{0, 0});
    for (long k = 0; k < taps; ++k) {
      const long c = k / (f.h * f.w), r = (k / f.w) % f.h, col = k % f.w;
      chan[k] = PwChannel{(int)(c / sp.win_c),
                          (int)(p->filt_origin - (c % sp.win_c) * sp.win_h * sp.win_w - r * sp.win_w - col)};
    }
    // 64-column tiles (spatial window, c0/c1, limb, 64 coefficients) the product has to compute: all of c1, of
    // c0 those holding a kept coefficient
    // (listed in pairs -- the limb-1 tile, then the limb-0 tile of the same 64 coefficients -- so that an item of
    // the product kernel can own both limbs, PwItems::paired; the digit image follows this order)
    std::vector<int> live_tiles;
    for (size_t b = 0; b < p->n_spatial * 4 * (size_t)N / 64; ++b) {
      const size_t sp_i = (b * 64) / (4 * (size_t)N), pl = ((b * 64) / (size_t)N) & 3, i0 = (b * 64) % (size_t)N;
      if (pl & 1) continue;   // limb-0 tiles name the pairs
      bool live = pl >= 2;
      for (size_t i = 0; i < 64 && !live; ++i) live = keep[sp_i * N + i0 + i] != 0;
      if (!live) continue;
      live_tiles.push_back((int)(b + (size_t)N / 64)), live_tiles.push_back((int)b);
      ++p->pw_live_tiles_limb[0], ++p->pw_live_tiles_limb[1];
    }
    p->pw_live_tiles = live_tiles.size();
    p->pw_taps_padded = chan.size();
    // the product's two LDS buffers exceed the static 64 KiB: opt in (per device; plans are made on the context's)
    if (hipFuncSetAttribute((const void *)pw_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPwLdsBytes) !=
            hipSuccess ||
        hipFuncSetAttribute((const void *)pw_narrow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPwNarrowLds) !=
            hipSuccess) {
      set_error("mmh_conv_plan_create: hipFuncSetAttribute failed");
      mmh_conv_plan_destroy(p);
      return MMH_ERR_INTERNAL;
    }
    if ((rc = upload_vec(p, chan, &p->d_pw_chan)) || (rc = upload_vec(p, live_tiles, &p->d_pw_live))) {
      mmh_conv_plan_destroy(p);
      return rc;
    }
    // Which route is cheaper (microseconds, from the measured rates of DESIGN.md 5): the direct product costs
    // 16 byte products per tap, filter and live column plus the digit image; the transform route two filter
    // transforms (none when prepared) and S dyadic products per filter polynomial plus the transforms of the
    // image and of the result.  MMH_CONV_POINTWISE=0 / 1 forces the transform / the direct route.
    const double M_ = (double)meta->n_filters, Cs_ = (double)p->n_cslices, S_ = (double)p->n_spatial;
    const double live_cols = 64.0 * (double)p->pw_live_tiles, kpad = (double)chan.size();
    const double t_direct = std::ceil(M_ / 128) * 128 * kpad * live_cols * 16 / 0.8e9 + kpad * live_cols * 8 / 1.5e6;
    const double t_rest = ((double)p->n_in * 4 + M_ * S_ * 4) / 54.0;
    const double t_fresh = M_ * Cs_ * (2 + S_) / 54.0 + t_rest, t_prepared = M_ * Cs_ * S_ / 54.0 + t_rest;
    const char *pw_env = getenv("MMH_CONV_POINTWISE");
    const bool never = pw_env && pw_env[0] == '0', always = pw_env && pw_env[0] == '1';
    p->direct_fresh = !never && (always || t_direct < t_fresh);
    p->direct_prepared = !never && (always || t_direct < t_prepared);
    // Layers of at most 32 filters multiply in one launch that builds its digits itself (pw_narrow_kernel) or in the
    // two launches of the digit image, and either beats the transform route's eight or nine launches at every size measured (128 -> 32 3x3 at 56x56 / 28x28 / 14x14 /
    // 7x7: 88 / 80 / 74 / 75 us against 328 / 133 / 92 / 82 us per layer call; 256 -> 32 1x1 at 27x27: 56 against 167)
    if (!never && meta->n_filters <= 32 && pw_mfma_enabled(ctx)) p->direct_fresh = p->direct_prepared = true;
  }
  std::vector<u32> kept_idx, kept_off(p->n_spatial + 1, 0);
  std::vector<int> kept_rank(p->n_spatial * (size_t)N, -1);
  for (size_t s_ = 0; s_ < p->n_spatial; ++s_) {
    kept_off[s_] = (u32)kept_idx.size();
    for (long i = 0; i < N; ++i)
      if (keep[s_ * N + i]) {
        kept_rank[s_ * N + i] = (int)(kept_idx.size() - kept_off[s_]);
        kept_idx.push_back((u32)i);
      }
    p->max_kept = std::max(p->max_kept, kept_idx.size() - kept_off[s_]);
  }
  kept_off[p->n_spatial] = (u32)kept_idx.size();
  if ((rc = upload_vec(p, kept_idx, &p->d_kept_idx)) || (rc = upload_vec(p, kept_off, &p->d_kept_off)) ||
      (rc = upload_vec(p, kept_rank, &p->d_kept_rank))) {
    mmh_conv_plan_destroy(p);
    return rc;
  }
  if ((rc = upload_vec(p, wins, &p->d_windows)) || (rc = upload_vec(p, coef, &p->d_tgt_coef)) ||
      (rc = upload_vec(p, dst, &p->d_tgt_dst)) || (rc = upload_vec(p, slice, &p->d_tgt_slice)) ||
      (rc = upload_vec(p, keep, &p->d_keep)) || (rc = upload_vec(p, share_dst, &p->d_share_dst))) {
    mmh_conv_plan_destroy(p);
    return rc;
  }
  *out = p;
  return MMH_OK;
}

extern "C" int mmh_conv_plan_info(const mmh_conv_plan *p, size_t *n_in_cts, size_t *n_channel_slices,
                                  size_t *n_spatial_slices, size_t *n_out_cts, int64_t out_shape[3],
                                  int64_t slice_strides[3]) {
  MMH_REQUIRE(p, MMH_ERR_NULL_POINTER);
  if (n_in_cts) *n_in_cts = p->n_in;
  if (n_channel_slices) *n_channel_slices = p->n_cslices;
  if (n_spatial_slices) *n_spatial_slices = p->n_spatial;
  if (n_out_cts) *n_out_cts = p->n_out;
  if (out_shape) out_shape[0] = p->meta.n_filters, out_shape[1] = p->out_h, out_shape[2] = p->out_w;
  if (slice_strides) slice_strides[0] = p->split.win_c, slice_strides[1] = p->split.win_h, slice_strides[2] = p->split.win_w;
  return MMH_OK;
}

extern "C" int mmh_conv_plan_work(const mmh_conv_plan *p, mmh_conv_work *w) {
  MMH_REQUIRE(p && w, MMH_ERR_NULL_POINTER);
  w->direct_fresh = p->direct_fresh, w->direct_prepared = p->direct_prepared;
  w->taps = p->f.count(), w->taps_padded = (int64_t)p->pw_taps_padded;
  w->live_columns[0] = 64 * (int64_t)p->pw_live_tiles_limb[0], w->live_columns[1] = 64 * (int64_t)p->pw_live_tiles_limb[1];
  w->n_in = (int64_t)p->n_in, w->n_out = (int64_t)p->n_out, w->n_cslices = (int64_t)p->n_cslices;
  w->n_spatial = (int64_t)p->n_spatial, w->n_targets = (int64_t)p->n_targets;
  w->limbs = (int64_t)p->ctx->L, w->poly_degree = (int64_t)p->ctx->N;
  return MMH_OK;
}

extern "C" int mmh_conv_encode_image(const mmh_conv_plan *p, const uint64_t *d_img, uint64_t *d_pts, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_img && d_pts, MMH_ERR_NULL_POINTER);
  const mmh_ctx *ctx = p->ctx;
  const size_t total = p->n_in << ctx->logN;
  hipLaunchKernelGGL(encode_image_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, d_img, d_pts,
                     p->d_windows, p->n_in, ctx->logN, (int)p->split.win_c, (int)p->split.win_h, (int)p->split.win_w,
                     (int)p->in.h, (int)p->in.w, p->sub_h, p->sub_w, ctx->bfv.t);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_conv_encode_filters(const mmh_conv_plan *p, const uint64_t *d_filters, uint64_t *d_pts,
                                       void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_filters && d_pts, MMH_ERR_NULL_POINTER);
  const mmh_ctx *ctx = p->ctx;
  hipStream_t s = (hipStream_t)stream;
  const size_t n_polys = (size_t)p->meta.n_filters * p->n_cslices;
  MMH_HIP(hipMemsetAsync(d_pts, 0, (n_polys << ctx->logN) * 8, s));
  const size_t total = (size_t)p->meta.n_filters * p->f.count();
  hipLaunchKernelGGL(encode_filter_kernel, dim3(nblk(total)), dim3(256), 0, s, d_filters, d_pts, total, ctx->logN,
                     (int)p->f.c, (int)p->f.h, (int)p->f.w, (int)p->split.win_c, (int)p->n_cslices, p->filt_origin,
                     p->split.win_h * p->split.win_w, p->split.win_w, ctx->bfv.t);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_conv_check_filters(const mmh_conv_plan *p, const uint64_t *d_filters, uint32_t *d_flag,
                                      void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_filters && d_flag, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  MMH_HIP(hipMemsetAsync(d_flag, 0, 4, s));
  hipLaunchKernelGGL(zero_filter_kernel, dim3((unsigned)p->meta.n_filters), dim3(256), 0, s, d_filters,
                     (size_t)p->f.count(), p->ctx->bfv.t, d_flag);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_conv_encrypt_image(const mmh_conv_plan *p, const uint64_t *d_sk_ntt, const uint64_t *d_img,
                                      const uint64_t *d_a, const int64_t *d_e, uint64_t *d_cts, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_sk_ntt && d_img && d_a && d_e && d_cts, MMH_ERR_NULL_POINTER);
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (p->n_in << p->ctx->logN) * 8, s));
  int rc = mmh_conv_encode_image(p, d_img, (u64 *)pts, stream);
  if (!rc) rc = mmh_encrypt_symmetric(p->ctx, 0, d_sk_ntt, d_a, d_e, (const u64 *)pts, p->n_in, d_cts, stream);
  (void)scratch_free(pts, s);
  return rc;
}

// conv2DSS for all filters.  Exactly one of d_filter_pts (encoded plaintexts, [M*Cs][N]) and
// d_filters_raw ([M][C][fh][fw]) is given; both produce the same bits.
static int conv_server_impl(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                            const uint64_t *d_filter_pts, const uint64_t *d_filters_raw,
                            const uint64_t *d_pk_ntt_last, const int64_t *d_u, const int64_t *d_e1,
                            const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share, void *stream,
                            const uint64_t *d_flood = nullptr, const int64_t *d_e0 = nullptr,
                            const mmh_conv_filters *prepared = nullptr, const uint64_t *d_share_raw = nullptr,
                            uint32_t *d_zero_filter = nullptr, const uint64_t *d_zero_enc = nullptr,
                            void *verdict_event = nullptr, const uint32_t *draw_key = nullptr, uint64_t draw_sid = 0,
                            size_t draw_first = 0) {
  // draw_key: the call draws u, e1 and the mask itself -- polynomials [draw_first, draw_first + n_out) of the streams
  // mmh_sample_reshare(key, draw_sid) makes -- instead of taking them as arrays (d_u, d_e1, d_mask null)
  const bool exact = d_flood != nullptr;  // USE_APPROX_RESHARE=0 build
  MMH_REQUIRE(!exact || d_e0, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!prepared || prepared->plan == p, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(!draw_key || (!exact && !d_zero_enc && !d_u && !d_e1 && !d_mask), MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(p && d_in_cts && (d_filter_pts || d_filters_raw || prepared) &&
                  ((d_pk_ntt_last && (d_u || draw_key)) || d_zero_enc) && ((d_e1 && d_mask) || draw_key) && d_out_cts &&
                  d_out_share,
              MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!d_zero_enc || !exact, MMH_ERR_INVALID_ARG);
  MMH_REQUIRE(!p->meta.is_shared_input || d_share_pts || d_share_raw, MMH_ERR_DIM_MISMATCH);  // conv2DSS :633-637
  const mmh_ctx *ctx = p->ctx;
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = ctx->N, M = (size_t)p->meta.n_filters, S = p->n_spatial, Cs = p->n_cslices;
  void *image = nullptr, *acc = nullptr, *zero = nullptr, *f_ntt = nullptr, *mask_t = nullptr;
  void *wp = nullptr, *wl = nullptr, *w8 = nullptr, *anymap = nullptr;   // weight forms of a fresh direct-route call
  void *u_ntt = nullptr;                                                  // u of the tail, transformed by the front launch
  void *dr_u = nullptr, *dr_e1 = nullptr, *dr_mask = nullptr, *dr_zero = nullptr;   // a drawing call's own randomness
  void *flag_word = nullptr;   // the direct route's weight-form flag, cleared by the kernel that builds the image
  // the route: prepared filters carry the one they were prepared for
  const bool direct = prepared ? prepared->w8 != nullptr : p->direct_fresh;
  int rc = MMH_OK;
  auto cleanup = [&]() {
    for (void *b : {image, acc, zero, f_ntt, mask_t, flag_word, wp, wl, w8, anymap, u_ntt, dr_u, dr_e1, dr_mask, dr_zero})
      if (b) (void)scratch_free(b, s);
  };
  auto grab = [&](void **b, size_t bytes) {
    if (!rc && scratch_alloc(b, bytes, s) != hipSuccess) rc = MMH_ERR_MEMORY;
  };
  grab(&image, p->n_in * 4 * N * 8);
  grab(&acc, p->n_out * 4 * N * 8);
  if (!direct && !prepared) grab(&f_ntt, M * Cs * 2 * N * 8);
  // conv2DSS :655-671: image = share0 + Delta*share1 (or share0 alone), then to NTT form once
  // The conv2DOneFilter :573-580 verdict, when the caller wants it from this call: on the direct route it falls
  // out of the weight pass (a map of filters with a non-zero weight, read by the next kernel), else a pass of its own
  const bool via_image_kernel = d_share_raw && p->meta.is_shared_input;
  const ImageShareArgs img_args{d_in_cts, d_share_raw, (u64 *)image, p->d_windows, p->n_in, ctx->logN,
                                (int)p->split.win_c, (int)p->split.win_h, (int)p->split.win_w, (int)p->in.h, (int)p->in.w,
                                p->sub_h, p->sub_w};
  // the front of a fresh direct-route call from the raw tensors is one launch (conv_front_kernel): image and weight
  // forms side by side, flag bits and the zero-filter map left for the digit kernel to fold
  // (with prepared filters the launch has no weight part: image and u's transform)
  const bool front = via_image_kernel && direct && (prepared || d_filters_raw) && pw_mfma_enabled(ctx);
  const bool nz_inline = d_zero_filter && front && !prepared;
  // A drawing call: where the front launch exists and the tail's u is transformed there, the draws happen inside it
  // (FrontDraw); elsewhere one sampling launch makes the arrays here and the call goes on as if they had been passed
  const bool draw_in_front = draw_key && front && ctx->h_mods[0].fold_c != 0 && !ctx->h_mods[0].f64;
  const bool draw_zero = draw_in_front && ctx->logN == 12 && p->n_out <= reshare_split_max();
  // The approximate tail's first act on the accumulator is divide_and_round_q_last: where the matrix-core product
  // runs, its items own both limbs and write the switched value, one word per coefficient instead of two
  const bool sw_acc = !exact && direct && pw_switched_ok(ctx) && (prepared ? prepared->form != 1 : front) && pw_paired_pays(p);
  const bool ride_zero = draw_zero && !prepared && pw_zero_rides(p, sw_acc);   // made in the product launch instead
  ZeroRide ride{};
  if (draw_key) {
    grab(&dr_e1, p->n_out * N * 8);
    grab(&dr_mask, p->n_out * N * 8);
    if (!draw_in_front) grab(&dr_u, p->n_out * N * 8);
    if (draw_zero) grab(&dr_zero, p->n_out * 2 * N * 8);
    if (!rc && !draw_in_front)
      rc = mmh_sample_reshare_range(ctx, draw_key, draw_sid, draw_first, p->n_out, (int64_t *)dr_u, (int64_t *)dr_e1,
                                    (u64 *)dr_mask, stream);
    d_u = (const int64_t *)dr_u, d_e1 = (const int64_t *)dr_e1, d_mask = (const u64 *)dr_mask;
  }
  PwFold fold{nullptr, 0, nullptr, nullptr, 0, 0};
  if (!rc && front) {
    const PwSizes z = pw_sizes(p);
    const unsigned n_img = nblk(p->n_in << ctx->logN), n_w = prepared ? 0 : nblk((size_t)(z.Cpad64 / 16) * z.Mpad128);
    if (!prepared) {
      // byte digits only: the biased 32-bit form belongs to the vector-ALU kernels, and the rare lifted-weight branch
      // of the product kernel lifts from the raw tensor itself (20 of 23 bytes per weight not written)
      grab(&w8, z.w8_bytes());
      grab(&flag_word, 4 * (size_t)(1 + n_w));              // the folded word, then one word per weights workgroup
    }
    if (nz_inline) grab(&anymap, (size_t)(z.Cpad64 / 16) * z.Mpad128);
    if (!rc) {
      if (!prepared)
        fold = PwFold{(const u32 *)flag_word + 1, (int)n_w, (u32 *)flag_word, (const uint8_t *)anymap, z.Cpad64 / 16, z.Mpad128};
      // u's transform rides along when the tail will want it (approximate build, q0 of the folding shape)
      const bool with_u = !exact && !d_zero_enc && ctx->h_mods[0].fold_c != 0 && !ctx->h_mods[0].f64;
      if (with_u && !draw_zero) grab(&u_ntt, p->n_out * N * 8);
      const unsigned n_u = with_u && !rc && !ride_zero ? (unsigned)p->n_out * (draw_zero ? 2u : 1u) : 0;
      NttSource usrc{};
      usrc.p = reinterpret_cast<const u64 *>(d_u);
      FrontDraw dr{};
      unsigned n_draw = 0;
      if (draw_in_front) {
        for (int i = 0; i < 8; ++i) usrc.key.k[i] = draw_key[i];
        usrc.sid = draw_sid, usrc.first_tile = draw_first * (N / 4096);
        const size_t first = draw_first * N, total = p->n_out * N;
        auto lo = [&](int per) { return (unsigned)(first / (256 * (size_t)per)); };
        auto cnt = [&](int per) { return (unsigned)((first + total + 256 * (size_t)per - 1) / (256 * (size_t)per)) - lo(per); };
        dr.mode = draw_zero ? 2 : 1, dr.pk = d_pk_ntt_last, dr.zero = (u64 *)dr_zero;
        if (ride_zero && with_u && !rc)
          ride = ZeroRide{(unsigned)p->n_out, d_pk_ntt_last, (u64 *)dr_zero, ctx->d_tables, usrc.key, usrc.sid, usrc.first_tile};
        dr.e1 = (int64_t *)dr_e1, dr.mask = (u64 *)dr_mask;
        dr.n_e = cnt(kNoisePerBlock), dr.n_m = cnt(kUniformPerBlock), dr.lo_e = lo(kNoisePerBlock), dr.lo_m = lo(kUniformPerBlock);
        dr.first = first, dr.total = total;
        n_draw = dr.n_e + dr.n_m;
      }
      const PwWeightArgs wa = pw_weight_args(p, d_filters_raw, (u32 *)wp, (u64 *)wl, (uint8_t *)w8);
      if (!rc && ctx->logN == 12)
        hipLaunchKernelGGL(conv_front_kernel<12>, dim3(n_u + n_img + n_w + n_draw), dim3(256), lds_bytes<12>(), s, img_args,
                           ctx->bfv, ctx->d_mods, n_u, n_img, wa, flag_word ? (u32 *)flag_word + 1 : nullptr, (uint8_t *)anymap,
                           d_zero_filter, (u64 *)u_ntt, usrc, ctx->d_tables, n_w, dr);
      else if (!rc && ctx->logN == 13) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_front_kernel<13>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<13>()) != hipSuccess)
          rc = MMH_ERR_INTERNAL;
        if (!rc)
          hipLaunchKernelGGL(conv_front_kernel<13>, dim3(n_u + n_img + n_w + n_draw), dim3(256), lds_bytes<13>(), s, img_args,
                             ctx->bfv, ctx->d_mods, n_u, n_img, wa, flag_word ? (u32 *)flag_word + 1 : nullptr,
                             (uint8_t *)anymap, d_zero_filter, (u64 *)u_ntt, usrc, ctx->d_tables, n_w, dr);
      } else if (!rc) {
        rc = MMH_ERR_CONFIG;
      }
      if (!rc && hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
      if (!n_u && u_ntt) (void)scratch_free(u_ntt, s), u_ntt = nullptr;
    }
  } else if (!rc && via_image_kernel) {
    const int n_words = direct && !prepared ? 1 : 0;
    if (n_words) grab(&flag_word, 4 * (size_t)n_words);
    if (!rc) {
      hipLaunchKernelGGL(image_plus_share_kernel, dim3(nblk(p->n_in << ctx->logN)), dim3(256), 0, s, img_args, ctx->bfv,
                         ctx->d_mods, (u32 *)flag_word, n_words, d_zero_filter);
      if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
    }
  } else {
    if (!rc && d_zero_filter && hipMemsetAsync(d_zero_filter, 0, 4, s) != hipSuccess) rc = MMH_ERR_INTERNAL;
    if (!rc && p->meta.is_shared_input)
      rc = mmh_add_plain(ctx, 0, d_in_cts, d_share_pts, p->n_in, 0, (u64 *)image, stream);
    if (!rc && !p->meta.is_shared_input &&
        hipMemcpyAsync(image, d_in_cts, p->n_in * 4 * N * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
      rc = MMH_ERR_INTERNAL;
  }
  if (!rc && d_zero_filter && !nz_inline) {
    if (!d_filters_raw) {
      rc = MMH_ERR_INVALID_ARG;
    } else {
      hipLaunchKernelGGL(zero_filter_kernel, dim3((unsigned)M), dim3(256), 0, s, d_filters_raw, (size_t)p->f.count(),
                         ctx->bfv.t, d_zero_filter);
      if (hipGetLastError() != hipSuccess) rc = MMH_ERR_INTERNAL;
    }
  }
  // (the verdict is settled here unless the digit kernel of the direct route folds it: recorded there)
  if (!rc && verdict_event && !nz_inline && hipEventRecord((hipEvent_t)verdict_event, s) != hipSuccess) rc = MMH_ERR_INTERNAL;
  if (!rc && !direct) rc = ntt_dispatch(ctx, false, (u64 *)image, p->n_in * 2, 2, 0, s);
  // multiply_plain's lift + NTT of every filter polynomial, from the plaintexts or straight from the taps
  const u64 *filt_ntt = prepared ? prepared->f_ntt : (const u64 *)f_ntt;
  if (!rc && direct && prepared)
    rc = pointwise_product(p, (const u64 *)image, prepared->wp, prepared->wl, prepared->w8, prepared->flag,
                           (u64 *)acc, s, exact, prepared->form, nullptr, nullptr, nullptr, nullptr, nullptr, sw_acc);
  if (!rc && front && !prepared)
    rc = pointwise_product(p, (const u64 *)image, (const u32 *)wp, (const u64 *)wl, (const uint8_t *)w8,
                           (const u32 *)flag_word, (u64 *)acc, s, exact, -1, nullptr, d_zero_filter, &fold,
                           nz_inline ? verdict_event : nullptr, d_filters_raw, sw_acc, ride.wgs ? &ride : nullptr);
  if (!rc && direct && !prepared && !front)
    rc = pointwise_mac(p, (const u64 *)image, d_filters_raw, d_filter_pts, (u64 *)acc, s, exact, (u32 *)flag_word);
  if (!rc && !direct && !prepared && d_filters_raw)
    rc = ntt_filter_dispatch(ctx, d_filters_raw, (u64 *)f_ntt, M, (int)p->f.c, (int)p->f.h, (int)p->f.w,
                             (int)p->split.win_c, (int)Cs, p->filt_origin, p->split.win_h * p->split.win_w,
                             p->split.win_w, 2, s);
  if (!rc && !direct && !prepared && !d_filters_raw)
    rc = ntt_lift_dispatch(ctx, d_filter_pts, (u64 *)f_ntt, M * Cs, 2, s);
  // conv2DOneFilter :597-616 for every filter: out[m][i] = sum_c image[c*S + i] * filter[m][c]
  if (!rc && !direct)
    rc = mmh_dyadic_mac(ctx, 0, (const u64 *)image, p->n_in, filt_ntt, M * Cs, M, S, Cs, S, 1, Cs, 1, 0, (u64 *)acc, stream);
  if (!rc && !exact) {
    // addRandomMask :761-788 (mod-switch, pk-encryption of zero without e0, RNS mask), truncation
    // :698-703 and removeUnusedCoeffs :800-851 as one fused tail; the share is the mask's scale-and-round
    if (draw_zero) d_zero_enc = (const u64 *)dr_zero;   // made by the front launch: the tail is its one-thread-per-coefficient half
    rc = d_zero_enc
             ? reshare_tail_online(ctx, (u64 *)acc, !direct, p->n_out, d_zero_enc, d_e1, d_mask, p->d_keep, S, d_out_cts,
                                   s, p->d_share_dst, (size_t)(p->out_h * p->out_w), d_out_share, sw_acc)
             : reshare_tail_approx_u(ctx, (u64 *)acc, !direct, p->n_out, d_pk_ntt_last, d_u, (const u64 *)u_ntt, d_e1,
                                     d_mask, p->d_keep, S, d_out_cts, s, p->d_share_dst,
                                     (size_t)(p->out_h * p->out_w), d_out_share, sw_acc);
  }
  if (exact) {
    // USE_APPROX_RESHARE=0: uniform noise on c0 first, e0 kept, the mask is a plaintext mod t subtracted
    // by sub_plain and the share is the mask itself (:225-272, :358-380)
    grab(&zero, p->n_out * 2 * N * 8);
    grab(&mask_t, p->n_out * N * 8);
    if (!rc && !direct) rc = ntt_dispatch(ctx, true, (u64 *)acc, p->n_out * 2, 2, 0, s);
    if (!rc) rc = mmh_flood_add(ctx, (u64 *)acc, d_flood, p->n_out, stream);
    if (!rc) rc = mmh_mod_switch_to_last(ctx, (const u64 *)acc, p->n_out, d_out_cts, stream);
    if (!rc) rc = mmh_encrypt_zero_asym(ctx, 1, d_pk_ntt_last, d_u, d_e0, d_e1, p->n_out, (u64 *)zero, stream);
    if (!rc) rc = mmh_ct_add(ctx, 1, d_out_cts, (const u64 *)zero, p->n_out, stream);
    if (!rc) rc = mmh_reduce_mod_t(ctx, d_mask, (u64 *)mask_t, p->n_out * N, stream);
    if (!rc) rc = mmh_add_plain(ctx, 1, d_out_cts, (const u64 *)mask_t, p->n_out, 1, d_out_cts, stream);
    if (!rc)
      rc = launch_gather_targets(ctx, (const u64 *)mask_t, d_out_share, M, p->n_targets, S,
                                 (size_t)(p->out_h * p->out_w), p->d_tgt_coef, p->d_tgt_dst, p->d_tgt_slice, s);
    if (!rc) rc = mmh_truncate_and_zero(ctx, d_out_cts, p->n_out, p->d_keep, S, stream);
  }
  cleanup();
  return rc;
}

extern "C" int mmh_conv_server(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                               const uint64_t *d_filter_pts, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                               const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                               uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(d_filter_pts, MMH_ERR_NULL_POINTER);
  return conv_server_impl(p, d_in_cts, d_share_pts, d_filter_pts, nullptr, d_pk_ntt_last, d_u, d_e1, d_mask,
                          d_out_cts, d_out_share, stream);
}

extern "C" int mmh_conv_server_exact(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                                     const uint64_t *d_filter_pts, const uint64_t *d_pk_ntt_last,
                                     const uint64_t *d_flood, const int64_t *d_u, const int64_t *d_e0,
                                     const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                                     uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(d_filter_pts && d_flood && d_e0, MMH_ERR_NULL_POINTER);
  return conv_server_impl(p, d_in_cts, d_share_pts, d_filter_pts, nullptr, d_pk_ntt_last, d_u, d_e1, d_mask,
                          d_out_cts, d_out_share, stream, d_flood, d_e0);
}

extern "C" int mmh_conv_server_weights(const mmh_conv_plan *p, const uint64_t *d_in_cts,
                                       const uint64_t *d_share_pts, const uint64_t *d_filters,
                                       const uint64_t *d_pk_ntt_last, const int64_t *d_u, const int64_t *d_e1,
                                       const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share,
                                       void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(d_filters, MMH_ERR_NULL_POINTER);
  return conv_server_impl(p, d_in_cts, d_share_pts, nullptr, d_filters, d_pk_ntt_last, d_u, d_e1, d_mask,
                          d_out_cts, d_out_share, stream);
}

extern "C" int mmh_conv_server_tensors(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share,
                                       const uint64_t *d_filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                                       const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                                       uint64_t *d_out_share, uint32_t *d_zero_filter, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_filters, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!p->meta.is_shared_input || d_share, MMH_ERR_DIM_MISMATCH);
  return conv_server_impl(p, d_in_cts, nullptr, nullptr, d_filters, d_pk_ntt_last, d_u, d_e1, d_mask, d_out_cts,
                          d_out_share, stream, nullptr, nullptr, nullptr, p->meta.is_shared_input ? d_share : nullptr,
                          d_zero_filter);
}

extern "C" int mmh_conv_server_tensors_ev(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share,
                                          const uint64_t *d_filters, const uint64_t *d_pk_ntt_last, const int64_t *d_u,
                                          const int64_t *d_e1, const uint64_t *d_mask, uint64_t *d_out_cts,
                                          uint64_t *d_out_share, uint32_t *zero_filter, void *verdict_event,
                                          void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_filters, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!p->meta.is_shared_input || d_share, MMH_ERR_DIM_MISMATCH);
  MMH_REQUIRE(!verdict_event || zero_filter, MMH_ERR_INVALID_ARG);
  return conv_server_impl(p, d_in_cts, nullptr, nullptr, d_filters, d_pk_ntt_last, d_u, d_e1, d_mask, d_out_cts,
                          d_out_share, stream, nullptr, nullptr, nullptr, p->meta.is_shared_input ? d_share : nullptr,
                          zero_filter, nullptr, verdict_event);
}

extern "C" int mmh_conv_server_tensors_sampled(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share,
                                               const uint64_t *d_filters, const uint64_t *d_pk_ntt_last,
                                               const uint32_t key[8], uint64_t stream_id, size_t first_poly,
                                               uint64_t *d_out_cts, uint64_t *d_out_share, uint32_t *zero_filter,
                                               void *verdict_event, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_filters && key, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!p->meta.is_shared_input || d_share, MMH_ERR_DIM_MISMATCH);
  MMH_REQUIRE(!verdict_event || zero_filter, MMH_ERR_INVALID_ARG);
  return conv_server_impl(p, d_in_cts, nullptr, nullptr, d_filters, d_pk_ntt_last, nullptr, nullptr, nullptr, d_out_cts,
                          d_out_share, stream, nullptr, nullptr, nullptr, p->meta.is_shared_input ? d_share : nullptr,
                          zero_filter, nullptr, verdict_event, key, stream_id, first_poly);
}

extern "C" int mmh_conv_filters_destroy(mmh_conv_filters *f) {
  if (!f) return MMH_OK;
  for (void *b : {(void *)f->f_ntt, (void *)f->wp, (void *)f->wl, (void *)f->w8, (void *)f->flag})
    if (b) (void)hipFree(b);
  delete f;
  return MMH_OK;
}

extern "C" int mmh_conv_filters_prepare(const mmh_conv_plan *p, const uint64_t *d_filters, mmh_conv_filters **out,
                                        void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_filters && out, MMH_ERR_NULL_POINTER);
  *out = nullptr;
  const mmh_ctx *ctx = p->ctx;
  MMH_REQUIRE(ctx->L == 2, MMH_ERR_CONFIG);
  hipStream_t s = (hipStream_t)stream;
  mmh_conv_filters *f = new mmh_conv_filters();
  f->plan = p;
  int rc = MMH_OK;
  auto own = [&](void **b, size_t bytes) {
    if (!rc && hipMalloc(b, bytes) != hipSuccess) rc = MMH_ERR_MEMORY;
  };
  if (p->direct_prepared) {
    const PwSizes z = pw_sizes(p);
    own((void **)&f->wp, (size_t)z.Cpad * z.Mpad * 4);
    own((void **)&f->wl, (size_t)z.Cpad * z.Mpad * 16);
    own((void **)&f->w8, z.w8_bytes());
    own((void **)&f->flag, 4);
    if (!rc) rc = pointwise_weights(p, d_filters, nullptr, f->wp, f->wl, f->w8, f->flag, s);
    if (!rc) {  // model-load time: one synchronous read of which weight form the layer needs
      u32 general = 0;
      if (hipMemcpyAsync(&general, f->flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess)
        rc = MMH_ERR_INTERNAL;
      f->form = (general & 1u) ? 1 : 0;
    }
  } else {
    const size_t M = (size_t)p->meta.n_filters, Cs = p->n_cslices;
    own((void **)&f->f_ntt, M * Cs * 2 * ctx->N * 8);
    if (!rc)
      rc = ntt_filter_dispatch(ctx, d_filters, f->f_ntt, M, (int)p->f.c, (int)p->f.h, (int)p->f.w, (int)p->split.win_c,
                               (int)Cs, p->filt_origin, p->split.win_h * p->split.win_w, p->split.win_w, 2, s);
  }
  if (rc) {
    if (rc == MMH_ERR_MEMORY) set_error("mmh_conv_filters_prepare: hipMalloc failed");
    mmh_conv_filters_destroy(f);
    return rc;
  }
  *out = f;
  return MMH_OK;
}

extern "C" int mmh_conv_server_prepared(const mmh_conv_plan *p, const uint64_t *d_in_cts,
                                        const uint64_t *d_share_pts, const mmh_conv_filters *filters,
                                        const uint64_t *d_pk_ntt_last, const int64_t *d_u, const int64_t *d_e1,
                                        const uint64_t *d_mask, uint64_t *d_out_cts, uint64_t *d_out_share,
                                        void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(filters, MMH_ERR_NULL_POINTER);
  return conv_server_impl(p, d_in_cts, d_share_pts, nullptr, nullptr, d_pk_ntt_last, d_u, d_e1, d_mask, d_out_cts,
                          d_out_share, stream, nullptr, nullptr, filters);
}

// mmh_conv_server_prepared from the raw share tensor (encodeImage fused into the call, as _tensors does)
extern "C" int mmh_conv_server_prepared_tensors(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share,
                                                const mmh_conv_filters *filters, const uint64_t *d_pk_ntt_last,
                                                const int64_t *d_u, const int64_t *d_e1, const uint64_t *d_mask,
                                                uint64_t *d_out_cts, uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && filters, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!p->meta.is_shared_input || d_share, MMH_ERR_DIM_MISMATCH);
  return conv_server_impl(p, d_in_cts, nullptr, nullptr, nullptr, d_pk_ntt_last, d_u, d_e1, d_mask, d_out_cts, d_out_share,
                          stream, nullptr, nullptr, filters, p->meta.is_shared_input ? d_share : nullptr);
}

// The online half of a layer whose weights and whose encryptions of zero were prepared ahead of the inputs
// (mmh_conv_filters_prepare, mmh_reshare_zero_prepare with the layer's n_out): same bits as mmh_conv_server_prepared
// with the u the zeros were made from.  d_share_raw (the share tensor as it is) or d_share_pts (its encoding), not both.
extern "C" int mmh_conv_server_online(const mmh_conv_plan *p, const uint64_t *d_in_cts, const uint64_t *d_share_pts,
                                      const uint64_t *d_share_raw, const mmh_conv_filters *filters,
                                      const uint64_t *d_zero, const int64_t *d_e1, const uint64_t *d_mask,
                                      uint64_t *d_out_cts, uint64_t *d_out_share, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(filters && d_zero, MMH_ERR_NULL_POINTER);
  MMH_REQUIRE(!(d_share_pts && d_share_raw), MMH_ERR_INVALID_ARG);
  return conv_server_impl(p, d_in_cts, d_share_pts, nullptr, nullptr, nullptr, nullptr, d_e1, d_mask, d_out_cts,
                          d_out_share, stream, nullptr, nullptr, filters, d_share_raw, nullptr, d_zero);
}

// ---------------------------------------------------------------- compact wire form of the results
// A result ciphertext of conv2DSS holds far fewer bits than its [2][N] words: truncate_for_decryption cleared the
// low bits of every coefficient (hom_conv2d_ss.cc:418-436) and remove_unused_coeffs zeroed c0 outside the kept
// coefficients (:800-851).  The reference leaves the squeezing to SEAL's serializer (Ciphertext::save compresses,
// cheetah-api.cpp:11-29); here the two facts are used directly: per ciphertext N values of c1 with the cleared
// low bits dropped, then the kept coefficients of c0 likewise, each section a bit stream padded to whole words.
namespace mmh {
struct PackGeom {
  int tz1, b1, tz0, b0;        // cleared low bits and the width that is left, c1 and c0
  size_t words1, words0;       // words of the two sections
  __host__ __device__ size_t words() const { return words1 + words0; }
};
static PackGeom pack_geom(const mmh_conv_plan *p) {
  const mmh_ctx *ctx = p->ctx;
  PackGeom g;
  const int qbits = 64 - __builtin_clzll(ctx->q[0]);
  g.tz1 = ctx->bfv.mask1 ? __builtin_ctzll(ctx->bfv.mask1) : 0, g.tz0 = ctx->bfv.mask0 ? __builtin_ctzll(ctx->bfv.mask0) : 0;
  g.b1 = qbits - g.tz1, g.b0 = qbits - g.tz0;
  g.words1 = (ctx->N * (size_t)g.b1 + 63) / 64, g.words0 = (p->max_kept * (size_t)g.b0 + 63) / 64;
  return g;
}
// one output word per thread: the values whose bits [i*b, i*b + b) overlap bits [64w, 64w + 64) of the section
template <class Value>
__device__ __forceinline__ u64 pack_word(size_t w, int b, size_t count, Value value) {
  const size_t lo = w * 64, hi = lo + 64;
  u64 out = 0;
  for (size_t i = lo / b; i * b < hi && i < count; ++i) {
    const u64 v = value(i);
    const long long shift = (long long)(i * b) - (long long)lo;  // where the value's bit 0 lands in this word
    out |= shift >= 0 ? v << shift : v >> -shift;
  }
  return out;
}
__device__ __forceinline__ u64 unpack_value(const u64 *__restrict__ sec, size_t i, int b) {
  const size_t bit = i * b, w = bit / 64;
  const int off = (int)(bit % 64);
  u64 v = sec[w] >> off;
  if (off + b > 64) v |= sec[w + 1] << (64 - off);
  return b == 64 ? v : v & ((u64(1) << b) - 1);
}
__global__ __launch_bounds__(256) void result_pack_kernel(const u64 *__restrict__ cts, u64 *__restrict__ packed,
                                                          size_t n_out, size_t S, int logN, PackGeom g,
                                                          const u32 *__restrict__ kept_idx,
                                                          const u32 *__restrict__ kept_off) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, per = g.words();
  if (idx >= n_out * per) return;
  const size_t k = idx / per, w = idx % per, N = size_t(1) << logN;
  const u64 *c0 = cts + ((k * 2) << logN), *c1 = c0 + N;
  if (w < g.words1) {
    packed[idx] = pack_word(w, g.b1, N, [&](size_t i) { return c1[i] >> g.tz1; });
  } else {
    const u32 first = kept_off[k % S], count = kept_off[k % S + 1] - first;
    packed[idx] = pack_word(w - g.words1, g.b0, count, [&](size_t i) { return c0[kept_idx[first + i]] >> g.tz0; });
  }
}
__global__ __launch_bounds__(256) void result_unpack_kernel(const u64 *__restrict__ packed, u64 *__restrict__ cts,
                                                            size_t n_out, size_t S, int logN, PackGeom g,
                                                            const int *__restrict__ kept_rank) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // over [n_out][2][N]
  if (idx >= (n_out * 2) << logN) return;
  const size_t k = idx >> (logN + 1), i = idx & ((size_t(1) << logN) - 1);
  const bool is_c1 = (idx >> logN) & 1;
  const u64 *sec = packed + k * g.words();
  if (is_c1) {
    cts[idx] = unpack_value(sec, i, g.b1) << g.tz1;
  } else {
    const int r = kept_rank[((k % S) << logN) + i];
    cts[idx] = r < 0 ? 0 : unpack_value(sec + g.words1, (size_t)r, g.b0) << g.tz0;
  }
}
}  // namespace mmh

extern "C" int mmh_conv_result_packed_words(const mmh_conv_plan *p, size_t *words_per_ct) {
  MMH_REQUIRE(p && words_per_ct, MMH_ERR_NULL_POINTER);
  *words_per_ct = pack_geom(p).words();
  return MMH_OK;
}
extern "C" int mmh_conv_result_pack(const mmh_conv_plan *p, const uint64_t *d_cts, uint64_t *d_packed, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_cts && d_packed, MMH_ERR_NULL_POINTER);
  const PackGeom g = pack_geom(p);
  hipLaunchKernelGGL(result_pack_kernel, dim3(nblk(p->n_out * g.words())), dim3(256), 0, (hipStream_t)stream, d_cts,
                     d_packed, p->n_out, p->n_spatial, p->ctx->logN, g, p->d_kept_idx, p->d_kept_off);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}
extern "C" int mmh_conv_result_unpack(const mmh_conv_plan *p, const uint64_t *d_packed, uint64_t *d_cts,
                                      void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_cts && d_packed, MMH_ERR_NULL_POINTER);
  const PackGeom g = pack_geom(p);
  hipLaunchKernelGGL(result_unpack_kernel, dim3(nblk((p->n_out * 2) << p->ctx->logN)), dim3(256), 0,
                     (hipStream_t)stream, d_packed, d_cts, p->n_out, p->n_spatial, p->ctx->logN, g, p->d_kept_rank);
  MMH_HIP(hipGetLastError());
  return MMH_OK;
}

extern "C" int mmh_conv_decrypt(const mmh_conv_plan *p, const uint64_t *d_sk_ntt, const uint64_t *d_cts,
                                uint64_t *d_out, void *stream) {
  MMH_DEVICE_SCOPE(p ? p->ctx : nullptr);
  MMH_REQUIRE(p && d_sk_ntt && d_cts && d_out, MMH_ERR_NULL_POINTER);
  const mmh_ctx *ctx = p->ctx;
  hipStream_t s = (hipStream_t)stream;
  void *pts = nullptr;
  MMH_HIP(scratch_alloc(&pts, (p->n_out << ctx->logN) * 8, s));
  int rc = mmh_decrypt(ctx, 1, d_sk_ntt, d_cts, p->n_out, (u64 *)pts, stream);
  if (!rc)
    rc = launch_gather_targets(ctx, (const u64 *)pts, d_out, (size_t)p->meta.n_filters, p->n_targets, p->n_spatial,
                               (size_t)(p->out_h * p->out_w), p->d_tgt_coef, p->d_tgt_dst, p->d_tgt_slice, s);
  (void)scratch_free(pts, s);
  return rc;
}
