/*
 * mm_geom_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * Restatement of the tensor <-> polynomial geometry of include/gemini/cheetah:
 * shape_inference.cc, sliced_3d_tensor.h, tensor_encoder.cc, hom_fc_ss.cc (getSplit and the
 * FC encoders).  Written as plain element loops that follow the reference step by step,
 * quirks included; citations are relative to /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

static long ceil_div(long a, long b) { return (a + b - 1) / b; }
static long lmin(long a, long b) { return a < b ? a : b; }
static long lmax(long a, long b) { return a > b ? a : b; }

/* shape_inference.cc:28-52: output H,W of the convolution */
int mmo_conv_out_shape(const long ishape[3], const long fshape[3], int same, int stride, long out_hw[2]) {
  long H = ishape[1], W = ishape[2];
  if (same) { H += fshape[1] - 1; W += fshape[2] - 1; }      /* MakeSamePadShape :10-26 */
  long h = (H - fshape[1] + stride) / stride, w = (W - fshape[2] + stride) / stride;
  if (h <= 0 || w <= 0) return -1;
  out_hw[0] = h; out_hw[1] = w;
  return 0;
}

/* shape_inference.cc:54-133 */
int mmo_conv_shape_infer(long N, const long ishape[3], const long fshape[3], int same, int stride,
                         long strided[3], int pads[2], int slice_strides[3]) {
  if (ishape[0] != fshape[0]) return -1;
  if (fshape[0] <= 0 || fshape[1] <= 0 || fshape[2] <= 0) return -1;
  const int Cw = (int)(N / (fshape[1] * fshape[2]));
  if (Cw <= 0) return -1;
  long t[3] = {ishape[0], ishape[1], ishape[2]};
  if (same) { t[0] = fshape[0]; t[1] += fshape[1] - 1; t[2] += fshape[2] - 1; }
  pads[0] = (int)(t[1] - ishape[1]);
  pads[1] = (int)(t[2] - ishape[2]);
  for (int d = 0; d < 3; d++) if (t[d] < fshape[d]) return -1;
  for (int d = 0; d < 3; d++) strided[d] = t[d];
  for (int d = 1; d <= 2; d++)
    if (fshape[d] == 1) strided[d] = (t[d] + stride - 1) / stride;     /* :99-104 */
  const int H = (int)strided[1], W = (int)strided[2];
  const size_t HW = (size_t)H * W;
  if (HW <= (size_t)N) {
    slice_strides[0] = (int)lmin((long)(N / HW), t[0]);
    slice_strides[0] = (int)lmin(slice_strides[0], Cw);
    slice_strides[1] = H;
    slice_strides[2] = W;
  } else {
    double ratio = (double)H / W;                                      /* :117-130 */
    slice_strides[0] = 1;
    slice_strides[1] = (int)lmin((int)sqrt((double)N), H);
    slice_strides[2] = slice_strides[1];
    slice_strides[1] = (int)lmin((int)sqrt((double)N * ratio), H);
    slice_strides[2] = (int)lmin((long)(N / slice_strides[1]), W);
    if (abs(slice_strides[2] - slice_strides[1]) < 2) {
      int m = slice_strides[1] < slice_strides[2] ? slice_strides[1] : slice_strides[2];
      slice_strides[1] = m;
      slice_strides[2] = m;
    }
  }
  return 0;
}

/* Conv2DSliceHelper, sliced_3d_tensor.h:124-250 */
typedef struct {
  long shape[3], padded[3], f[3];
  int strides[3], overlap[3], pads[2], half_pads[3], slices[3];
} slicer;

static int slicer_init(slicer *s, const long shape[3], const long fshape[3], const int strides[3],
                       const int paddings[2]) {
  for (int d = 0; d < 3; d++) { s->shape[d] = shape[d]; s->f[d] = fshape[d]; s->strides[d] = strides[d]; }
  s->pads[0] = paddings[0]; s->pads[1] = paddings[1];
  s->padded[0] = shape[0]; s->padded[1] = shape[1] + paddings[0]; s->padded[2] = shape[2] + paddings[1];
  for (int d = 0; d < 3; d++) if (strides[d] < 1) return -1;
  if (paddings[0] < 0 || paddings[1] < 0) return -1;
  if (strides[1] < fshape[1] || strides[2] < fshape[1]) return -1;     /* :148-149, compares dim 1 twice */
  s->overlap[0] = strides[0];
  s->slices[0] = (int)ceil_div(shape[0], s->overlap[0]);
  for (int d = 1; d <= 2; d++) {
    s->overlap[d] = strides[d];
    if (s->padded[d] > strides[d]) {                                   /* :159-163 */
      s->overlap[d] -= (int)(fshape[d] - 1);
      s->slices[d] = 1 + (int)ceil_div(s->padded[d] - strides[d], s->overlap[d]);
    } else {
      s->slices[d] = (int)ceil_div(s->padded[d], strides[d]);
    }
    s->half_pads[d] = (int)ceil_div(paddings[d - 1], 2);               /* :168 */
  }
  s->half_pads[0] = 0;
  return 0;
}
static void slicer_shape(const slicer *s, const int idx[3], long out[3]) {   /* :191-204 */
  for (int d = 0; d < 3; d++) {
    long start = (long)idx[d] * s->overlap[d];
    long end = lmin(start + s->strides[d], s->padded[d]);
    out[d] = end - start;
  }
}
/* :206-240: where the slice sits in the unpadded tensor and how much zero padding it carries */
static int slicer_clip(const slicer *s, const int idx[3], long sliced[3], int offs[3], int pad_lo[3], int pad_hi[3]) {
  slicer_shape(s, idx, sliced);
  for (int d = 0; d < 3; d++) {
    long p_coord = (long)idx[d] * s->overlap[d] - s->half_pads[d];
    long p_size = sliced[d];
    offs[d] = (int)lmax(0, p_coord);
    long clipped = lmin(s->shape[d], p_coord + p_size) - offs[d];
    if (clipped <= 0) return -1;
    pad_lo[d] = (int)(offs[d] - p_coord);
    pad_hi[d] = (int)((p_coord + p_size) - (offs[d] + clipped));
  }
  return 0;
}

/* TensorEncoder::EncodeImageShare, tensor_encoder.cc:178-244 (+ Strided3DTensor
 * sliced_3d_tensor.h:13-50, SlicedPaddedTensor::operator() :57-73, ImageIndexer :37-67,
 * Encode :246-276, A2HBFV :303-321).  img is C x H x W row-major.  Returns the number of
 * polynomials (slices); polys may be NULL to query it. */
long mmo_conv_encode_image(long N, uint64_t t, const uint64_t *img, const long ishape[3], const long fshape[3],
                           int same, int stride, uint64_t *polys) {
  long strided[3]; int pads[2], ss[3];
  if (mmo_conv_shape_infer(N, ishape, fshape, same, stride, strided, pads, ss)) return -1;
  int s3[3] = {1, 1, 1};
  for (int d = 1; d <= 2; d++) if (strided[d] < ishape[d]) s3[d] = stride;     /* :197-202 */
  long sshape[3];
  for (int d = 0; d < 3; d++) sshape[d] = ceil_div(ishape[d], s3[d]);           /* Strided3DTensor ctor */
  slicer h;
  if (slicer_init(&h, sshape, fshape, ss, pads)) return -1;
  const long n_slices = (long)h.slices[0] * h.slices[1] * h.slices[2];
  if (!polys) return n_slices;
  const long one_channel = (long)h.slices[1] * h.slices[2];
  if ((long)ss[0] * ss[1] * ss[2] > N) return -1;                               /* Encode: n_elt <= N */
  for (long i = 0; i < n_slices; i++) {
    int idx[3] = {(int)(i / one_channel), (int)((i % one_channel) / h.slices[2]), (int)(i % h.slices[2])};
    long sl[3]; int off[3], lo[3], hi[3];
    if (slicer_clip(&h, idx, sl, off, lo, hi)) return -1;
    uint64_t *p = polys + i * N;
    memset(p, 0, (size_t)N * 8);
    for (long c = 0; c < ss[0]; c++)            /* loops run over the MOCK shape = slice strides */
      for (long y = 0; y < ss[1]; y++)
        for (long x = 0; x < ss[2]; x++) {
          uint64_t v = 0;
          /* SlicedPaddedTensor::operator(): no padding on the channel axis */
          if (c < sl[0] && !(y < lo[1] || y + hi[1] >= sl[1]) && !(x < lo[2] || x + hi[2] >= sl[2])) {
            long cc = c + off[0], yy = (y - lo[1]) + off[1], xx = (x - lo[2]) + off[2];
            v = img[(cc * s3[0]) * ishape[1] * ishape[2] + (yy * s3[1]) * ishape[2] + xx * s3[2]];
          }
          p[c * ss[1] * ss[2] + y * ss[2] + x] = v % t;
        }
  }
  return n_slices;
}

/* TensorEncoder::EncodeFilter, tensor_encoder.cc:124-176 (+ FilterIndexer :69-106).
 * filter is C x fh x fw.  Returns the number of channel slices. */
long mmo_conv_encode_filter(long N, uint64_t t, const uint64_t *filter, const long ishape[3], const long fshape[3],
                            int same, int stride, uint64_t *polys) {
  long strided[3]; int pads[2], ss[3];
  if (fshape[1] * fshape[2] > N) return -1;
  if (mmo_conv_shape_infer(N, ishape, fshape, same, stride, strided, pads, ss)) return -1;
  const long Hs = ss[1], Ws = ss[2];
  int fs[3] = {ss[0], (int)fshape[1], (int)fshape[2]};
  int zero[2] = {0, 0};
  slicer h;
  if (slicer_init(&h, fshape, fshape, fs, zero)) return -1;
  const long n_slices = (long)h.slices[0] * h.slices[1] * h.slices[2];
  if (!polys) return n_slices;
  const long hq = fshape[2];                            /* :78 "h" is read from dim 2 */
  const long offset = Hs * Ws;
  const long begin = offset * (ss[0] - 1) + Ws * (hq - 1) + hq - 1;   /* :82-84 with C = mock channels */
  for (long c = 0; c < n_slices; c++) {
    int idx[3] = {(int)c, 0, 0};
    long sl[3]; int off[3], lo[3], hi[3];
    if (slicer_clip(&h, idx, sl, off, lo, hi)) return -1;
    uint64_t *p = polys + c * N;
    memset(p, 0, (size_t)N * 8);
    for (long cc = 0; cc < ss[0]; cc++)
      for (long r = 0; r < fshape[1]; r++)
        for (long col = 0; col < fshape[2]; col++) {
          long ci = begin - cc * offset - r * Ws - col;
          if (ci < 0 || ci >= N) return -1;
          uint64_t v = 0;
          if (cc < sl[0]) v = filter[((cc + off[0]) * fshape[1] + r + off[1]) * fshape[2] + col + off[2]];
          p[ci] = v % t;
        }
  }
  return n_slices;
}

/* ConvCoeffIndexCalculator, tensor_encoder.cc:334-425 */
struct mmo_conv_indexer {
  long N; int stride; long f[3]; int mock[3]; slicer h;
};
mmo_conv_indexer *mmo_conv_indexer_create(long N, const long ishape[3], const long fshape[3], int same, int stride) {
  long strided[3]; int pads[2], ss[3];
  if (mmo_conv_shape_infer(N, ishape, fshape, same, stride, strided, pads, ss)) return NULL;
  int compressed = strided[0] * strided[1] * strided[2] < ishape[0] * ishape[1] * ishape[2];   /* :353-354 */
  mmo_conv_indexer *x = (mmo_conv_indexer *)calloc(1, sizeof(*x));
  x->N = N; x->stride = stride;
  for (int d = 0; d < 3; d++) { x->f[d] = fshape[d]; x->mock[d] = ss[d]; }
  if (slicer_init(&x->h, compressed ? strided : ishape, fshape, ss, pads)) { free(x); return NULL; }
  return x;
}
void mmo_conv_indexer_destroy(mmo_conv_indexer *x) { free(x); }
int mmo_conv_indexer_slices(const mmo_conv_indexer *x, int d) { return d >= 0 && d < 3 ? x->h.slices[d] : -1; }
/* Get({sh,sw}): output slice H,W and (optionally) the coefficient indices, :376-425 */
int mmo_conv_indexer_get(const mmo_conv_indexer *x, int sh, int sw, long out_hw[2], uint64_t *indices) {
  if (sh < 0 || sh >= x->h.slices[1] || sw < 0 || sw >= x->h.slices[2]) return -1;
  int idx[3] = {0, sh, sw};
  long sl[3];
  slicer_shape(&x->h, idx, sl);
  const long Hs = x->mock[1], Ws = x->mock[2], Cs = sl[0];
  const long hq = x->f[2];
  const long O = Hs * Ws * (Cs - 1) + Ws * (hq - 1) + hq - 1;         /* FilterIndexer::index_begin */
  long offs[2]; int coords[2] = {sh, sw};
  for (int d = 1; d <= 2; d++) {
    long start = (long)coords[d - 1] * x->h.overlap[d];              /* slice_start_at */
    long tmp = start % x->stride;
    offs[d - 1] = (x->stride - tmp) % x->stride;
    tmp = sl[d] - x->f[d];
    tmp += (x->stride - offs[d - 1]);
    tmp /= x->stride;
    out_hw[d - 1] = tmp;
  }
  if (indices) {
    uint64_t *dst = indices;
    for (long h = 0; h < out_hw[0]; h++)
      for (long w = 0; w < out_hw[1]; w++) {
        long r = h * x->stride + offs[0], c = w * x->stride + offs[1];
        if (r >= Hs || c >= Ws) return -2;                           /* ImageIndexer bound check :56-59 */
        *dst++ = (uint64_t)(O + r * Ws + c);
      }
  }
  return 0;
}

/* HomFCSS::getSplit, hom_fc_ss.cc:25-50 (first minimum in (d0,d1) scan order wins) */
void mmo_fc_split(long nrows, long ncols, long N, long split[2]) {
  size_t min_cost = (size_t)-1;
  split[0] = split[1] = 0;
  for (long d0 = 1; d0 <= lmin(N, nrows); d0++)
    for (long d1 = 1; d1 <= lmin(N, ncols); d1++) {
      if (d0 * d1 > N) continue;
      size_t cost = (size_t)(ceil_div(ncols, d1) + ceil_div(nrows, d0));
      if (cost < min_cost) { min_cost = cost; split[0] = d0; split[1] = d1; }
    }
}
/* HomFCSS::encodeInputVector hom_fc_ss.cc:275-325 / encryptInputVector :214-273: chunk i as
 * v0 - sum_{j>=1} v_j X^{N-j}.  The reference reuses one tmp buffer and leaves coefficient
 * N-len unwritten (Appendix B), so a stale value of the previous chunk can survive there:
 * reproduced by carrying `tmp` across chunks. */
long mmo_fc_encode_input(long N, uint64_t t, const uint64_t *vec, long len_total, long d1, uint64_t *polys) {
  long nout = ceil_div(len_total, d1);
  if (!polys) return nout;
  uint64_t *tmp = (uint64_t *)calloc((size_t)N, 8);
  for (long i = 0; i < nout; i++) {
    long start = i * d1, end = lmin(len_total, start + d1), len = end - start;
    tmp[0] = vec[start];
    for (long j = 1; j < len; j++) { uint64_t u = vec[start + j]; tmp[N - j] = u > 0 ? t - u : 0; }   /* :308-310 */
    if (len < N) for (long j = 0; j < N - len - 1; j++) tmp[N - 1 - len - j] = 0;                   /* :311-313 */
    for (long j = 0; j < N; j++) polys[i * N + j] = tmp[j] % t;                                     /* vec2PolyBFV */
  }
  free(tmp);
  return nout;
}
/* HomFCSS::encodeWeightMatrix hom_fc_ss.cc:327-392: block (r_blk, c_blk) row-major at r*d1 + c */
long mmo_fc_encode_weights(long N, uint64_t t, const uint64_t *W, long nrows, long ncols, long d0, long d1,
                           uint64_t *polys) {
  long nrb = ceil_div(nrows, d0), ncb = ceil_div(ncols, d1);
  if (!polys) return nrb * ncb;
  for (long rb = 0; rb < nrb; rb++)
    for (long cb = 0; cb < ncb; cb++) {
      uint64_t *p = polys + (rb * ncb + cb) * N;
      memset(p, 0, (size_t)N * 8);
      long r0 = rb * d0, re = lmin(r0 + d0, nrows), c0 = cb * d1, ce = lmin(c0 + d1, ncols);
      for (long r = r0; r < re; r++)
        for (long c = c0; c < ce; c++) p[(r - r0) * d1 + (c - c0)] = W[r * ncols + c] % t;
    }
  return nrb * ncb;
}
