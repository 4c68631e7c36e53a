// Host-side tensor <-> polynomial geometry for HomConv / HomFC (product code).
// Reproduces the slicing rules of include/gemini/cheetah/{shape_inference.cc,
// sliced_3d_tensor.h, tensor_encoder.cc} as flat descriptor tables that the device kernels
// consume; nothing here touches the GPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace mmh {

struct Dim3 {
  long c = 0, h = 0, w = 0;
  long count() const { return c * h * w; }
  long operator[](int d) const { return d == 0 ? c : (d == 1 ? h : w); }
};

// How one axis of the (padded) tensor is cut into windows of `window` entries that overlap by
// filter-1 (sliced_3d_tensor.h:153-169).
struct AxisCut {
  long extent = 0;   // unpadded length
  long padded = 0;   // extent + padding
  long window = 0;   // slice_strides[d]
  long step = 0;     // distance between window starts
  long count = 0;    // number of windows
  long half_pad = 0; // ceil(padding / 2)
  void init(long extent_, long padding, long window_, long filter, bool spatial) {
    extent = extent_;
    padded = extent_ + padding;
    window = window_;
    step = window_;
    if (spatial && padded > window) {
      step -= filter - 1;
      count = 1 + (padded - window + step - 1) / step;
    } else {
      count = (padded + window - 1) / window;
    }
    half_pad = spatial ? (padding + 1) / 2 : 0;
  }
  long start(long k) const { return k * step; }
  long size(long k) const { return std::min(start(k) + window, padded) - start(k); }
  // window k in unpadded coordinates: first covered entry, and zero margins on both sides
  bool clip(long k, long &first, long &lo, long &hi) const {
    const long p0 = start(k) - half_pad, sz = size(k);
    first = std::max<long>(0, p0);
    const long covered = std::min(extent, p0 + sz) - first;
    if (covered <= 0) return false;
    lo = first - p0;
    hi = (p0 + sz) - (first + covered);
    return true;
  }
};

// shape_inference::Conv2D (shape_inference.cc:54-133)
struct ConvSplit {
  Dim3 padded_shape;   // "tshape": SAME padding applied (channels from the filter)
  Dim3 strided_shape;  // 1x1 kernels with stride > 1 are subsampled (:99-104)
  long pad_h = 0, pad_w = 0;
  long win_c = 0, win_h = 0, win_w = 0;  // slice_strides
  bool ok = false;
};
inline ConvSplit conv_split(long N, Dim3 in, Dim3 f, bool same, long stride) {
  ConvSplit s;
  if (in.c != f.c || f.count() <= 0 || in.count() <= 0 || stride < 1) return s;
  const long per_window = N / (f.h * f.w);
  if (per_window <= 0) return s;
  s.padded_shape = in;
  if (same) s.padded_shape = Dim3{f.c, in.h + f.h - 1, in.w + f.w - 1};
  s.pad_h = s.padded_shape.h - in.h;
  s.pad_w = s.padded_shape.w - in.w;
  if (s.padded_shape.c < f.c || s.padded_shape.h < f.h || s.padded_shape.w < f.w) return s;
  s.strided_shape = s.padded_shape;
  if (f.h == 1) s.strided_shape.h = (s.padded_shape.h + stride - 1) / stride;
  if (f.w == 1) s.strided_shape.w = (s.padded_shape.w + stride - 1) / stride;
  const long H = s.strided_shape.h, W = s.strided_shape.w;
  if (H * W <= N) {
    s.win_c = std::min<long>(std::min<long>(N / (H * W), s.padded_shape.c), per_window);
    s.win_h = H;
    s.win_w = W;
  } else {
    // the reference evaluates these in int/double arithmetic (:117-130); keep the same types
    const double ratio = static_cast<double>(static_cast<int>(H)) / static_cast<int>(W);
    int a = std::min(static_cast<int>(std::sqrt(static_cast<double>(N))), static_cast<int>(H));
    int b = a;
    a = std::min(static_cast<int>(std::sqrt(N * ratio)), static_cast<int>(H));
    b = std::min<int>(static_cast<int>(N / a), static_cast<int>(W));
    if (std::abs(b - a) < 2) a = b = std::min(a, b);
    s.win_c = 1;
    s.win_h = a;
    s.win_w = b;
  }
  s.ok = true;
  return s;
}

// Output coefficients of one spatial window (ConvCoeffIndexCalculator::Get, tensor_encoder.cc:376-425)
struct OutWindow {
  long out_h = 0, out_w = 0;       // result pixels this window contributes
  long first_r = 0, first_c = 0;   // offset of the first stride-aligned pixel inside the window
};
inline OutWindow out_window(const AxisCut &ah, const AxisCut &aw, long kh, long kw, Dim3 f, long stride) {
  OutWindow o;
  o.first_r = (stride - ah.start(kh) % stride) % stride;
  o.first_c = (stride - aw.start(kw) % stride) % stride;
  o.out_h = (ah.size(kh) - f.h + (stride - o.first_r)) / stride;
  o.out_w = (aw.size(kw) - f.w + (stride - o.first_c)) / stride;
  return o;
}

// HomFCSS::getSplit (hom_fc_ss.cc:25-50): block (d0 rows, d1 cols), d0*d1 <= N, minimising
// ceil(cols/d1) + ceil(rows/d0); the first minimum in (d0, d1) lexicographic order wins.
// Same answer as the exhaustive scan, found per d0 with the largest admissible d1... the cost is
// non-increasing in d1, so for each d0 only the smallest d1 reaching the row's minimum matters.
inline void fc_split(long rows, long cols, long N, long &d0, long &d1) {
  size_t best = static_cast<size_t>(-1);
  d0 = d1 = 0;
  for (long a = 1; a <= std::min(N, rows); ++a) {
    const long bmax = std::min<long>(std::min(N, cols), N / a);
    if (bmax < 1) continue;
    const size_t row_cost = (rows + a - 1) / a;
    const size_t min_in = (cols + bmax - 1) / bmax;  // fewest input chunks reachable for this a
    if (row_cost + min_in >= best) continue;
    // smallest b with ceil(cols/b) == min_in
    const long b = (cols + static_cast<long>(min_in) - 1) / static_cast<long>(min_in);
    best = row_cost + min_in;
    d0 = a;
    d1 = b;
  }
}
}  // namespace mmh
