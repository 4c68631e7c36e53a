// Micro-benchmark: sustained FP64 VALU rates on gfx950, to price an NTT butterfly for moduli
// below 2^50 done in double precision (exact integer arithmetic through FMA).
// Build: hipcc -O3 --offload-arch=gfx950 tools/f64_probe.hip -o tools/f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void probe(double *out, double w, double q, double qinv, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = (double)(threadIdx.x * 977 + i + blockIdx.x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fma(x[i], w, q);
      if (MODE == 1) x[i] = x[i] * w;
      if (MODE == 2) x[i] = x[i] + w;
      if (MODE == 3) x[i] = __builtin_rint(x[i]) + w;  // rndne + add
      if (MODE == 4) {                                // butterfly: T = w*Y mod q (lazy), X +- T
        const double X = x[i], Y = x[(i + 1) & 7];
        const double h = w * Y;
        const double l = __builtin_fma(w, Y, -h);
        const double c = __builtin_rint(h * qinv);
        const double d = __builtin_fma(-c, q, h);
        const double T = d + l;
        x[i] = X + T;
        x[(i + 1) & 7] = X - T;
      }
      if (MODE == 5) {                                // same with the magic-number rounding
        const double X = x[i], Y = x[(i + 1) & 7];
        const double h = w * Y;
        const double l = __builtin_fma(w, Y, -h);
        const double c = __builtin_fma(h, qinv, 6755399441055744.0) - 6755399441055744.0;
        const double d = __builtin_fma(-c, q, h);
        const double T = d + l;
        x[i] = X + T;
        x[(i + 1) & 7] = X - T;
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, double ops_per_iter) {
  const int blocks = 256 * 8, iters = 4096;
  double *d;
  hipMalloc(&d, blocks * 256 * 8);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const double q = 562949953216513.0, w = 75260022318.0;
  probe<MODE><<<blocks, 256>>>(d, w, q, 1.0 / q, 16);
  hipEventRecord(a);
  probe<MODE><<<blocks, 256>>>(d, w, q, 1.0 / q, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double n = (double)blocks * 256 * iters * ops_per_iter;
  printf("%-28s %8.3f ms  %10.1f Gop/s  (%.2f cycles/op/SIMD at 2.4GHz x 1024 SIMDs, wave64)\n", name, ms,
         n / ms / 1e6, 2.4e9 * 1024 * 64 / (n / (ms / 1e3)));
  hipFree(d);
}
int main() {
  run<0>("fma_f64", 8);
  run<1>("mul_f64", 8);
  run<2>("add_f64", 8);
  run<3>("rndne_f64+add", 8);
  run<4>("f64 butterfly (rndne)", 8);
  run<5>("f64 butterfly (magic)", 8);
  return 0;
}
