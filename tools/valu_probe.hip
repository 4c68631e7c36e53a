// Micro-benchmark: sustained 64-bit integer multiply rates on gfx950 VALU, to price the NTT
// butterfly (no vendor figure exists for INT64 multiply).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;

template <int MODE>
__global__ __launch_bounds__(256) void probe(u64 *out, u64 w, u64 wsh, u64 q, int iters) {
  u64 x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = x[i] * w + q;                       // mullo64 + add
      if (MODE == 1) x[i] = __umul64hi(x[i], wsh) + w;          // mulhi64
      if (MODE == 2) x[i] = x[i] * w - __umul64hi(x[i], wsh) * q;  // Shoup lazy mulmod
      if (MODE == 3) {                                          // full forward butterfly pair
        u64 X = x[i], Y = x[(i + 1) & 7];
        X -= (X >= 2 * q) ? 2 * q : 0;
        u64 T = Y * w - __umul64hi(Y, wsh) * q;
        x[i] = X + T;
        x[(i + 1) & 7] = X + 2 * q - T;
      }
      if (MODE == 4) x[i] = (u64)((uint32_t)x[i]) * (uint32_t)w + x[i];  // v_mad_u64_u32
    }
  }
  u64 s = 0;
  for (int i = 0; i < 8; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, double ops_per_iter) {
  const int blocks = 256 * 8, iters = 4096;
  u64 *d;
  hipMalloc(&d, blocks * 256 * 8);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const u64 q = 1152921504606830593ull, w = 116777451583545ull, wsh = (u64)(((unsigned __int128)w << 64) / q);
  probe<MODE><<<blocks, 256>>>(d, w, wsh, q, 16);
  hipEventRecord(a);
  probe<MODE><<<blocks, 256>>>(d, w, wsh, q, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double n = (double)blocks * 256 * iters * ops_per_iter;
  printf("%-28s %8.3f ms  %10.1f Gop/s  (%.2f cycles/op/SIMD at 2.4GHz x 1024 SIMDs, wave64)\n", name, ms,
         n / ms / 1e6, 2.4e9 * 1024 * 64 / (n / (ms / 1e3)));
  hipFree(d);
}
int main() {
  run<0>("mullo64+add", 8);
  run<1>("mulhi64+add", 8);
  run<2>("shoup lazy mulmod", 8);
  run<3>("harvey fwd butterfly", 8);
  run<4>("mad_u64_u32", 8);
  return 0;
}
