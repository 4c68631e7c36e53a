// Micro-benchmark: sustained issue rate of v_mfma_i32_16x16x64_i8 / v_mfma_i32_32x32x32_i8 on gfx950, to price the
// byte-digit matrix product of the direct convolution route (conv.hip).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_i8_probe.hip -o tools/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(int *out, int iters, int seed) {
  v4i a = {(int)threadIdx.x * 977 + seed, (int)blockIdx.x * 31 + 7, (int)threadIdx.x ^ 0x5a5a5a5a, seed * 13};
  v4i b = {(int)threadIdx.x * 131 + 3, seed ^ 0x0f0f0f0f, (int)blockIdx.x + 99, (int)threadIdx.x * 7};
  int s = 0;
  if (MODE == 0) {
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{i, i, i, i};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  } else {
    v16i c[4];
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 16; ++k) c[i][k] = i + k;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 16; ++k) s += c[i][k];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int per_iter, double macs, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 8192;
  int *d;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(d, 64, 1);
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(d, iters, 12345);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 4 * iters * per_iter;   // wave-level instructions
  printf("%-34s %d wave/SIMD %8.3f ms  %8.1f ns/instr/SIMD  (%.1f cycles at 2.4 GHz)  %.2f P MAC/s\n", name,
         waves_per_simd, ms, ms * 1e6 / (n / 1024), ms * 1e-3 / (n / 1024) * 2.4e9, n * macs / (ms * 1e-3) / 1e15);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_mfma_i32_16x16x64_i8 (8 acc)", 8, 16.0 * 16 * 64, w);
    run<1>("v_mfma_i32_32x32x32_i8 (4 acc)", 4, 32.0 * 32 * 32, w);
  }
  return 0;
}
