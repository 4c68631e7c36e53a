// Stand-alone driver of libmm_hip.so for profiling runs (rocprofv3 wants the program itself
// after `--`).  Build: g++ -O2 tools/ntt_bench.cpp -Iinclude -Lopenmm_amd -lmm_hip -Wl,-rpath,'$ORIGIN/../openmm_amd' -o tools/ntt_bench
// Usage: ntt_bench [logN=12] [polys=4096] [iters=20] [inverse=0] [only_limb=-1]
//   only_limb >= 0: a one-modulus context holding just that limb's prime (per-modulus kernel rates)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mm_hip.h"
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mmh_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
  int logn = argc > 1 ? atoi(argv[1]) : 12;
  size_t polys = argc > 2 ? atol(argv[2]) : 4096;
  int iters = argc > 3 ? atoi(argv[3]) : 20;
  int inverse = argc > 4 ? atoi(argv[4]) : 0;
  int only = argc > 5 ? atoi(argv[5]) : -1;
  size_t N = size_t(1) << logn, L = logn == 12 ? 2 : 4;
  int bits12[] = {60, 49}, bits13[] = {60, 60, 60, 49};
  mmh_ctx *ctx;
  CK(mmh_ctx_create_bits(&ctx, N, logn == 12 ? bits12 : bits13, L, 0, 0));
  uint64_t q[4];
  CK(mmh_ctx_info(ctx, nullptr, nullptr, q, nullptr, nullptr, nullptr));
  if (only >= 0 && (size_t)only < L) {
    const uint64_t one = q[only];
    CK(mmh_ctx_destroy(ctx));
    CK(mmh_ctx_create(&ctx, N, &one, 1, 0, 0));
    q[0] = one, L = 1;
  }
  std::vector<uint64_t> h(polys * L * N);
  uint64_t s = 0x4D264D;
  for (size_t i = 0; i < h.size(); ++i) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    h[i] = z % q[(i / N) % L];
  }
  void *d;
  CK(mmh_malloc(&d, h.size() * 8));
  CK(mmh_memcpy_h2d(d, h.data(), h.size() * 8, nullptr));
  auto run = [&]() { return inverse ? mmh_ntt_inverse(ctx, (uint64_t *)d, polys, L, nullptr) : mmh_ntt_forward(ctx, (uint64_t *)d, polys, L, nullptr); };
  for (int i = 0; i < 3; ++i) CK(run());
  CK(mmh_stream_sync(nullptr));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) CK(run());
  CK(mmh_stream_sync(nullptr));
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
  printf("N=%zu polys=%zu limbs=%zu %s: %.1f us/launch, %.2f M NTT/s, %.1f GB/s algorithmic\n", N, polys, L,
         inverse ? "inverse" : "forward", dt * 1e6, polys * L / dt / 1e6, polys * L * N * 16 / dt / 1e9);
  mmh_free(d);
  mmh_ctx_destroy(ctx);
  return 0;
}
