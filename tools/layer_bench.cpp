// Stand-alone driver of ONE convolution layer's server call through the C ABI (mmh_conv_server_tensors, the call
// CheetahLinear::conv2d's server half makes: encodeImage + encodeFilters + conv2DSS, cheetah-api.cpp:417-445), its
// randomness sampled on the device before every call, for profiling runs: rocprofv3 wants the program itself after `--`.
// Build: g++ -O2 tools/layer_bench.cpp -Iinclude -Lopenmm_amd -lmm_hip -Wl,-rpath,'$ORIGIN/../openmm_amd' -o tools/layer_bench
// Usage: layer_bench C H W fh fw M same stride [iters=20] [warm=5]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mm_hip.h"
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mmh_last_error()); return 1; } } while (0)
static uint64_t s_ = 0x4D264D;
static uint64_t next() {
  s_ += 0x9E3779B97F4A7C15ull;
  uint64_t z = s_; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
int main(int argc, char **argv) {
  if (argc < 9) { fprintf(stderr, "usage: layer_bench C H W fh fw M same stride [iters] [warm]\n"); return 2; }
  const long C = atol(argv[1]), H = atol(argv[2]), W = atol(argv[3]), fh = atol(argv[4]), fw = atol(argv[5]), M = atol(argv[6]);
  const int same = atoi(argv[7]), stride = atoi(argv[8]), iters = argc > 9 ? atoi(argv[9]) : 20, warm = argc > 10 ? atoi(argv[10]) : 5;
  const size_t N = 4096;
  const uint64_t t = 1ull << 37;
  int bits[] = {60, 49};
  mmh_ctx *ctx;
  CK(mmh_ctx_create_bits(&ctx, N, bits, 2, t, 0));
  uint64_t q[2];
  CK(mmh_ctx_info(ctx, nullptr, nullptr, q, nullptr, nullptr, nullptr));
  mmh_conv_meta meta{C, H, W, fh, fw, M, same, stride, C != 3};
  mmh_conv_plan *plan;
  CK(mmh_conv_plan_create(ctx, &meta, &plan));
  size_t n_in, n_cs, n_sp, n_out;
  int64_t oshape[3];
  CK(mmh_conv_plan_info(plan, &n_in, &n_cs, &n_sp, &n_out, oshape, nullptr));
  mmh_conv_work wk;
  CK(mmh_conv_plan_work(plan, &wk));
  auto dev = [&](const std::vector<uint64_t> &h, uint64_t **d) {
    void *p;
    if (mmh_malloc(&p, h.size() * 8)) return 1;
    *d = (uint64_t *)p;
    return mmh_memcpy_h2d(p, h.data(), h.size() * 8, nullptr);
  };
  std::vector<uint64_t> cts(n_in * 4 * N), share((size_t)C * H * W), filt((size_t)M * C * fh * fw), pk(2 * N);
  for (size_t i = 0; i < cts.size(); ++i) cts[i] = next() % q[(i / N) & 1];
  for (auto &v : share) v = next() & (t - 1);
  for (auto &v : filt) v = (uint64_t)((int64_t)(next() % 8191) - 4095) | 1, v &= t - 1;
  for (auto &v : pk) v = next() % q[0];
  uint64_t *d_cts, *d_share, *d_filt, *d_pk, *d_out, *d_osh, *d_u, *d_e1, *d_mask;
  CK(dev(cts, &d_cts)); CK(dev(share, &d_share)); CK(dev(filt, &d_filt)); CK(dev(pk, &d_pk));
  void *p;
  CK(mmh_malloc(&p, n_out * 2 * N * 8)); d_out = (uint64_t *)p;
  CK(mmh_malloc(&p, (size_t)(oshape[0] * oshape[1] * oshape[2]) * 8)); d_osh = (uint64_t *)p;
  CK(mmh_malloc(&p, n_out * N * 8)); d_u = (uint64_t *)p;
  CK(mmh_malloc(&p, n_out * N * 8)); d_e1 = (uint64_t *)p;
  CK(mmh_malloc(&p, n_out * N * 8)); d_mask = (uint64_t *)p;
  uint32_t key[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  uint64_t sid = 1;
  auto run = [&]() -> int {
    // MM_LAYER_ARRAYS=1: the randomness as arrays from one sampling launch, then the call; else the call draws it itself
    static const bool arrays = getenv("MM_LAYER_ARRAYS") != nullptr;
    sid += 3;
    if (!arrays)
      return mmh_conv_server_tensors_sampled(plan, d_cts, meta.is_shared_input ? d_share : nullptr, d_filt, d_pk, key, sid, 0,
                                             d_out, d_osh, nullptr, nullptr, nullptr);
    int rc;
    if ((rc = mmh_sample_reshare(ctx, key, sid, n_out, (int64_t *)d_u, (int64_t *)d_e1, d_mask, nullptr))) return rc;
    return mmh_conv_server_tensors(plan, d_cts, meta.is_shared_input ? d_share : nullptr, d_filt, d_pk, (const int64_t *)d_u,
                                   (const int64_t *)d_e1, d_mask, d_out, d_osh, nullptr, nullptr);
  };
  for (int i = 0; i < warm; ++i) CK(run());
  CK(mmh_stream_sync(nullptr));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) CK(run());
  CK(mmh_stream_sync(nullptr));
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
  const double macs = wk.direct_fresh ? (double)M * wk.taps * (wk.live_columns[0] * 16.0 + wk.live_columns[1] * 14.0) : 0;
  printf("conv %ldx%ldx%ld %ldx%ld -> %ld same=%d s%d: %s route, %zu in / %zu out cts, %.1f us per call (sampling in), "
         "%.3g byte-MACs = %.3f P MAC/s\n", C, H, W, fh, fw, M, same, stride, wk.direct_fresh ? "direct" : "transform",
         n_in, n_out, dt * 1e6, macs, macs / dt / 1e15);
  return 0;
}
