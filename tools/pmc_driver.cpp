// Minimal torch-free driver of the C ABI for rocprofv3 --pmc passes (rocprofv3 counter collection
// crashes under the python/torch process on this image).  usage: pmc_driver <n> <nb> [lookahead]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../include/dhqr.h"
int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 8192;
  const int nb = argc > 2 ? atoi(argv[2]) : 0;
  dhqr_ctx *ctx = nullptr;
  if (dhqr_create(&ctx, 0) != DHQR_OK) { fprintf(stderr, "%s\n", dhqr_last_error()); return 1; }
  double *A = nullptr, *al = nullptr;
  hipMalloc((void **)&A, (size_t)n * n * 8);
  hipMalloc((void **)&al, (size_t)n * 8);
  dhqr_fill_uniform_f64(ctx, A, n, n, n, 0, n, 0, 128, 1, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dhqr_synchronize(ctx);
  if (dhqr_factor_f64(ctx, A, n, n, n, al, nb) != DHQR_OK) { fprintf(stderr, "%s\n", dhqr_last_error()); return 1; }
  dhqr_synchronize(ctx);
  printf("done n=%lld nb=%d\n", (long long)n, nb);
  return 0;
}
