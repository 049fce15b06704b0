// tools/pmc_driver.cpp -- torch-free driver for hardware-counter runs (rocprofv3 --pmc ...): a handful of launches
// with KNOWN byte counts (the streaming-copy micro-benchmark: 2 x `bytes` per launch) followed by one factorisation.
//   pmc_driver stream <MiB> | unblocked <n> | blocked <n> | gemm <kind 0 NN256 / 1 TN2> <rows> <ncols> <reps>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../include/dhqr.h"

#define CK(x)                                                              \
  do {                                                                     \
    int rc_ = (x);                                                         \
    if (rc_ != 0) {                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, dhqr_last_error());      \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  dhqr_ctx *c = nullptr;
  CK(dhqr_create(&c, 0));
  const long v = atol(argv[2]);
  if (!strcmp(argv[1], "gemm")) {
    if (argc < 6) return 2;
    double out[4] = {0, 0, 0, 0};
    CK(dhqr_bench_gemm_f64(c, (int32_t)v, atol(argv[3]), atol(argv[4]), (int32_t)atol(argv[5]), out));
    printf("gemm kind %ld %sx%s: %.3f ms/launch, %.2f TFLOP/s, shader clock %.0f MHz\n", v, argv[3], argv[4], out[0], out[1], out[2]);
  } else if (!strcmp(argv[1], "stream")) {
    double gbps = 0;
    CK(dhqr_bench_stream_f64(c, (int64_t)v << 20, &gbps));
    printf("stream %ld MiB: %.1f GB/s (6 launches of k_stream_bench, each reads and writes %ld MiB)\n", v, gbps, v);
  } else {
    const int64_t n = v;
    const int nb = !strcmp(argv[1], "blocked") ? 128 : 0;
    double *A = nullptr, *al = nullptr;
    if (hipMalloc((void **)&A, (size_t)n * n * 8) != hipSuccess || hipMalloc((void **)&al, (size_t)n * 8) != hipSuccess) return 3;
    CK(dhqr_fill_uniform_f64(c, A, n, n, n, 0, n, 0, 128, 1, 0));
    CK(dhqr_factor_f64(c, A, n, n, n, al, nb));
    CK(dhqr_synchronize(c));
    printf("%s %ld done\n", argv[1], (long)n);
  }
  CK(dhqr_destroy(c));
  return 0;
}
