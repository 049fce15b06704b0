// tools/top_probe.cpp -- phase clock of k_panel_top (the single-workgroup kernel on every panel's critical chain):
// runs the TIME instantiation alone on a 4096 x 128 random panel and prints the summed shader cycles of wave 0 per
// phase, the kernel's duration and the shader clock during it.  Build: tools/gpu_top_probe.sh.  Not part of the product.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../distributedhouseholderqr.jl_amd/csrc/dhqr_recon.h"

#define HC(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

__global__ void k_clock2(long long *out) {  // {shader cycles, wall ticks} at this instant
  out[0] = clock64();
  out[1] = wall_clock64();
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int rows = 4096, n = RC_N;
  std::vector<double> P((size_t)rows * n), G((size_t)n * n, 0.0);
  unsigned long long s = 88172645463325252ull;
  for (auto &x : P) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    x = (double)(s >> 11) * (1.0 / 9007199254740992.0);
  }
  for (int i = 0; i < n; ++i)
    for (int k = i; k < n; ++k) {
      double acc = 0.0;
      for (int r = 0; r < rows; ++r) acc += P[r + (size_t)i * rows] * P[r + (size_t)k * rows];
      G[i + (size_t)k * n] = G[k + (size_t)i * n] = acc;
    }
  double *dG, *dP, *dal, *dR, *dM;
  int *dflag;
  long long *dclk;
  HC(hipMalloc((void **)&dG, G.size() * 8));
  HC(hipMalloc((void **)&dP, P.size() * 8));
  HC(hipMalloc((void **)&dal, 256 * 8));
  HC(hipMalloc((void **)&dR, G.size() * 8));
  HC(hipMalloc((void **)&dM, G.size() * 8));
  HC(hipMalloc((void **)&dflag, 64));
  HC(hipMalloc((void **)&dclk, 64));
  HC(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
  HC(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
  HC(hipMemset(dflag, 0, 64));
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  for (int pass = 0; pass < 2; ++pass) {  // pass 0: plain kernel, pass 1: TIME instantiation
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HC(hipMemcpyToSymbol(HIP_SYMBOL(g_top_phase), zero, sizeof(zero)));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_panel_top<false>), dim3(1), dim3(1024), 0, 0, dG, dP, (int64_t)rows, dal, dR, dM, dflag);
    HC(hipDeviceSynchronize());
    long long c0[2], c1[2];
    hipLaunchKernelGGL(k_clock2, dim3(1), dim3(1), 0, 0, dclk);
    HC(hipMemcpy(c0, dclk, 16, hipMemcpyDeviceToHost));
    HC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) {
      if (pass == 0) hipLaunchKernelGGL((k_panel_top<false>), dim3(1), dim3(1024), 0, 0, dG, dP, (int64_t)rows, dal, dR, dM, dflag);
      else hipLaunchKernelGGL((k_panel_top<true>), dim3(1), dim3(1024), 0, 0, dG, dP, (int64_t)rows, dal, dR, dM, dflag);
    }
    HC(hipEventRecord(e1, 0));
    hipLaunchKernelGGL(k_clock2, dim3(1), dim3(1), 0, 0, dclk);
    HC(hipMemcpy(c1, dclk, 16, hipMemcpyDeviceToHost));
    float ms = 0.f;
    HC(hipEventElapsedTime(&ms, e0, e1));
    const double mhz = (double)(c1[0] - c0[0]) / ((double)(c1[1] - c0[1]) / 100.0);  // wall clock: 100 MHz
    printf("%s: %.1f us per launch (%d back-to-back launches), shader clock %.0f MHz\n", pass ? "TIME instantiation" : "k_panel_top", ms * 1e3 / reps, reps, mhz);
    if (pass == 1) {
      unsigned long long ph[8];
      HC(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_top_phase), sizeof(ph)));
      const double steps = (double)ph[5];
#ifdef RC5_TIME
      unsigned long long p5[8];
      HC(hipMemcpyFromSymbol(p5, HIP_SYMBOL(g_rc5_phase), sizeof(p5)));
      printf("inverse phases per launch (thread 0, cycles; %d launches incl. warm-up): P0 %.0f, P1 %.0f, P2 %.0f, P3 %.0f\n", reps + 6,
             p5[0] / (double)(reps * 2 + 6), p5[1] / (double)(reps * 2 + 6), p5[2] / (double)(reps * 2 + 6), p5[3] / (double)(reps * 2 + 6));
#endif
      printf("per step (wave 0, cycles): shuffles %.0f, owner work %.0f, barrier wait %.0f, updates %.0f; inverse + stores per launch %.0f cycles; %0.f steps\n",
             ph[0] / steps, ph[1] / steps, ph[2] / steps, ph[3] / steps, (double)ph[4] / reps, steps);
    }
  }
  int flag[2];
  HC(hipMemcpy(flag, dflag, 8, hipMemcpyDeviceToHost));
  std::vector<double> al(128);
  HC(hipMemcpy(al.data(), dal, 128 * 8, hipMemcpyDeviceToHost));
  printf("breakdown flag %d, alpha[0] %.6f (expect -||column 0|| = %.6f)\n", flag[0], al[0], -std::sqrt(G[0]));
  return 0;
}
