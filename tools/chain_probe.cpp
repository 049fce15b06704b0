#include <hip/hip_runtime.h>
#include <cstdio>
#include "../distributedhouseholderqr.jl_amd/csrc/dhqr_recon.h"
__global__ void k_chain(double *out, long long *cyc, int iters, int mode) {
  double d = 3.0 + threadIdx.x * 1e-3, ajj = 0.7;
  long long t0 = clock64();
  double acc = 0.0;
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) {
      rc_step_scalars s = rc_step_chain(d, ajj);
      d = fma(s.u, 1e-3, 3.0) + s.rinv * 1e-3;
      ajj = fma(s.al, 1e-3, 0.7);
      acc += s.q + s.sg;
    } else if (mode == 1) {  // sqrt_rsqrt only
      double r, ri;
      dhqr_sqrt_rsqrt(d, r, ri);
      d = fma(ri, 1e-3, 3.0) + r * 1e-9;
    } else if (mode == 2) {  // rcp only
      d = fma(dhqr_rcp(d), 1e-3, 3.0);
    } else if (mode == 3) {  // one dependent fma
      d = fma(d, 0.999, 1e-3);
    } else if (mode == 4) {  // raw rsq
      d = fma(__builtin_amdgcn_rsq(d), 1e-3, 3.0);
    } else if (mode == 5) {  // readlane round trip
      d = fma(rc_readlane(d, i & 63), 0.999, 1e-3);
    } else if (mode == 6) {  // shfl round trip
      d = fma(__shfl(d, (i + threadIdx.x) & 63, 64), 0.999, 1e-3);
    }
  }
  long long t1 = clock64();
  out[threadIdx.x] = d + ajj + acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *o; long long *c;
  hipMalloc(&o, 8192); hipMalloc(&c, 64);
  const char *names[] = {"rc_step_chain", "sqrt_rsqrt", "rcp", "fma", "rsq raw + fma", "readlane + fma", "shfl + fma"};
  for (int threads : {64, 1024})
    for (int mode = 0; mode < 7; ++mode) {
      const int iters = 2000;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, o, c, iters, mode);
      hipDeviceSynchronize();
      long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
      printf("threads %4d  %-16s %.1f cycles per iteration\n", threads, names[mode], (double)h / iters);
    }
  return 0;
}
