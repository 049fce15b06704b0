// tools/stream_probe.cpp -- what does a read+write stream reach on this box?  Variants of a copy / scale kernel (bytes in flight
// per thread, workgroups per CU, non-temporal accesses, read-only and write-only) over 2 x 4 GiB.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const d2 *__restrict__ in, d2 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(x[u], out + i + u * stride);
      else out[i + u * stride] = x[u];
    }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ in, double *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    d2 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) s += x[u].x + x[u].y;
  }
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(d2 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = (d2){1.0, 2.0};
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  d2 *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](auto &&launch, const char *name, double gb) {
    for (int w = 0; w < 2; ++w) launch();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / 5, gb / (ms / 5 * 1e-3) / 1e9);
  };
  for (int wgs : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    char nm[96];
    snprintf(nm, sizeof nm, "copy U=1            grid %5d", wgs);
    time([&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(wgs), dim3(256), 0, 0, a, b, n); }, nm, 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy U=4            grid %5d", wgs);
    time([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(wgs), dim3(256), 0, 0, a, b, n); }, nm, 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy U=8            grid %5d", wgs);
    time([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(wgs), dim3(256), 0, 0, a, b, n); }, nm, 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy U=4 nontemporal grid %5d", wgs);
    time([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(wgs), dim3(256), 0, 0, a, b, n); }, nm, 2.0 * bytes);
  }
  time([&] { hipLaunchKernelGGL((k_read<4>), dim3(256 * 8), dim3(256), 0, 0, a, (double *)b, n); }, "read only U=4       grid  2048", 1.0 * bytes);
  time([&] { hipLaunchKernelGGL((k_read<8>), dim3(256 * 16), dim3(256), 0, 0, a, (double *)b, n); }, "read only U=8       grid  4096", 1.0 * bytes);
  time([&] { hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, b, n); }, "write only          grid  2048", 1.0 * bytes);
  time([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy device to device", 2.0 * bytes);
  return 0;
}
