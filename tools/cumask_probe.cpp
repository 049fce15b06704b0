// Which CUs does a CU-masked stream use?  hipExtStreamCreateWithCUMask with bit patterns given on the command line; a kernel of
// many short workgroups records (XCC_ID, SE_ID, CU_ID) from the hardware registers; prints the distinct CUs seen per mask and
// the time of a fixed amount of work.  Build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.cpp -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_where(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffffu) | ((xcc & 0xfu) << 16) | (x == 0.5 ? 1u << 31 : 0u);
}
int main(int argc, char **argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, nw = (ncu + 31) / 32;
  printf("CUs %d, mask words %d\n", ncu, nw);
  const int nblk = 8192;
  unsigned *d; CK(hipMalloc(&d, nblk * 4));
  std::vector<unsigned> h(nblk);
  // patterns: "full", "clearN" (lowest N bits cleared), "only:a-b" (bits a..b set), "words:k" (only the first k words set)
  for (int a = 1; a < argc; ++a) {
    std::vector<uint32_t> mask(nw, 0u);
    auto setbit = [&](int i) { mask[i / 32] |= 1u << (i % 32); };
    if (!strcmp(argv[a], "full")) { for (int i = 0; i < ncu; ++i) setbit(i); }
    else if (!strncmp(argv[a], "clear", 5)) { const int n = atoi(argv[a] + 5); for (int i = n; i < ncu; ++i) setbit(i); }
    else if (!strncmp(argv[a], "only:", 5)) { int lo, hi; sscanf(argv[a] + 5, "%d-%d", &lo, &hi); for (int i = lo; i <= hi; ++i) setbit(i); }
    else if (!strncmp(argv[a], "skip:", 5)) { const int k = atoi(argv[a] + 5); for (int i = 0; i < ncu; ++i) if (i != k) setbit(i); }
    hipStream_t s;
    if (!strcmp(argv[a], "plain")) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)nw, mask.data()));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_where, dim3(nblk), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_where, dim3(nblk), dim3(256), 0, s, d, 20000);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus; int perx[16] = {0};
    for (unsigned v : h) { const unsigned key = (v & 0xf0000u) | (v & 0xff00u & 0x7f00u) | 0; cus.insert(((v >> 16) & 0xf) << 16 | ((v >> 8) & 0xff)); (void)key; }
    for (unsigned c : cus) perx[(c >> 16) & 0xf]++;
    printf("%-12s %7.3f ms  distinct (xcc, hw_id[15:8]) = %zu  per xcc:", argv[a], ms, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
