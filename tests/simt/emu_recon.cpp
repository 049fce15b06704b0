// tests/simt/emu_recon.cpp -- TEST INFRASTRUCTURE: runs the single-workgroup panel kernels of
// csrc/dhqr_recon.h (unmodified source) on the CPU through the SIMT emulator in fake/hip/hip_runtime.h.
// Driven by tests/test_simt_emulation.py:  emu_recon <op> <variant> <files...>  (raw float64 files).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dhqr_recon.h"

static std::vector<double> rd(const char *path, size_t n) {
  std::vector<double> v(n);
  FILE *f = fopen(path, "rb");
  if (!f || fread(v.data(), sizeof(double), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  fclose(f);
  return v;
}
static void wr(const char *path, const std::vector<double> &v) {
  FILE *f = fopen(path, "wb");
  if (!f || fwrite(v.data(), sizeof(double), v.size(), f) != v.size()) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
  fclose(f);
}

// rig self-test: a deliberately missing barrier between an LDS write and a neighbour's read must be
// reported by ThreadSanitizer (op "racy"), and the same kernel with the barrier must be clean ("sync")
template <bool WITH_BARRIER>
__global__ void k_selftest(double *out) {
  __shared__ double buf[128];
  const int t = threadIdx.x;
  buf[t] = (double)t;
  if (WITH_BARRIER) __syncthreads();
  out[t] = buf[(t + 64) & 127] + wave_sum(1.0);
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const std::string op = argv[1];
  const int variant = atoi(argv[2]);  // kernel generation (one at present)
  (void)variant;
  const size_t NN = RC_N * RC_N;
  if (op == "chol") {  // chol <variant> G Rprev|- want_inv outR outNegX outFlag
    auto G = rd(argv[3], NN);
    std::vector<double> Rprev;
    const bool has_prev = std::string(argv[4]) != "-";
    if (has_prev) Rprev = rd(argv[4], NN);
    const bool want_inv = atoi(argv[5]) != 0;
    std::vector<double> R(NN, -7.0), X(NN, -7.0);
    int flag[2] = {0, 0};
    simt::launch_block(1024, [&] {
      k_chol_inv(G.data(), has_prev ? Rprev.data() : nullptr, R.data(), want_inv ? X.data() : nullptr, flag);
    });
    wr(argv[6], R);
    wr(argv[7], X);
    wr(argv[8], std::vector<double>{(double)flag[0], (double)flag[1]});
  } else if (op == "recon") {  // recon <variant> P R outAlpha outRref outNegMinv
    auto P = rd(argv[3], NN);
    auto R = rd(argv[4], NN);
    std::vector<double> alpha(RC_N, -7.0), Rref(NN, -7.0), negMinv(NN, -7.0);
    simt::launch_block(1024, [&] {
      k_recon_top(P.data(), (int64_t)RC_N, R.data(), alpha.data(), Rref.data(), negMinv.data());
    });
    wr(argv[5], alpha);
    wr(argv[6], Rref);
    wr(argv[7], negMinv);
  } else if (op == "buildt") {  // buildt <variant> S ncols outT outTt
    auto S = rd(argv[3], NN);
    const int ncols = atoi(argv[4]);
    std::vector<double> T(NN, -7.0), Tt(NN, -7.0);
    simt::launch_block(1024, [&] {
      k_build_t(S.data(), ncols, T.data(), Tt.data(), 0.0, nullptr, 0, nullptr, nullptr);
    });
    wr(argv[5], T);
    wr(argv[6], Tt);
  } else if (op == "top") {  // top 0 G P outAlpha outRref outNegMinv outFlag: the fused top-block kernel
    auto G = rd(argv[3], NN);
    auto P = rd(argv[4], NN);
    std::vector<double> alpha(RC_N, -7.0), Rref(NN, -7.0), negMinv(NN, -7.0);
    int flag[2] = {0, 0};
    simt::launch_block(1024, [&] {
      k_panel_top<false>(G.data(), P.data(), (int64_t)RC_N, alpha.data(), Rref.data(), negMinv.data(), flag);
    });
    wr(argv[5], alpha);
    wr(argv[6], Rref);
    wr(argv[7], negMinv);
    wr(argv[8], std::vector<double>{(double)flag[0], (double)flag[1]});
  } else if (op == "decide") {  // decide 0 S tol outT outTt outStat: k_build_t with the acceptance decision
    auto S = rd(argv[3], NN);
    const double tol = atof(argv[4]);
    std::vector<double> T(NN, -7.0), Tt(NN, -7.0);
    int stat[2] = {2147483647, 0};
    double statword = -1.0;
    simt::launch_block(1024, [&] { k_build_t(S.data(), RC_N, T.data(), Tt.data(), tol, stat, 5, &statword, nullptr); });
    wr(argv[5], T);
    wr(argv[6], Tt);
    wr(argv[7], std::vector<double>{(double)stat[0], (double)stat[1], statword});
  } else if (op == "racy" || op == "sync") {  // <op> 0 out
    std::vector<double> out(128, -1.0);
    if (op == "racy") simt::launch_block(128, [&] { k_selftest<false>(out.data()); });
    else simt::launch_block(128, [&] { k_selftest<true>(out.data()); });
    wr(argv[3], out);
  } else {
    return 2;
  }
  return 0;
}
