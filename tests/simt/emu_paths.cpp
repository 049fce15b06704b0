// tests/simt/emu_paths.cpp -- TEST INFRASTRUCTURE: the unblocked factorisation / solve kernels
// (csrc/dhqr_rank1.h, dhqr_complex.h, dhqr_solve.h; unmodified source) on the CPU SIMT emulator.
// The launch sequences restate the host loops of csrc/dhqr_api.hip (factor_unblocked_cols,
// dhqr_factor_c64, dhqr_solve_c64, the back substitution of dhqr_solve_f64); workgroups of a launch
// are independent in these kernels, so they are run one after the other.
//   emu_paths f64   <m> <n> <Ain> <Hout> <alphaout> <threads: 256|512|1024|0=generic>
//   emu_paths c64   <m> <n> <Ain> <Hout> <alphaout> <threads: 256|512|1024>
//   emu_paths zsolve <m> <n> <H> <alpha> <b> <xout>
//   emu_paths backsub <m> <n> <H> <alpha> <b> <xout>        (Float64, b already holds Q'b)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dhqr_complex.h"
#include "dhqr_rank1.h"
#include "dhqr_solve.h"

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
template <typename F>
static void grid(int blocks, int threads, F &&body) {
  simt::launch_grid(blocks, 1, threads, body);
}

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  const std::string op = argv[1];
  const int64_t m = atoll(argv[2]), n = atoll(argv[3]);
  if (op == "f64") {
    auto A = rd(argv[4], (size_t)m * n);
    const int T = atoi(argv[7]);
    std::vector<double> alpha(n, 0.0), v0(m + 16, 0.0), v1(m + 16, 0.0);
    double *vb[2] = {v0.data(), v1.data()};
    const bool vec = (m % 2 == 0);
    grid(1, 1024, [&] { k_reflector<1024>(A.data(), m, (int64_t)0, vb[0], alpha.data()); });
    for (int64_t j = 0; j + 1 < n; ++j) {
      const int nupd = (int)(n - (j + 1));
      const double *vc = vb[j & 1];
      double *vn = vb[(j + 1) & 1];
      if (T == 0) {
        if (vec) grid(nupd, 1024, [&] { k_rank1_generic<1024, 2>(A.data(), m, m, j, vc, vn, alpha.data()); });
        else grid(nupd, 1024, [&] { k_rank1_generic<1024, 1>(A.data(), m, m, j, vc, vn, alpha.data()); });
      } else if (T == 256) {  // EPT 8 covers m <= 2048
        if (vec) grid(nupd, 256, [&] { k_rank1_fused<256, 8, 2>(A.data(), m, m, j, vc, vn, alpha.data()); });
        else grid(nupd, 256, [&] { k_rank1_fused<256, 8, 1>(A.data(), m, m, j, vc, vn, alpha.data()); });
      } else {
        return 2;
      }
    }
    wr(argv[5], A);
    wr(argv[6], alpha);
  } else if (op == "c64") {
    auto A = rd(argv[4], (size_t)2 * m * n);
    const int T = atoi(argv[7]);
    std::vector<double> alpha(2 * n, 0.0), v0(2 * (m + 16), 0.0), v1(2 * (m + 16), 0.0);
    double2 *Az = reinterpret_cast<double2 *>(A.data()), *al = reinterpret_cast<double2 *>(alpha.data());
    double2 *vb[2] = {reinterpret_cast<double2 *>(v0.data()), reinterpret_cast<double2 *>(v1.data())};
    grid(1, 1024, [&] { k_zreflector<1024>(Az, m, (int64_t)0, vb[0], al); });
    for (int64_t j = 0; j + 1 < n; ++j) {
      const int nupd = (int)(n - (j + 1));
      const double2 *vc = vb[j & 1];
      double2 *vn = vb[(j + 1) & 1];
      if (T == 256) grid(nupd, 256, [&] { k_zrank1<256>(Az, m, m, j, vc, vn, al); });
      else if (T == 512) grid(nupd, 512, [&] { k_zrank1<512>(Az, m, m, j, vc, vn, al); });
      else grid(nupd, 1024, [&] { k_zrank1<1024>(Az, m, m, j, vc, vn, al); });
    }
    wr(argv[5], A);
    wr(argv[6], alpha);
  } else if (op == "zsolve") {
    auto H = rd(argv[4], (size_t)2 * m * n);
    auto alpha = rd(argv[5], (size_t)2 * n);
    auto b = rd(argv[6], (size_t)2 * m);
    const double2 *Hz = reinterpret_cast<const double2 *>(H.data());
    const double2 *al = reinterpret_cast<const double2 *>(alpha.data());
    double2 *bz = reinterpret_cast<double2 *>(b.data());
    for (int64_t j = 0; j < n; ++j) grid(1, 256, [&] { k_zqtb_col<256>(Hz + j * m, bz, m, j); });
    for (int64_t hi = n; hi > 0; hi -= ZBS_NB) {
      const int64_t lo = hi - ZBS_NB > 0 ? hi - ZBS_NB : 0;
      grid(1, 64, [&] { k_zbacksub_diag(Hz, m, al, bz, lo, hi); });
      if (lo > 0) grid((int)((lo + 255) / 256), 256, [&] { k_zbacksub_update(Hz, m, bz, lo, hi); });
    }
    b.resize(2 * n);
    wr(argv[7], b);
  } else if (op == "backsub") {
    auto H = rd(argv[4], (size_t)m * n);
    auto alpha = rd(argv[5], (size_t)n);
    auto b = rd(argv[6], (size_t)m);
    for (int64_t hi = n; hi > 0; hi -= BS_NB) {
      const int64_t lo = hi - BS_NB > 0 ? hi - BS_NB : 0;
      grid(1, 64, [&] { k_backsub_diag(H.data(), m, alpha.data(), b.data(), lo, hi); });
      if (lo > 0) grid((int)((lo + 255) / 256), 256, [&] { k_backsub_update(H.data(), m, b.data(), lo, hi); });
    }
    b.resize(n);
    wr(argv[7], b);
  } else {
    return 2;
  }
  return 0;
}
