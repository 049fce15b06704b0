// tests/simt/emu_gemm.cpp -- TEST INFRASTRUCTURE: the FP64-MFMA trailing-update kernels of
// csrc/dhqr_gemm.h (unmodified source) on the CPU SIMT emulator; v_mfma_f64_16x16x4_f64 is emulated
// wave-synchronously with the documented operand maps.  Workgroups are independent and run in sequence.
//   emu_gemm tn <vec> <nbv> <rows> <ncols> <ldv> <ldc> <rps> V C out          out: nsplit x (nbv... ld 128) x ncols
//   emu_gemm nn <vec> <kw>  <rows> <ncols> <ldv> <ldc> <swz> V W Cin Cout     W: ld = kw
//   emu_gemm quad <2> <512> <rows> <ncols> <ldv> <ldc> <swz> V1 V2 W Cin Cout <skip>   four-panel update (k_gemm_nn_quad)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dhqr_gemm.h"

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
static void grid2(int gx, int gy, int threads, F &&kernel) {
  simt::launch_grid(gx, gy, threads, kernel);
}

int main(int argc, char **argv) {
  if (argc < 10) return 2;
  const std::string op = argv[1];
  const int vec = atoi(argv[2]), kparam = atoi(argv[3]);
  const int64_t rows = atoll(argv[4]), ncols = atoll(argv[5]), ldv = atoll(argv[6]), ldc = atoll(argv[7]);
  if (op == "tn") {
    const int64_t rps = atoll(argv[8]);
    auto V = rd(argv[9], (size_t)ldv * 128);
    auto C = rd(argv[10], (size_t)ldc * ncols);
    const int nsplit = (int)((rows + rps - 1) / rps), ntiles = (int)((ncols + 127) / 128);
    const int64_t ostride = 128 * ncols;
    std::vector<double> out((size_t)nsplit * ostride, -7.0);
#define TN(VEC_, NBV_)                                                                                   \
  grid2(ntiles, nsplit, 256, [&] {                                                                       \
    k_gemm_tn<VEC_, 1, NBV_>(V.data(), ldv, C.data(), ldc, 1, (int64_t)0, rows, ncols, rps, out.data(),  \
                             (int64_t)128, ostride);                                                     \
  })
    if (vec == 2 && kparam == 128) TN(2, 128);
    else if (vec == 1 && kparam == 128) TN(1, 128);
    else if (vec == 2 && kparam == 64) TN(2, 64);
    else if (vec == 2 && kparam == 32) TN(2, 32);
    else return 2;
#undef TN
    wr(argv[11], out);
  } else if (op == "tn2") {  // two-panel W = [V_a V_b]' C (k_gemm_tn2): V has 256 columns, out[slab][column][256]
    const int64_t rps = atoll(argv[8]);
    auto V = rd(argv[9], (size_t)ldv * 256);
    auto C = rd(argv[10], (size_t)ldc * ncols);
    const int nsplit = (int)((rows + rps - 1) / rps), ntiles = (int)((ncols + 127) / 128);
    const int64_t ostride = 256 * ncols;
    std::vector<double> out((size_t)nsplit * ostride, -7.0);
    // persistent: fewer workgroups than units, so every workgroup loops (3 is coprime to most unit counts)
    const int wgs = std::max(1, std::min(3, ntiles * nsplit - 1));
    if (vec == 2) grid2(wgs, 1, 512, [&] { k_gemm_tn2<2>(V.data(), ldv, C.data(), ldc, rows, ncols, rps, out.data(), ostride, (int64_t)0, 1, 0); });
    else grid2(wgs, 1, 512, [&] { k_gemm_tn2<1>(V.data(), ldv, C.data(), ldc, rows, ncols, rps, out.data(), ostride, (int64_t)0, 1, 0); });
    wr(argv[11], out);
  } else if (op == "tn2sk") {  // stream-K k_gemm_tn2 + k_reduce_pieces: kparam = row groups R, argv[8] = workgroups G; fine units of 32 rows
    const int R = kparam;
    const int64_t G = atoll(argv[8]), FU = 32;
    auto V = rd(argv[9], (size_t)ldv * 256);
    auto C = rd(argv[10], (size_t)ldc * ncols);
    const int64_t ntiles = (ncols + 127) / 128, S = (rows + FU - 1) / FU, U = ntiles * S, wstride = 256 * ncols;
    const int64_t q = (U + G - 1) / G, Gq = (U + q - 1) / q;
    int P = 0;
    if (R > 1) for (int g = 0; g < R; ++g) P = std::max(P, tn2_sk_pieces(tn2_sk_group_of(g, R, G, S, ntiles, q)));
    else P = (int)((q >= S) ? 2 : (S + q - 1) / q + 1);
    std::vector<double> part((size_t)std::max(R, 1) * P * wstride, -7.0), Y((size_t)wstride, -9.0);
    const int wgs = (int)(R > 1 ? G : Gq);
    if (vec == 2) grid2(wgs, 1, 512, [&] { k_gemm_tn2<2, true>(V.data(), ldv, C.data(), ldc, rows, ncols, FU, part.data(), wstride, q, R, P); });
    else grid2(wgs, 1, 512, [&] { k_gemm_tn2<1, true>(V.data(), ldv, C.data(), ldc, rows, ncols, FU, part.data(), wstride, q, R, P); });
    grid2((int)((wstride + 255) / 256), 1, 256, [&] { k_reduce_pieces(part.data(), S, q, wstride, wstride, Y.data(), R, P, (int64_t)wgs, ntiles); });
    wr(argv[11], Y);
  } else if (op == "quad") {  // four-panel C -= [V1 | V2] W (k_gemm_nn_quad): V2 starts `skip` rows below V1 (same ldv); swz 2 = 64-row tiles
    const int swz = atoi(argv[8]);
    auto V1 = rd(argv[9], (size_t)ldv * 256);
    auto V2 = rd(argv[10], (size_t)ldv * 256);  // row r of V2's storage is row r + skip of the operand
    auto W = rd(argv[11], (size_t)512 * ncols);
    auto C = rd(argv[12], (size_t)ldc * ncols);
    const int64_t skip = atoll(argv[14]);
    const int gx = (int)((rows + 127) / 128), gy = (int)((ncols + 127) / 128);
    int lx = gx, ly = gy;
    if (swz == 1) { lx = (((gx + 7) / 8) * ((gy + 7) / 8) + 7) / 8 * 512; ly = 1; }
    if (swz == 2)
      grid2((int)((rows + 63) / 64), gy, 256, [&] { k_gemm_nn_quad<2, 64>(V1.data(), V2.data() - skip, ldv, skip, W.data(), (int64_t)512, C.data(), ldc, rows, ncols, 0, nullptr, 0); });
    else
      grid2(lx, ly, 256, [&] { k_gemm_nn_quad<2, 128>(V1.data(), V2.data() - skip, ldv, skip, W.data(), (int64_t)512, C.data(), ldc, rows, ncols, swz, nullptr, 0); });
    wr(argv[13], C);
  } else if (op == "nn") {
    const int swz = atoi(argv[8]);
    auto V = rd(argv[9], (size_t)ldv * kparam);
    auto W = rd(argv[10], (size_t)kparam * ncols);
    auto C = rd(argv[11], (size_t)ldc * ncols);
    const int gx = (int)((rows + 127) / 128), gy = (int)((ncols + 127) / 128);
    int lx = gx, ly = gy;
    if (swz) { lx = (((gx + 7) / 8) * ((gy + 7) / 8) + 7) / 8 * 512; ly = 1; }  // the library's 1-D launch
#define NN(VEC_, KW_)                                                                                     \
  grid2(lx, ly, 256, [&] {                                                                                \
    k_gemm_nn_sub<VEC_, KW_>(V.data(), ldv, W.data(), (int64_t)KW_, C.data(), ldc, rows, ncols, swz, nullptr, 0);     \
  })
    if (swz == 2) {  // 64-row tiles (the lane's narrow products): grid (ceil(rows / 64), gy)
      const int g64 = (int)((rows + 63) / 64);
      if (kparam == 128) grid2(g64, gy, 256, [&] { k_gemm_nn_sub<2, 128, false, false, 64>(V.data(), ldv, W.data(), (int64_t)128, C.data(), ldc, rows, ncols, 0, nullptr, 0); });
      else if (kparam == 256) grid2(g64, gy, 256, [&] { k_gemm_nn_sub<2, 256, false, false, 64>(V.data(), ldv, W.data(), (int64_t)256, C.data(), ldc, rows, ncols, 0, nullptr, 0); });
      else return 2;
    } else
    if (vec == 2 && kparam == 128) NN(2, 128);
    else if (vec == 1 && kparam == 128) NN(1, 128);
    else if (vec == 2 && kparam == 256) NN(2, 256);
    else if (vec == 2 && kparam == 64) NN(2, 64);
    else return 2;
#undef NN
    wr(argv[12], C);
  } else {
    return 2;
  }
  return 0;
}
