// dhqr_recon.h -- "R first" panel factorisation: the fast path of the blocked driver.
//
// The reference needs one full-column reduction per panel column (norm + dots, src:129, 208): 128
// dependent global synchronisations per 128-column panel.  For a panel P (rows x 128) we instead
//   1. get R of P = QR with two Gram/Cholesky passes (CholeskyQR2): G = P'P, R1 = chol(G),
//      Q1 = P R1^{-1}, G2 = Q1'Q1, R2 = chol(G2), R = R2 R1     -- GEMMs on the MFMA kernels,
//   2. replay the unblocked algorithm ON THE TOP 128 x 128 BLOCK ONLY (k_recon_top): with R known,
//      every quantity the reference derives from a global reduction follows from the pivot row:
//         s_j = |R_jj|, alpha_j = -sign(a_jj) s_j, f_j            (src:129-131)
//         v_j' a_k = (a_jk - R_jk) / v_jj                         (row j of the update, src:209)
//      which gives the upper-triangular M with  P - alpha E - striu(R) = V M,
//   3. obtain all 128 reflectors at once:  V = tril( (P - alpha E) M^{-1} )   -- one more GEMM,
//   4. verify ||v_j||^2 = 2 for every column (diag of V'V, needed for T anyway).
// The result is the SAME factorisation the reference computes (Householder QR is unique given the
// sign rule), to rounding: numpy prototype and GPU tests agree with the oracle to ~5e-15.  For
// numerically rank-deficient panels CholeskyQR loses accuracy (or breaks down); step 4 detects
// that BEFORE anything is written to P and the driver falls back to the column-by-column kernels
// of dhqr_panel.h, so robustness is that of the reference algorithm.
#pragma once
#include "dhqr_common.h"

#define RC_N 128

// =================================================================================================
// The single-workgroup kernels use ONE workgroup barrier per elimination step.  They sit on the critical
// path of every panel (and of the multi-GPU panel chain), where a step is mostly barrier + LDS round trips.
//   * the pivot scalars are broadcast with a wavefront shuffle inside the half-wave that owns the pivot
//     row (thread (ti,tk) = lane (ti&1)*32 + tk of wave ti>>1), not through LDS + barrier;
//   * the row / column staging buffers are double buffered by step parity, so the only barrier of a
//     step is the one between "owners wrote the buffers" and "everybody reads them": a writer of step
//     s+2 has passed barrier s+1, which every thread reaches only after its reads of step s;
//   * the replay folds f into the broadcast row (a -= a_ij * ((a_jk - R_jk)/(a_jj - alpha))), so the
//     column owners need no scalar at all.
// tests/test_simt_emulation.py runs them on the CPU SIMT emulator (ThreadSanitizer build: a missing barrier
// shows up as a data race).  Measured on MI355X against the first generation (two or three barriers per step,
// LDS-broadcast pivots): panel lane of a 32768^2 factorisation 127 -> 108 ms (gpurun_out/smallk_phases_k*.txt).
__device__ __forceinline__ void rc_upper_inverse_regs(const double (&r)[4][4], double (&x)[4][4],
                                                       double *dinv, double *rowbuf2, double *colbuf2) {
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (ti == tk) dinv[ti + 32 * a] = 1.0 / r[a][a];
#pragma unroll
    for (int b = 0; b < 4; ++b) { acc[a][b] = 0.0; x[a][b] = 0.0; }
  }
  __syncthreads();  // dinv visible; also separates the caller's last use of the staging buffers
#pragma unroll
  for (int la = 3; la >= 0; --la)
    for (int lm = 31; lm >= 0; --lm) {
      const int l = la * 32 + lm;
      double *rowbuf = rowbuf2 + (l & 1) * RC_N, *colbuf = colbuf2 + (l & 1) * RC_N;
      if (ti == lm) {  // owners of row l of X
        const double di = dinv[l];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int k = tk + 32 * b;
          const double v = (k >= l) ? ((k == l ? 1.0 : 0.0) - acc[la][b]) * di : 0.0;
          x[la][b] = v;
          rowbuf[k] = v;
        }
      }
      if (tk == lm) {  // owners of column l of R
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int i = ti + 32 * a;
          colbuf[i] = (i < l) ? r[a][la] : 0.0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double ci = colbuf[ti + 32 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(ci, rowbuf[tk + 32 * b], acc[a][b]);
      }
    }
}

// Cholesky G = R'R (upper R) of a 128 x 128 matrix held in registers in the 4 x 4 cyclic layout (thread (ti,tk)
// owns G[ti+32a][tk+32b]); on return the upper triangle of g is R (entries below the diagonal are scratch).
// One barrier per step; rowbuf: 2 x 128 doubles of LDS.  flag[0] = 1 when a pivot is not positive (breakdown).
__device__ __forceinline__ void rc_cholesky_regs(double (&g)[4][4], double *rowbuf, int *__restrict__ flag) {
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31, lane = t & 63;
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
    for (int jm = 0; jm < 32; ++jm) {
      const int j = ja * 32 + jm;
      // G[j][j] lives in thread (jm, jm): lane (lane & 32) + jm of the half-wave that owns row j.
      // Every wave executes the shuffle (full EXEC); only the row owners use the value.
      double d = __shfl(g[ja][ja], (lane & 32) + jm, 64);
      double *rb = rowbuf + (j & 1) * RC_N;
      if (ti == jm) {  // owners of row j: R[j,k] = G[j,k] / r (k > j), R[j,j] = r
        if (!(d > 0.0)) {  // breakdown (or NaN): flag it, keep going with a harmless pivot
          if (tk == 0) flag[0] = 1;
          d = 1.0;
        }
        const double r = sqrt(d), rinv = 1.0 / r;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int k = tk + 32 * b;
          const double x = (k == j) ? r : g[ja][b] * rinv;
          g[ja][b] = x;
          rb[k] = (k > j) ? x : 0.0;
        }
      }
      __syncthreads();
      // Only the blocks that still change and are read later: rows > j live in the cyclic blocks a >= ja, and only the
      // upper triangle of G is ever used (block b >= a).  The skipped products are exact zeros (rb[k] = 0 for k <= j)
      // or land strictly below the diagonal: 10 / 6 / 3 / 1 of the 16 blocks per thread for ja = 0 .. 3.
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a < ja) continue;
        const double ri = rb[ti + 32 * a];  // 0 for rows <= j
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (b >= a) g[a][b] = fma(-ri, rb[tk + 32 * b], g[a][b]);
      }
    }
}

__global__ __launch_bounds__(1024) void k_chol_inv(const double *__restrict__ G,
                                                    const double *__restrict__ Rprev,
                                                    double *__restrict__ Rout,
                                                    double *__restrict__ negXout,
                                                    int *__restrict__ flag) {
  __shared__ double rowbuf[2 * RC_N], colbuf[2 * RC_N], dinv[RC_N];
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31;
  double g[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) g[a][b] = G[(ti + 32 * a) + (tk + 32 * b) * RC_N];

  rc_cholesky_regs(g, rowbuf, flag);
  // registers now hold R in the upper triangle
  if (Rprev) {  // R <- R * Rprev (second CholeskyQR pass)
    __shared__ double Rl[RC_N * (RC_N + 1) / 2];
    auto pidx = [](int i, int l) { return i * RC_N - (i * (i - 1)) / 2 + (l - i); };
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = ti + 32 * a, k = tk + 32 * b;
        if (i <= k) Rl[pidx(i, k)] = g[a][b];
      }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = ti + 32 * a, k = tk + 32 * b;
        double x = 0.0;
        if (i <= k)
          for (int l = i; l <= k; ++l) x = fma(Rl[pidx(i, l)], Rprev[l + k * RC_N], x);
        g[a][b] = x;
      }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = ti + 32 * a, k = tk + 32 * b;
      Rout[i + k * RC_N] = (i <= k) ? g[a][b] : 0.0;
    }
  if (!negXout) return;
  double x[4][4];
  rc_upper_inverse_regs(g, x, dinv, rowbuf, colbuf);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) negXout[(ti + 32 * a) + (tk + 32 * b) * RC_N] = -x[a][b];
}

// =================================================================================================
// Blocked inverse of a 128 x 128 upper-triangular matrix with FIVE workgroup barriers instead of 128.
// The elimination order of the inverse is not prescribed by the reference (T and M^{-1} are our own
// operands), so it can be organised for the machine:
//   P1  the four 32 x 32 diagonal blocks: their 16 x 16 diagonal blocks by ONE LANE PER COLUMN (U x = e_k by back
//       substitution in registers, fully unrolled, the block read from LDS as broadcasts), then one 16^3 merge per block;
//   P2  the two 64 x 64 blocks [[A,B],[0,C]]^{-1} = [[A^{-1}, -A^{-1} B C^{-1}],[0, C^{-1}]]: two 32^3
//       products per block, one output element per thread;
//   P3  the same step once more for the 128 x 128 matrix: two 64^3 products, four elements per thread.
// Input: Mg = global 128 x 128 column-major, strictly upper part used for columns < ncols (others are
// treated as 0), diagonal taken as 1 when `unit` (T^{-1} = I + striu(V'V)) else read from Mg.
// Output: x12[4] (this thread's entries of the upper-right 64 x 64 block in the MFMA tile map: wave w = t >> 6 owns tile
// (w & 3, w >> 2), register g of lane l is row 16 (w & 3) + (l >> 4) + 4 g, column 64 + 16 (w >> 2) + (l & 15)) and the two diagonal 64 x 64 inverses in Xh.  All global reads of Mg
// are finished when the function returns, so the caller may overwrite Mg with the result.
#define RC5_LDD 33
#define RC5_LDH 65
struct rc5_lds {
  double Ud[4][32][RC5_LDD];    // diagonal blocks of the input (strictly upper part)
  double Xh[2][64][RC5_LDH];    // inverses of the two 64 x 64 diagonal blocks, [block][row][col]
  double T1[64][RC5_LDH];       // B * C^{-1} staging
  double dinv[RC_N];
};
// ncols < 0: real embedding of -ncols/2 complex reflectors (dhqr_complex.h): the strict upper part is taken at the
// level of the 2 x 2 blocks [[Re, -Im], [Im, Re]] -- the (2p, 2p+1) entry of a diagonal block is the imaginary part of
// ||v_p||^2 (zero up to rounding) and does not belong to striu(V^H V).
// rs != nullptr (k_panel_top): the matrix is given as D * Mg off the diagonal with D = diag(rs[0..128)) and its diagonal
// in rs[128..256) (LDS): M[j][k] = dl_jk / v_jj is stored unscaled by the elimination loop, whose owners know v_jj only
// after a square root that is kept off their critical chain.
__device__ __forceinline__ double rc5_in(const double *__restrict__ Mg, int i, int k, int ncols, const double *rs = nullptr) {
  const bool upper = (ncols < 0) ? ((i >> 1) < (k >> 1)) : (i < k);
  const int nc = ncols < 0 ? -ncols : ncols;
  const double v = (upper && k < nc) ? Mg[i + k * RC_N] : 0.0;
  return rs ? v * rs[i] : v;
}
#ifdef RC5_TIME
__device__ unsigned long long g_rc5_phase[8];
#define RC5_MARK(q) do { if (threadIdx.x == 0) { const long long now_ = clock64(); g_rc5_phase[q] += (unsigned long long)(now_ - tm_); tm_ = now_; } } while (0)
#else
#define RC5_MARK(q) do { } while (0)
#endif
__device__ __forceinline__ void rc_upper_inverse_blocked(const double *__restrict__ Mg, int ncols, bool unit,
                                                         rc5_lds &L, double (&x12)[4], const double *rs = nullptr) {
  const int t = threadIdx.x;
#ifdef RC5_TIME
  long long tm_ = clock64();
#endif
  // ---- P0: stage the diagonal blocks, 1/diag, clear Xh
  for (int e = t; e < 2 * 64 * RC5_LDH; e += 1024) (&L.Xh[0][0][0])[e] = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = t + 1024 * r, d = e >> 10, i = e & 31, l = (e >> 5) & 31;
    L.Ud[d][i][l] = rc5_in(Mg, 32 * d + i, 32 * d + l, ncols, rs);
  }
  if (t < RC_N) L.dinv[t] = unit ? 1.0 : 1.0 / (rs ? rs[RC_N + t] : Mg[t + t * RC_N]);
  __syncthreads();
  RC5_MARK(0);
  // ---- P1: the four 32 x 32 diagonal blocks.  P1a: their eight 16 x 16 diagonal blocks, ONE LANE PER COLUMN (back
  // substitution in registers: 120 dependent fma instead of the 496 of a 32-column solve); P1b: the upper-right
  // 16 x 16 block of each, X12 = -A^{-1} (B C^{-1}), on the matrix cores by one wave per block -- the D registers of the
  // first product ARE the B fragments of the second (register g of lane (fk, fi) holds row 4 g + fk, column fi), so
  // nothing goes through LDS in between.
  if (t < RC_N) {
    const int d = t >> 5, hb = (t >> 4) & 1, k = t & 15, q = 16 * hb;
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double s = 0.0;
#pragma unroll
      for (int l = i + 1; l < 16; ++l) s = fma(L.Ud[d][q + i][q + l], x[l], s);  // x[l] == 0 for l > k
      x[i] = (i == k) ? L.dinv[32 * d + q + k] : ((i < k) ? -s * L.dinv[32 * d + q + i] : 0.0);
    }
    const int h = d >> 1, o = (d & 1) * 32 + q;
#pragma unroll
    for (int i = 0; i < 16; ++i) L.Xh[h][o + i][o + k] = x[i];
  }
  __syncthreads();
  if (t < 256) {  // waves 0..3: diagonal block d = wave
    const int d = t >> 6, ln = t & 63, pi = ln & 15, pk = ln >> 4, h = d >> 1, o = (d & 1) * 32;
    dhqr_d4 tb = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)  // T = B C^{-1}
      tb = __builtin_amdgcn_mfma_f64_16x16x4f64(L.Ud[d][pi][16 + 4 * ks + pk], L.Xh[h][o + 16 + 4 * ks + pk][o + 16 + pi], tb, 0, 0, 0);
    dhqr_d4 xb = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)  // A^{-1} T
      xb = __builtin_amdgcn_mfma_f64_16x16x4f64(L.Xh[h][o + pi][o + 4 * ks + pk], tb[ks], xb, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) L.Xh[h][o + pk + 4 * g][o + 16 + pi] = -xb[g];  // read by nobody in this phase
  }
  __syncthreads();
  RC5_MARK(1);
  // The off-diagonal blocks the next phases multiply with are read from global memory HERE, once, coalesced (b3: the
  // 64 x 64 block B of P3, parked in registers until P2 is done; the two 32 x 32 blocks B_h of P2 go straight to LDS: the Ud
  // array is free now, Bh[h][i][l] = L.Ud[h][i][l]).  With the loads inside the product loops (data-dependent trip counts,
  // one L2 round trip per iteration) the inverse took 114k of the kernel's 315k cycles (tools/top_probe); loading them
  // before P1 instead would keep 12 more registers alive across its 32-element solution vectors (spills).
  double b3[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = t + 1024 * r;
    b3[r] = rc5_in(Mg, e & 63, 64 + (e >> 6), ncols, rs);
  }
  {
    const int i = t & 31, l = t >> 5;
    L.Ud[0][i][l] = rc5_in(Mg, i, 32 + l, ncols, rs);
    L.Ud[1][i][l] = rc5_in(Mg, 64 + i, 96 + l, ncols, rs);
  }
  __syncthreads();
  // ---- P2 / P3 on the matrix cores.  The triangular products are small dense GEMMs (the zeros below the diagonals are
  // stored zeros): v_mfma_f64_16x16x4_f64, one 16 x 16 output tile per wave, fragments straight from LDS.  The scalar
  // version (one dot product of up to 64 terms per output, two LDS reads per fma) was LDS-bandwidth bound: 49k of the
  // kernel's cycles (tools/top_probe).  Operand maps as in dhqr_gemm.h: A lane (i = l & 15, k = l >> 4), B lane
  // (k = l >> 4, j = l & 15), D register g of lane l = D[(l >> 4) + 4 g][l & 15].
  const int lane = t & 63, w = t >> 6, fi = lane & 15, fk = lane >> 4;
  // ---- P2: upper-right 32 x 32 block of each 64 x 64 diagonal block: X12_h = -A_h^{-1} (B_h C_h^{-1})
  {
    const int h = w >> 2, ti = (w >> 1) & 1, tj = w & 1;  // waves 0..7: one tile of T1a_h each
    if (w < 8) {
      dhqr_d4 acc = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L.Ud[h][16 * ti + fi][4 * ks + fk], L.Xh[h][32 + 4 * ks + fk][32 + 16 * tj + fi],
                                                  acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) L.T1[32 * h + 16 * ti + fk + 4 * g][16 * tj + fi] = acc[g];
    }
    __syncthreads();
    dhqr_d4 acc = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
    if (w < 8) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L.Xh[h][16 * ti + fi][4 * ks + fk], L.T1[32 * h + 4 * ks + fk][16 * tj + fi], acc, 0,
                                                  0, 0);
    }
    if (w < 8) {  // the X12 corners (columns 32..63 of Xh[h]) are read by nobody in this phase
#pragma unroll
      for (int g = 0; g < 4; ++g) L.Xh[h][16 * ti + fk + 4 * g][32 + 16 * tj + fi] = -acc[g];
    }
    // the block B of P3 -> LDS: B3[i][l] at (&L.Ud[0][0][0])[i * RC5_LDH + l] (the B_h copies were last read before the
    // first barrier of this phase)
    double *B3w = &L.Ud[0][0][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = t + 1024 * r;
      B3w[(e & 63) * RC5_LDH + (e >> 6)] = b3[r];
    }
  }
  __syncthreads();  // X12 corners and B visible; T1 free for reuse
  RC5_MARK(2);
  // ---- P3: upper-right 64 x 64 block of the whole matrix: X12 = -A^{-1} (B C^{-1}), A^{-1} = Xh[0], C^{-1} = Xh[1]
  {
    const double *B3 = &L.Ud[0][0][0];
    const int ti = w & 3, tj = w >> 2;  // 16 waves: one tile each
    dhqr_d4 acc = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(B3[(16 * ti + fi) * RC5_LDH + 4 * ks + fk], L.Xh[1][4 * ks + fk][16 * tj + fi], acc, 0, 0,
                                                0);
#pragma unroll
    for (int g = 0; g < 4; ++g) L.T1[16 * ti + fk + 4 * g][16 * tj + fi] = acc[g];
    __syncthreads();
    acc = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L.Xh[0][16 * ti + fi][4 * ks + fk], L.T1[4 * ks + fk][16 * tj + fi], acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) x12[g] = -acc[g];  // X12[16 ti + fk + 4 g][16 tj + fi]
  }
  RC5_MARK(3);
}
// store helper: calls put(i, k, value) for this thread's 16 entries of the full 128 x 128 result
template <typename F>
__device__ __forceinline__ void rc5_emit(const rc5_lds &L, const double (&x12)[4], F &&put) {
  const int t = threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = t + 1024 * r, i = e & 63, j = e >> 6;
    put(i, j, L.Xh[0][i][j]);            // upper-left  (zero below its diagonal)
    put(64 + i, 64 + j, L.Xh[1][i][j]);  // lower-right
    put(16 * ((t >> 6) & 3) + ((t & 63) >> 4) + 4 * r, 64 + 16 * (t >> 8) + (t & 15), x12[r]);  // upper-right (MFMA tile map)
    put(64 + i, j, 0.0);                 // lower-left
  }
}

// stat != nullptr: fused with the panel's acceptance decision (see "device-side commit" below), taken BEFORE the launches
// that commit the panel: one launch less on the critical chain.
__global__ __launch_bounds__(1024) void k_build_t(const double *__restrict__ S, int ncols,
                                                   double *__restrict__ Tout,
                                                   double *__restrict__ Ttout, double tol,
                                                   int *__restrict__ stat, int panel_idx,
                                                   double *__restrict__ statword, double *__restrict__ Tt2) {
  __shared__ rc5_lds L;
  __shared__ int bad[RC_N];
  if (stat != nullptr) {
    const int j = threadIdx.x;
    if (j < RC_N) bad[j] = (fabs(S[j + j * RC_N] - 2.0) <= tol) ? 0 : 1;
    __syncthreads();
    if (j == 0) {
      int any = stat[1];
      for (int q = 0; q < RC_N; ++q) any |= bad[q];
      if (any && stat[0] > panel_idx) stat[0] = panel_idx;
      stat[1] = 0;
      if (statword) *statword = (double)stat[0];
    }
  }
  double x12[4];
  rc_upper_inverse_blocked(S, ncols, true, L, x12);
  rc5_emit(L, x12, [&](int i, int k, double v) {
    Tout[i + k * RC_N] = v;
    Ttout[k + i * RC_N] = v;
    if (Tt2) Tt2[k + i * RC_N] = v;  // the context's copy of T' for a later solve (dhqr_api.hip, "kept T factors")
  });
}

// Replay of the unblocked algorithm on the top 128 x 128 block with R known (see the file header), then
// -M^{-1} by the blocked inverse.  a: top block of P, r: R (upper triangle used, any row signs), both in the
// 4 x 4 cyclic register layout.  Outputs: alpha[128]; Rref = strict upper part of the reference's R (row signs
// fixed so that R_jj = alpha_j), dense; negMinv = -M^{-1}, dense.  M is parked in the negMinv buffer (global,
// L2 resident), inverted from there, and the buffer is overwritten last.
__device__ __forceinline__ void rc_replay_and_invert(double (&a)[4][4], const double (&r)[4][4], double *wrow,
                                                     double *vcol, rc5_lds &L, double *__restrict__ alpha,
                                                     double *__restrict__ Rref, double *__restrict__ negMinv) {
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31, lane = t & 63;
  double mm[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) mm[x][y] = 0.0;
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
    for (int jm = 0; jm < 32; ++jm) {
      const int j = ja * 32 + jm;
      const int src = (lane & 32) + jm;
      const double ajj = __shfl(a[ja][ja], src, 64), rjj = __shfl(r[ja][ja], src, 64);
      double *wr = wrow + (j & 1) * RC_N, *vc = vcol + (j & 1) * RC_N;
      if (ti == jm) {
        const double s = fabs(rjj);                   // src:129 (norm of the updated column)
        const double al = s * dhqr_alphafactor(ajj);  // src:130
        const double q = s * (s + fabs(ajj));         // src:131: f = 1/sqrt(q), v_jj = (a_jj - alpha) f
        const double sq = sqrt(q);
        const double u = 1.0 / (ajj - al);            // = f / v_jj
        const double vinv = sq * u;                   // = 1 / v_jj
        const double sg = (al == 0.0) ? 0.0 : ((al < 0.0) == (rjj < 0.0) ? 1.0 : -1.0);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const int k = tk + 32 * y;
          const double rr = sg * r[ja][y];
          const double dl = (k > j) ? (a[ja][y] - rr) : 0.0;
          wr[k] = dl * u;                                              // f * (v_j' a_k)
          mm[ja][y] = (k > j) ? dl * vinv : (k == j ? sq : 0.0);       // M[j][k] = v_j' a_k, M[j][j] = 1/f_j
          Rref[j + k * RC_N] = (k > j) ? rr : 0.0;
        }
        if (tk == 0) alpha[j] = al;
      }
      if (tk == jm) {  // owners of column j: the unscaled a_ij (i > j); f travels in the row
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int i = ti + 32 * x;
          vc[i] = (i > j) ? a[x][ja] : 0.0;
        }
      }
      __syncthreads();
      // rows / columns <= j are finished (vc[i] = 0, wr[k] = 0 there): only the cyclic blocks x, y >= ja still change
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (x < ja) continue;
        const double vi = vc[ti + 32 * x];
#pragma unroll
        for (int y = 0; y < 4; ++y)
          if (y >= ja) a[x][y] = fma(-vi, wr[tk + 32 * y], a[x][y]);  // src:209, top rows
      }
    }
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) negMinv[(ti + 32 * x) + (tk + 32 * y) * RC_N] = mm[x][y];
  __syncthreads();  // the workgroup's global writes are visible to all of its threads
  double x12[4];
  rc_upper_inverse_blocked(negMinv, RC_N, false, L, x12);
  __syncthreads();  // every read of M is done (the last ones are in P3's first product): overwrite it
  rc5_emit(L, x12, [&](int i, int k, double v) { negMinv[i + k * RC_N] = -v; });
}

// Cholesky and replay in ONE loop (k_panel_top): row j of R is final after Cholesky step j and the replay's step j
// needs nothing else of R, and the thread that owns row j of G owns row j of the top block too -- so step j of both
// runs between the same two barriers: 128 barriers per panel instead of 256, and the latencies of one step's scalar
// chain (sqrt, reciprocal) overlap the other's.  Arithmetic identical to rc_cholesky_regs + rc_replay_and_invert.
// rb / wr / vc: 2 x 128 doubles of LDS each (double buffered by step parity).
// Phase clock of the TIME instantiation (tools/top_probe.cpp only): summed shader cycles of wave 0 per phase --
// [0] pivot shuffles, [1] row / column owners' work (the scalar chain when wave 0 owns the row), [2] wait at the barrier
// (= the owners' chain seen by everybody else), [3] rank-1 updates, [4] blocked inverse + stores, [5] steps.
__device__ unsigned long long g_top_phase[8];
template <bool TIME = false>
__device__ __forceinline__ void rc_chol_replay_invert(double (&g)[4][4], double (&a)[4][4], double *rbuf, double *wrow,
                                                      double *vcol, rc5_lds &L, double *__restrict__ alpha,
                                                      double *__restrict__ Rref, double *__restrict__ negMinv,
                                                      int *__restrict__ flag) {
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31, lane = t & 63;
  double *qu = &L.T1[0][0];  // q_j, u_j of every row (2 x 128 doubles; T1 is not used before P2 of the inverse)
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
    for (int jm = 0; jm < 32; ++jm) {
      const int j = ja * 32 + jm;
      const int src = (lane & 32) + jm;
      long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
      if constexpr (TIME) tq0 = clock64();
      double d = __shfl(g[ja][ja], src, 64);
      const double ajj = __shfl(a[ja][ja], src, 64);
      if constexpr (TIME) {
        asm volatile("" : "+v"(d));
        tq1 = clock64();
      }
      double *rb = rbuf + (j & 1) * RC_N, *wr = wrow + (j & 1) * RC_N, *vc = vcol + (j & 1) * RC_N;
      if (ti == jm) {  // owners of row j of G and of the top block
        if (!(d > 0.0)) {  // breakdown (or NaN): flag it, keep going with a harmless pivot
          if (tk == 0) flag[0] = 1;
          d = 1.0;
        }
        double r, rinv;
        dhqr_sqrt_rsqrt(d, r, rinv);                  // R_jj > 0 and its reciprocal (dhqr_common.h)
        const double al = r * dhqr_alphafactor(ajj);  // src:129-130 with s = |R_jj|
        const double u = dhqr_rcp(ajj - al);          // = f / v_jj
        // src:131: f = 1/sqrt(q), v_jj = (a_jj - alpha) f.  Only M needs sqrt(q) (M[j][j] = 1/f_j, M[j][k] = dl / v_jj),
        // and M is not used before the loop ends: q and u are parked in LDS and the square root leaves the critical chain
        if (tk == 0) {
          qu[j] = r * (r + fabs(ajj));
          qu[RC_N + j] = u;
        }
        const double sg = (al == 0.0) ? 0.0 : (al < 0.0 ? -1.0 : 1.0);  // row sign of the reference's R: R_jj = alpha_j
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const int k = tk + 32 * y;
          const double x = (k == j) ? r : g[ja][y] * rinv;  // R[j,k]
          g[ja][y] = x;
          rb[k] = (k > j) ? x : 0.0;
          const double rr = sg * x;
          const double dl = (k > j) ? (a[ja][y] - rr) : 0.0;
          wr[k] = dl * u;                                              // f * (v_j' a_k)
          negMinv[j + k * RC_N] = dl;                                  // M[j][k] = dl / v_jj: the scale joins in the inverse
          Rref[j + k * RC_N] = (k > j) ? rr : 0.0;
        }
        if (tk == 0) alpha[j] = al;
      }
      if (tk == jm) {  // owners of column j of the top block: the unscaled a_ij (i > j); f travels in the row
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int i = ti + 32 * x;
          vc[i] = (i > j) ? a[x][ja] : 0.0;
        }
      }
      if constexpr (TIME) tq2 = clock64();
      __syncthreads();
      if constexpr (TIME) tq3 = clock64();
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (x < ja) continue;
        const double ri = rb[ti + 32 * x], vi = vc[ti + 32 * x];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          if (y < ja) continue;
          const double rk = rb[tk + 32 * y];
          if (y >= x) g[x][y] = fma(-ri, rk, g[x][y]);
          a[x][y] = fma(-vi, wr[tk + 32 * y], a[x][y]);  // src:209, top rows
        }
      }
      if constexpr (TIME) {
        asm volatile("" : "+v"(a[3][3]), "+v"(g[3][3]));
        const long long tq4 = clock64();
        if (t == 0) {
          g_top_phase[0] += (unsigned long long)(tq1 - tq0);
          g_top_phase[1] += (unsigned long long)(tq2 - tq1);
          g_top_phase[2] += (unsigned long long)(tq3 - tq2);
          g_top_phase[3] += (unsigned long long)(tq4 - tq3);
          g_top_phase[5] += 1ull;
        }
      }
    }
  long long tinv = 0;
  if constexpr (TIME) tinv = clock64();
  __syncthreads();  // qu of the last rows visible
  if (t < RC_N) {  // M[j][j] = 1/f_j = sqrt(q_j); row scale of M: 1 / v_jj = sqrt(q_j) u_j
    const double sq = sqrt(qu[t]);
    qu[t] = sq * qu[RC_N + t];
    qu[RC_N + t] = sq;
  }
  __syncthreads();  // the workgroup's global writes are visible to all of its threads
  double x12[4];
  rc_upper_inverse_blocked(negMinv, RC_N, false, L, x12, qu);
  __syncthreads();  // every read of M is done (the last ones are in P3's first product): overwrite it
  rc5_emit(L, x12, [&](int i, int k, double v) { negMinv[i + k * RC_N] = -v; });
  if constexpr (TIME) {
    __builtin_amdgcn_s_waitcnt(0);
    if (t == 0) g_top_phase[4] += (unsigned long long)(clock64() - tinv);
  }
}

// Replay with R given in global memory (row-split driver: R comes from the all-reduced Gram matrix; second
// CholeskyQR pass).
__global__ __launch_bounds__(1024) void k_recon_top(const double *__restrict__ P, int64_t ldp,
                                                     const double *__restrict__ R,
                                                     double *__restrict__ alpha,
                                                     double *__restrict__ Rref,
                                                     double *__restrict__ negMinv) {
  __shared__ double wrow[2 * RC_N], vcol[2 * RC_N];
  __shared__ rc5_lds L;
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31;
  double a[4][4], r[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int i = ti + 32 * x, k = tk + 32 * y;
      a[x][y] = P[i + (int64_t)k * ldp];
      r[x][y] = R[i + k * RC_N];
    }
  rc_replay_and_invert(a, r, wrow, vcol, L, alpha, Rref, negMinv);
}

// The whole top block of an R-first panel in ONE single-workgroup launch: G = P'P -> R = chol(G) (registers) ->
// replay -> -M^{-1}.  One launch (and one wait for an idle CU under the trailing-update GEMMs) less per panel
// than k_chol_inv + k_recon_top, and R never leaves the registers.
template <bool TIME = false>
__global__ __launch_bounds__(1024) void k_panel_top(const double *__restrict__ G, const double *__restrict__ P,
                                                     int64_t ldp, double *__restrict__ alpha,
                                                     double *__restrict__ Rref, double *__restrict__ negMinv,
                                                     int *__restrict__ flag) {
  __shared__ double rbuf[2 * RC_N], wrow[2 * RC_N], vcol[2 * RC_N];
  __shared__ rc5_lds L;
  const int t = threadIdx.x, ti = t >> 5, tk = t & 31;
  double g[4][4], a[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      g[x][y] = G[(ti + 32 * x) + (tk + 32 * y) * RC_N];
      a[x][y] = P[(ti + 32 * x) + (int64_t)(tk + 32 * y) * ldp];
    }
  rc_chol_replay_invert<TIME>(g, a, rbuf, wrow, vcol, L, alpha, Rref, negMinv, flag);
}

// Vw currently holds P * M^{-1}; finish V = tril((P - alpha E) M^{-1}) on the top 128 rows:
// Vw[i][j] -= alpha_i * Minv[i][j] (i <= j ... only i == row index < 128), and zero above the diagonal.
__global__ __launch_bounds__(256) void k_recon_fix(double *__restrict__ Vw, int64_t ldv,
                                                   const double *__restrict__ alpha,
                                                   const double *__restrict__ negMinv) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // 128 x 128 entries
  if (idx >= RC_N * RC_N) return;
  const int i = idx & (RC_N - 1), j = idx >> 7;
  double x = 0.0;
  if (i >= j) x = Vw[i + (int64_t)j * ldv] + alpha[i] * negMinv[i + j * RC_N];  // - alpha_i * Minv[i][j]
  Vw[i + (int64_t)j * ldv] = x;
}

// ---- device-side commit of the asynchronous panel pipeline ----------------------------------------
// stat (ints): [0] index of the first panel of the running factorisation whose verification failed
// (INT_MAX: none), [1] Cholesky breakdown flag of the panel in flight.  The host never waits for a panel: the
// kernels that write to the matrix carry (stat, epoch) and do nothing once stat[0] <= epoch; the driver reads
// stat[0] once after the last launch and resumes from the failed panel with the robust kernels.
//
// The decision itself is taken by k_build_t (above): panel `panel_idx` is accepted iff every ||v_j||^2 (diag of
// S = V'V) is within tol of 2 (NaN fails) and the Cholesky did not break down; statword (in the panel's broadcast
// buffer) <- stat[0], so the ranks that receive the panel learn of a failure with the data.
// receiver side: adopt the sender's failure index
__global__ void k_adopt_status(const double *__restrict__ statword, int *__restrict__ stat) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double w = *statword;
    if (w < (double)stat[0]) stat[0] = (int)w;
  }
}
// commit: alpha of an accepted panel -> the caller's alpha vector (and the panel buffer); `src` may alias dst2
__global__ __launch_bounds__(128) void k_commit_alpha(const double *__restrict__ src, int w, double *__restrict__ dst1,
                                                      double *__restrict__ dst2, const int *__restrict__ stat,
                                                      int epoch) {
  if (stat != nullptr && stat[0] <= epoch) return;
  const int j = threadIdx.x;
  const double a = (j < w) ? src[j] : 0.0;
  if (dst1 && j < w) dst1[j] = a;
  if (dst2) dst2[j] = a;
}
// zero `nrows` leading rows of `ncols` columns (the rows above a pair's second panel inside the pair operand)
__global__ __launch_bounds__(128) void k_zero_rows(double *__restrict__ X, int64_t ldx, int nrows) {
  for (int r = threadIdx.x; r < nrows; r += blockDim.x) X[r + (int64_t)blockIdx.x * ldx] = 0.0;
}

// commit: strict upper part of the top block <- reference R
__global__ __launch_bounds__(256) void k_recon_write_r(double *__restrict__ P, int64_t ldp,
                                                       const double *__restrict__ Rref,
                                                       const int *__restrict__ stat, int epoch) {
  if (stat != nullptr && stat[0] <= epoch) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= RC_N * RC_N) return;
  const int i = idx & (RC_N - 1), j = idx >> 7;
  if (i < j) P[i + (int64_t)j * ldp] = Rref[idx];
}
// statword <- stat[0] (a host-verified panel publishes the run's status with its operands)
// The three commits of an accepted R-first panel in ONE predicated launch (one launch gap instead of three on the lane's
// chain): reflectors P[r, p] <- Vw[r, p] for r >= p (k_unpack_v), the reference's R above the diagonal of the top block
// (k_recon_write_r), alpha -> the caller's vector and the panel buffer (k_commit_alpha).  grid (x, 128), 256 threads.
__global__ __launch_bounds__(256) void k_commit_panel(double *__restrict__ P, int64_t ldp, int64_t rows,
                                                      const double *__restrict__ Vw, int64_t ldv,
                                                      const double *__restrict__ Rref, const double *alpha_src,
                                                      double *alpha_dst1, double *alpha_dst2,
                                                      const int *__restrict__ stat, int epoch) {
  if (stat != nullptr && stat[0] <= epoch) return;
  const int64_t p = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = p + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) P[r + p * ldp] = Vw[r + p * ldv];
  if (blockIdx.x == 0) {
    for (int64_t r = threadIdx.x; r < p; r += blockDim.x) P[r + p * ldp] = Rref[r + p * RC_N];
    if (threadIdx.x == 0) {
      const double a = alpha_src[p];
      if (alpha_dst1) alpha_dst1[p] = a;
      if (alpha_dst2) alpha_dst2[p] = a;
    }
  }
}
__global__ void k_set_statword(const int *__restrict__ stat, double *__restrict__ statword) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *statword = (double)stat[0];
}
// x[i] += s[i], i < n <= 128
__global__ void k_axpy1(double *__restrict__ x, const double *__restrict__ s, int n) {
  const int i = threadIdx.x;
  if (i < n) x[i] += s[i];
}

