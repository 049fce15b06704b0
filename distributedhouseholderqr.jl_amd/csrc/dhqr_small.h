// dhqr_small.h -- qr!(A) and H \ b of a SMALL matrix, each in ONE single-workgroup launch.
//
// The shapes the reference's own test file times (test/runtests.jl:42: n = 100, 200, ... with m = 1.1 n, ratio against
// LAPACK printed at :87-89) start far below anything a blocked driver is made for: at 110 x 100 the look-ahead driver's
// group buffers, status read-backs and ~250 launches cost 1.2 ms where LAPACK needs 0.35 ms.  A matrix of up to
// 224 x 224 doubles (392 KB) fits into the REGISTERS of one compute unit (512 KB), so the reference's algorithm
// (src:122-148, 198-213) runs here exactly as written -- one reflector after the other, every trailing column updated by
// every reflector -- with the whole matrix resident in VGPRs and ONE workgroup barrier per column:
//
//   k_small_qr<NR, NQ>   512 threads = 8 waves (two per SIMD, 256 registers per lane).  Lane (rg, cs) of wave w holds
//                        rows rg + 16 r (r < NR) of the columns 32 q + 4 w + cs (q < NQ): a 16-lane DPP row spans 16
//                        consecutive matrix rows of one column, the four rows of a wave are four adjacent columns.  The dot
//                        product v_j' a_c (partialdot, src:42-49) is NR fma per lane + a 4-step DPP reduction inside the
//                        16-lane row (VALU only, four columns per instruction); the update (hotloop!, src:156-160) NR fma.
//                        The wave that owns column j + 1 updates it FIRST and builds reflector j + 1 (norm in
//                        double-double like every other path, src:129-135) while the other waves are still applying
//                        reflector j: the reflector chain overlaps the trailing update, v travels through 2 x 16 NR
//                        doubles of LDS (double buffered by column parity).
//   k_small_ldiv         b <- Q'b (src:215-224) and the back substitution (src:244-254) in one launch: wave 0 keeps b in
//                        registers and walks the columns, waves 1-3 stream the factor in 16-column chunks into a
//                        double-buffered LDS stage ahead of it (once left to right for Q'b, once right to left -- upper
//                        triangle only -- for R).
//
// Both kernels take plain pointers that may be device memory OR pinned host memory: the host-array entry points
// (dhqr_qr_f64 / dhqr_ldiv_f64) hand them their pinned staging buffer, so a call is memcpy -> one launch -> one
// synchronisation -> memcpy, with no hipMemcpy on the path at all (the kernel's first loads / last stores cross PCIe
// themselves: 88 KB at 110 x 100).
#pragma once
#include <type_traits>
#include "dhqr_common.h"

#define SMQ_THREADS 512  // k_small_qr: 8 waves, two per SIMD, 256 registers per lane
#define SMQ_GW 32        // columns per group: 8 waves x 4 DPP rows
#define SML_THREADS 256  // k_small_ldiv

// sum over the 16 lanes of a DPP row; every lane of the row ends with the row's total
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x124, 0xf>(v);  // row_ror:4
  v += dpp_f64<0x128, 0xf>(v);  // row_ror:8
  return v;
}
// Cascaded (unnormalised) double-double sums: hi carries the running sum, lo collects the exact rounding errors of every
// addition (TwoSum) and whatever low parts come in; hi + lo is the sum as if accumulated in twice the working precision
// (Ogita, Rump, Oishi: Sum2).  Unlike dd_add there is no renormalisation, so the DEPENDENT chain of a reduction step is one
// DPP move and one addition; the error terms are side chains the FP64 pipe fills its idle slots with.  Both partners of a
// symmetric exchange compute identical values (s and its exact error are symmetric in the operands).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ dhqr_dd dd_cascade_step(const dhqr_dd v) {
  const double ohi = dpp_f64<CTRL, ROW_MASK>(v.hi), olo = dpp_f64<CTRL, ROW_MASK>(v.lo);
  dhqr_dd r;
  double e;
  dd_two_sum(v.hi, ohi, r.hi, e);
  r.lo = (v.lo + olo) + e;
  return r;
}
__device__ __forceinline__ dhqr_dd row16_sum_dd(dhqr_dd v) {  // every lane of a 16-lane row ends with the row's total
  v = dd_cascade_step<0xB1, 0xf>(v);
  v = dd_cascade_step<0x4E, 0xf>(v);
  v = dd_cascade_step<0x124, 0xf>(v);
  v = dd_cascade_step<0x128, 0xf>(v);
  return v;
}
__device__ __forceinline__ double smq_readlane(double v, int lane) {  // lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// wave sum of a double-double (lane 63's total, broadcast): the DPP pattern of wave_sum_dpp (dhqr_common.h)
__device__ __forceinline__ dhqr_dd wave_sum_dd(dhqr_dd v) {
  v = row16_sum_dd(v);
  v = dd_cascade_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3 (lanes without a source add zero)
  v = dd_cascade_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  dhqr_dd r;
  r.hi = smq_readlane(v.hi, 63);
  r.lo = smq_readlane(v.lo, 63);
  return r;
}

// One pass of reflector j (vr: its rows rg + 16 r, zeros above row j) over the column groups of this wave, rows r >= R0
// only (the caller knows that every row below 16 R0 lies above j).  Straight-line code per block of SMQ_GB groups -- the
// independent dot products of a block keep the FP64 pipe busy while a DPP reduction's dependent steps are in flight --
// and ONE uniform branch per block skips the blocks that are finished.
// qskip: a group this wave has already updated (the look-ahead column's), masked out through its coefficient.
#define SMQ_GB 4
template <int NR, int NQ, int R0>
__device__ __forceinline__ void smq_pass(double (&a)[NQ][NR], const double (&vr)[NR], int j, int cbase, int qskip) {
#pragma unroll
  for (int qb = 0; qb < NQ; qb += SMQ_GB) {
    if (SMQ_GW * (qb + SMQ_GB) - 1 > j) {
#pragma unroll
      for (int q = qb; q < qb + SMQ_GB; ++q) {
        if (q < NQ) {  // (compile time)
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int r = R0; r < NR; r += 2) {
            p0 = fma(vr[r], a[q][r], p0);  // src:42-49
            if (r + 1 < NR) p1 = fma(vr[r + 1], a[q][r + 1], p1);
          }
          const double p = row16_sum(p0 + p1);
          const double d = (SMQ_GW * q + cbase > j && q != qskip) ? p : 0.0;  // columns <= j are finished, columns >= n hold zeros
#pragma unroll
          for (int r = R0; r < NR; ++r) a[q][r] = fma(-vr[r], d, a[q][r]);  // src:156-160, src:209
        }
      }
    }
  }
}

// householder!(A, alpha) (src:113, 122-148, 198-213) for m <= 16 NR, n <= 32 NQ, m >= n.  Asrc / Adst may alias.
template <int NR, int NQ>
__global__ __launch_bounds__(SMQ_THREADS) void k_small_qr(const double *Asrc, int64_t lds, double *Adst, int64_t ldd, int m,
                                                          int n, double *__restrict__ alpha) {
  __shared__ double vb[2][16 * NR];
  __shared__ double als[SMQ_GW * NQ];  // alpha leaves in one piece at the end: a store per column to (possibly host) memory
                                       // would put a PCIe round trip in front of every barrier (measured: 2.4 us per column)
  const int t = threadIdx.x, w = t >> 6, l = t & 63, rg = l & 15, cs = l >> 4;
  const int cbase = 4 * w + cs;  // this lane's column of group q: 32 q + cbase
  double a[NQ][NR];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = rg + 16 * r, col = SMQ_GW * q + cbase;
      a[q][r] = 0.0;
      if (row < m && col < n) a[q][r] = Asrc[(int64_t)row + (int64_t)col * lds];
    }

  // Reflector of column jn from its updated entries (src:129-135); executed by every lane of the wave that holds it, on
  // the matrix registers in place (the column sits in the 16 lanes cs == cs1; the other three rows of the wave compute
  // along on their own columns and discard the result).  Row blocks above the pivot's are skipped (uniform branches).
  auto build = [&](int jn) __attribute__((always_inline)) {
    const int q1 = jn / SMQ_GW, r1 = jn >> 4, cs1 = jn & 3, srcl = (cs1 << 4) | (jn & 15);
    dhqr_dd acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
    double hc = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q == q1) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
          if (r >= r1) {
            const int row = rg + 16 * r;
            const double xv = (row >= jn) ? a[q][r] : 0.0;  // rows >= m hold zeros
            if (row == jn) hc = a[q][r];
            if (r & 1) dd_add_sq(acc1, xv);
            else dd_add_sq(acc0, xv);
          }
      }
    const double h = smq_readlane(hc, srcl);
    acc0.lo += acc1.lo;
    {
      double e;
      dd_two_sum(acc0.hi, acc1.hi, acc0.hi, e);
      acc0.lo += e;
    }
    const dhqr_dd ss = row16_sum_dd(acc0);
    const double s2 = smq_readlane(ss.hi + ss.lo, srcl);
    double sn, f;
    if (s2 > 0.0 && s2 < 1e300) {  // (uniform) the refinement chains of dhqr_common.h: half the dependent instructions
      double rinv, sq;
      dhqr_sqrt_rsqrt(s2, sn, rinv);                // src:129
      dhqr_sqrt_rsqrt(sn * (sn + fabs(h)), sq, f);  // src:131
    } else {
      sn = sqrt(s2);
      f = 1.0 / sqrt(sn * (sn + fabs(h)));
    }
    const double al = sn * dhqr_alphafactor(h);     // src:130
    double *vo = vb[jn & 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q == q1) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = rg + 16 * r;
          if (r >= r1) {
            const double xv = a[q][r];
            const double val = row > jn ? xv * f : (row == jn ? (h - al) * f : xv);  // src:132-135
            if (cs == cs1) {
              a[q][r] = val;
              vo[row] = row >= jn ? val : 0.0;  // src:138-140 (Hj)
            }
          } else if (cs == cs1) {
            vo[row] = 0.0;
          }
        }
      }
    if (l == srcl) als[jn] = al;
  };

  // one column step; R0: rows below 16 R0 lie above row j (compile time: the loop below runs in four phases, a quarter of
  // the rows apart -- four code versions inside ONE loop body made the register allocator spill ~1000 registers at the
  // merge, four loops one after the other do not)
  auto step = [&](auto r0c, int j) __attribute__((always_inline)) {
    constexpr int R0 = decltype(r0c)::value;
    double vr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) vr[r] = (r >= R0) ? vb[j & 1][rg + 16 * r] : 0.0;
    const int jn = j + 1, qn = jn / SMQ_GW;
    int qskip = -1;
    if (w == ((jn & (SMQ_GW - 1)) >> 2)) {
      // the wave that holds column j + 1 updates that column's group first and builds the next reflector while the
      // other waves are still applying this one
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q == qn) {
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int r = R0; r < NR; r += 2) {
            p0 = fma(vr[r], a[q][r], p0);
            if (r + 1 < NR) p1 = fma(vr[r + 1], a[q][r + 1], p1);
          }
          const double p = row16_sum(p0 + p1);
          const double d = (SMQ_GW * q + cbase > j) ? p : 0.0;
#pragma unroll
          for (int r = R0; r < NR; ++r) a[q][r] = fma(-vr[r], d, a[q][r]);
        }
      build(jn);
      qskip = qn;
    }
    smq_pass<NR, NQ, R0>(a, vr, j, cbase, qskip);
    __syncthreads();
  };
  if (w == 0) build(0);
  __syncthreads();
  {
    constexpr int RA = NR / 4, RB = NR / 2, RC = (3 * NR) / 4;
    int j = 0;
    for (; j + 1 < n && j < 16 * RA; ++j) step(std::integral_constant<int, 0>{}, j);
    for (; j + 1 < n && j < 16 * RB; ++j) step(std::integral_constant<int, RA>{}, j);
    for (; j + 1 < n && j < 16 * RC; ++j) step(std::integral_constant<int, RB>{}, j);
    for (; j + 1 < n; ++j) step(std::integral_constant<int, RC>{}, j);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = rg + 16 * r, col = SMQ_GW * q + cbase;
      if (row < m && col < n) Adst[(int64_t)row + (int64_t)col * ldd] = a[q][r];
    }
  for (int i = t; i < n; i += SMQ_THREADS) alpha[i] = als[i];
}

// solve_householder!(b, H, alpha) (src:284-294) for m <= 256: b (m) <- [x; tail of Q'b], xout (n, may be nullptr) <- x.
// b is carried in DOUBLE-DOUBLE through Q'b and the back substitution, in the reference's operation order (src:215-224
// reflectors in column order, src:244-254 from the last row up), x rounded to double once per entry -- what the
// ComplexF64 solve does (dhqr_complex.h, k_zqtb_col_dd): the reference's acceptance statistic ||A'(A x - b)||
// (test/runtests.jl:51,62) sees the rounding of the O(mn) solve next to that of the O(mn^2) factorisation, and a
// plain-double solve in this order scores like the restated reference itself, up to 11 x LAPACK on some draws
// (docs/DESIGN_rounds1-4.md, section 1).
// Awork != nullptr: the factor is first copied there (m x n, leading dimension m; device memory) with every load in
// flight at once -- A is then pinned HOST memory, and the chunk pipeline below would pay a PCIe round trip per chunk.
#define SML_CH 16    // columns per LDS chunk
#define SML_LDR 256  // rows of a staged column
__global__ __launch_bounds__(SML_THREADS) void k_small_ldiv(const double *__restrict__ A, int64_t lda, int m, int n,
                                                            const double *__restrict__ alpha, const double *bin, double *bout,
                                                            double *xout, double *__restrict__ Awork) {
  __shared__ double buf[2][SML_CH][SML_LDR];
  __shared__ double als[2][SML_CH];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int nch = (n + SML_CH - 1) / SML_CH;
  if (Awork) {
    const int total = m * n;
    for (int base = 0; base < total; base += 8 * SML_THREADS) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * SML_THREADS + t;
        x[u] = 0.0;
        if (idx < total) x[u] = A[(int64_t)(idx % m) + (int64_t)(idx / m) * lda];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * SML_THREADS + t;
        if (idx < total) Awork[idx] = x[u];
      }
    }
    __threadfence_block();
    __syncthreads();
    A = Awork;
    lda = m;
  }
  // waves 1..3: columns [16 c, 16 c + 16) rows [0, rtop) -> buf[c & 1] (zeros beyond the matrix); every load of a thread is
  // issued before its first LDS store
  auto stage = [&](int c, int rtop) {
    double(*B)[SML_LDR] = buf[c & 1];
    constexpr int PER = (SML_CH * (SML_LDR / 2) + SML_THREADS - 64 - 1) / (SML_THREADS - 64);
    double x0[PER], x1[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = t - 64 + u * (SML_THREADS - 64);
      const int jj = e / (SML_LDR / 2), row = 2 * (e % (SML_LDR / 2)), col = SML_CH * c + jj;
      x0[u] = 0.0;
      x1[u] = 0.0;
      if (jj < SML_CH && col < n && row < rtop) {
        if (row < m) x0[u] = A[(int64_t)row + (int64_t)col * lda];
        if (row + 1 < m) x1[u] = A[(int64_t)row + 1 + (int64_t)col * lda];
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = t - 64 + u * (SML_THREADS - 64);
      const int jj = e / (SML_LDR / 2), row = 2 * (e % (SML_LDR / 2));
      if (jj < SML_CH) {
        B[jj][row] = x0[u];
        B[jj][row + 1] = x1[u];
      }
    }
    if (t >= 64 && t < 64 + SML_CH) als[c & 1][t - 64] = (SML_CH * c + t - 64 < n) ? alpha[SML_CH * c + t - 64] : 1.0;
  };
  dhqr_dd b[4];
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b[r].hi = (l + 64 * r < m) ? bin[l + 64 * r] : 0.0;
      b[r].lo = 0.0;
    }
  } else {
    stage(0, m);
  }
  __syncthreads();
  // ---- b <- Q'b: reflectors left to right (src:215-224)
  for (int c = 0; c < nch; ++c) {
    if (w == 0) {
      const double(*B)[SML_LDR] = buf[c & 1];
#pragma unroll 4
      for (int jj = 0; jj < SML_CH; ++jj) {
        const int j = SML_CH * c + jj;  // (columns >= n are staged as zeros: no-ops)
        double v[4];
        dhqr_dd p = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = l + 64 * r;
          v[r] = row >= j ? B[jj][row] : 0.0;  // rows < j of a factored column hold R
          dd_add_prod(p, v[r], b[r].hi);       // src:217: sum v_i b_i, b_i = hi + lo
          p.lo = fma(v[r], b[r].lo, p.lo);
        }
        const dhqr_dd sd = wave_sum_dd(p);
        const double s = sd.hi + sd.lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // src:218-220: b_i -= v_i s
          dd_add_prod(b[r], -s, v[r]);
          dd_renorm(b[r]);
        }
      }
    } else if (c + 1 < nch) {
      stage(c + 1, m);
    }
    __syncthreads();
  }
  // ---- back substitution, columns right to left (src:244-254): x_j = b_j / alpha_j, b[0:j] -= R[0:j, j] x_j
  // chunk nch - 1 is staged first; its parity may collide with the buffer wave 0 read last, which the barrier above released
  if (w != 0) stage(nch - 1, n);
  __syncthreads();
  for (int c = nch - 1; c >= 0; --c) {
    if (w == 0) {
      const double(*B)[SML_LDR] = buf[c & 1];
      const double rinv = dhqr_rcp(als[c & 1][l & (SML_CH - 1)]);  // lane jj: 1 / alpha of the chunk's column jj
#pragma unroll 4
      for (int jj = SML_CH - 1; jj >= 0; --jj) {
        const int j = SML_CH * c + jj;
        if (j < n) {  // wave-uniform
          const int rj = j >> 6;
          double bj = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r == rj) bj = b[r].hi + b[r].lo;
          // src:251 b_j / alpha_j: reciprocal (computed for the whole chunk before the loop, off the chain) times b_j and one
          // correction step -- the quotient to the last bit in all but rare half-way cases, three dependent operations
          // instead of the ~25 of a division
          const double bq = smq_readlane(bj, j & 63), aj = als[c & 1][jj], ri = smq_readlane(rinv, jj);
          double xj = bq * ri;
          xj = fma(fma(-aj, xj, bq), ri, xj);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = l + 64 * r;
            if (row == j) {
              b[r].hi = xj;
              b[r].lo = 0.0;
            } else if (row < j) {
              dd_add_prod(b[r], -B[jj][row], xj);  // src:248-250
            }
          }
        }
      }
    } else if (c > 0) {
      stage(c - 1, SML_CH * c);  // only rows above the chunk's last column are needed
    }
    __syncthreads();
  }
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = l + 64 * r;
      const double val = b[r].hi + b[r].lo;
      if (row < m) bout[row] = val;
      if (xout && row < n) xout[row] = val;
    }
  }
}
