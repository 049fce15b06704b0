// dhqr_panel.h -- latency-bound part of the blocked path: factorisation of a tall m_k x 128 panel.
//
// The reference builds one reflector per column and needs a full-column reduction before every
// rank-1 update (src/DistributedHouseholderQR.jl:129, 208): 128 dependent global reductions per
// panel.  Here the panel is processed in sub-panels of DHQR_IB columns; inside a sub-panel ONE
// kernel launch per column (k_panel_step) does, for a grid of (row chunk x column) workgroups:
//     reduce the partial dots of this step                      (cross-workgroup, via `part`)
//     alpha, f, v = f*(a_j - alpha e_j)                         (src:129-135)
//     a_k -= v * (v' a_k)      with  v' a_k = f*(a_j'a_k - alpha*a_jk)     (src:208-209)
//     partial dots of the NEXT pivot column a_{j+1}' a_k on the just-updated registers
// so norm and dot of the reference collapse into one reduction per column and every element is
// read once and written once per step.  (a_j'a_k - alpha*a_jk is algebraically v'a_k/f; both are
// sums of the same products, so the rounding behaviour is that of the plain dot.)
// Between sub-panels the block reflector of the finished sub-panel is applied to the rest of the
// panel with the MFMA GEMMs (dhqr_gemm.h).
//
// Race freedom inside a launch: workgroup (c,k) owns rows-chunk c of column k.  Everything another
// workgroup needs from a column that is being modified is read from side buffers written by the
// PREVIOUS launch: `piv` (the current pivot column), `prow` (row j of the sub-panel).  The next
// pivot column is written to `pivnext`, not in place; v_j goes to the packed V buffers and is
// copied back into A by k_unpack_v once the panel is finished.
#pragma once
#include "dhqr_common.h"

#define DHQR_IB 64   // sub-panel width (measured: 64 beats 32 and 16 while the inter-sub-panel GEMMs are 128-wide)
#define PS_RC 1024   // rows per workgroup chunk (256 threads x 4 rows)

// chunk-local rows of thread t: VEC=2: pairs at 2*(t + i*256), i=0,1 ; VEC=1: t + i*256, i=0..3
template <int VEC>
__device__ __forceinline__ int64_t ps_row(int64_t cbase, int t, int e) {
  return (VEC == 2) ? cbase + 2 * (t + (e >> 1) * 256) + (e & 1) : cbase + t + e * 256;
}
template <int VEC>
__device__ __forceinline__ void ps_load(const double *__restrict__ col, int64_t cbase, int t,
                                        int64_t rows, double (&x)[4]) {
  if constexpr (VEC == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t r = cbase + 2 * (t + i * 256);
      const bool ok = r < rows;  // rows even: pairs all-or-nothing
      const double2 z = *reinterpret_cast<const double2 *>(col + (ok ? r : 0));
      x[2 * i] = ok ? z.x : 0.0;
      x[2 * i + 1] = ok ? z.y : 0.0;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t r = cbase + t + e * 256;
      const bool ok = r < rows;
      const double z = col[ok ? r : 0];
      x[e] = ok ? z : 0.0;
    }
  }
}
template <int VEC>
__device__ __forceinline__ void ps_store(double *__restrict__ col, int64_t cbase, int t,
                                         int64_t rows, const double (&x)[4]) {
  if constexpr (VEC == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t r = cbase + 2 * (t + i * 256);
      if (r < rows) *reinterpret_cast<double2 *>(col + r) = make_double2(x[2 * i], x[2 * i + 1]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t r = cbase + t + e * 256;
      if (r < rows) col[r] = x[e];
    }
  }
}

// Sub-panel start: partial dots of column 0 with every sub-panel column, copy of column 0 into the
// pivot buffer, copy of row 0 into prow.  grid = (nch, ncols_sub).  Ps = &A[j0, j0].
template <int VEC>
__global__ __launch_bounds__(256) void k_panel_init(const double *__restrict__ Ps, int64_t ldp,
                                                    int64_t rows, double *__restrict__ piv,
                                                    double *__restrict__ prow,
                                                    double *__restrict__ part, int nch) {
  __shared__ double red[6];
  const int t = threadIdx.x, c = blockIdx.x, k = blockIdx.y;
  const int64_t cbase = (int64_t)c * PS_RC;
  double a0[4], ak[4];
  ps_load<VEC>(Ps, cbase, t, rows, a0);
  ps_load<VEC>(Ps + (int64_t)k * ldp, cbase, t, rows, ak);
  double d = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) d = fma(a0[e], ak[e], d);
  d = block_sum<256>(d, red);
  if (t == 0) part[(int64_t)k * nch + c] = d;
  if (k == 0) ps_store<VEC>(piv, cbase, t, rows, a0);
  if (c == 0 && t == 0) prow[k] = ak[0];  // row 0 is element 0 of thread 0 in both layouts
}

// One reflector step of a sub-panel.  q = local column of the reflector (its diagonal is local
// row q); grid = (nch - q/PS_RC, ncs - q) with ncs = columns in this sub-panel; blockIdx.y = 0 is
// the pivot column itself (emits v), blockIdx.y = k' >= 1 updates column q + k'.
template <int VEC>
__global__ __launch_bounds__(256) void k_panel_step(
    double *__restrict__ Ps, int64_t ldp, int64_t rows, int q, int ncs,
    const double *__restrict__ piv, double *__restrict__ pivnext, const double *__restrict__ prow,
    double *__restrict__ prownext, const double *__restrict__ part, double *__restrict__ partnext,
    int nch, double *__restrict__ Vs, int64_t ldvs, double *__restrict__ Vw, int64_t ldvw,
    double *__restrict__ alpha_q) {
  __shared__ double red[6];
  const int t = threadIdx.x, lane = t & 63;
  const int c0 = q / PS_RC;
  const int c = c0 + blockIdx.x;
  const int kk = blockIdx.y, kq = q + kk;
  const int64_t cbase = (int64_t)c * PS_RC;
  const bool havenext = (q + 1 < ncs);

  // (1) finish this step's reductions (every wave redundantly; fixed order => deterministic)
  double dj = 0.0, dk = 0.0, dn = 0.0;
  for (int cc = c0 + lane; cc < nch; cc += 64) {
    dj += part[(int64_t)q * nch + cc];
    dk += part[(int64_t)kq * nch + cc];
    if (havenext) dn += part[(int64_t)(q + 1) * nch + cc];
  }
  dj = wave_sum(dj);
  dk = wave_sum(dk);
  dn = wave_sum(dn);

  // (2) reflector scalars (src:129-131), identical in every workgroup
  const double h = prow[q];
  const double s = sqrt(dj);
  const double al = s * dhqr_alphafactor(h);
  const double f = 1.0 / sqrt(s * (s + fabs(h)));

  // (3) v on this chunk from the unscaled pivot column
  double v[4];
  ps_load<VEC>(piv, cbase, t, rows, v);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t r = ps_row<VEC>(cbase, t, e);
    v[e] = (r == q) ? (h - al) * f : (r > q ? v[e] * f : 0.0);  // src:132-135
  }

  if (kk == 0) {  // (4) pivot column: publish v (packed buffers) and alpha
    ps_store<VEC>(Vs + (int64_t)q * ldvs, cbase, t, rows, v);
    ps_store<VEC>(Vw, cbase, t, rows, v);  // Vw already points at (row j0, column of this reflector)
    if (blockIdx.x == 0 && t == 0) *alpha_q = al;
    return;
  }

  // (5) rank-1 update of column kq on this chunk
  const double wk = f * (dk - al * prow[kq]);  // v' a_k
  double a[4];
  double *colk = Ps + (int64_t)kq * ldp;
  ps_load<VEC>(colk, cbase, t, rows, a);
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = fma(-v[e], wk, a[e]);  // src:209
  if (kk == 1) {
    // next pivot column: staged in pivnext (other workgroups still read the old values from A);
    // only its final R entry (row q) goes in place
    ps_store<VEC>(pivnext, cbase, t, rows, a);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ps_row<VEC>(cbase, t, e) == q) colk[q] = a[e];
  } else {
    ps_store<VEC>(colk, cbase, t, rows, a);
  }
  if (!havenext) return;

  // (6) partial dot of the next step on the updated registers: a_{q+1}' a_k over rows >= q+1
  double d = 0.0;
  if (kk == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ps_row<VEC>(cbase, t, e) > q) d = fma(a[e], a[e], d);
  } else {
    const double wn = f * (dn - al * prow[q + 1]);
    double an[4];
    ps_load<VEC>(Ps + (int64_t)(q + 1) * ldp, cbase, t, rows, an);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      an[e] = fma(-v[e], wn, an[e]);
      if (ps_row<VEC>(cbase, t, e) > q) d = fma(an[e], a[e], d);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (ps_row<VEC>(cbase, t, e) == q + 1) prownext[kq] = a[e];  // row q+1 after this step
  d = block_sum<256>(d, red);
  if (t == 0) partnext[(int64_t)kq * nch + c] = d;
}

// A[r, p] <- Vw[r, p] for r >= p (the reflectors, produced out of place by k_panel_step)
__global__ __launch_bounds__(256) void k_unpack_v(double *__restrict__ P, int64_t ldp, int64_t rows,
                                                  int64_t ncols, const double *__restrict__ Vw,
                                                  int64_t ldv) {
  const int64_t p = blockIdx.y;
  if (p >= ncols) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = p + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride)
    P[r + p * ldp] = Vw[r + p * ldv];
}

// Compact-WY T from S = V'V (see dhqr_gemm.h for the algebra): T = (I + striu(S))^{-1}, one thread
// per column j solving U x = e_j by back substitution; N = striu(S) and X = T live packed
// (row i holds entries i..127) in LDS, no barriers and no global loads inside the solve.
// Columns >= ncols of V are zero padding: T[j][j] = 1, rest of the column 0.
__global__ __launch_bounds__(128) void k_build_t2(const double *__restrict__ S, int ncols,
                                                  double *__restrict__ Tout,
                                                  double *__restrict__ Ttout) {
  constexpr int N = 128, PK = N * (N + 1) / 2;
  __shared__ double Nl[PK];
  __shared__ double Xl[PK];
  const int t = threadIdx.x;
  auto pidx = [](int i, int l) { return i * N - (i * (i - 1)) / 2 + (l - i); };  // i <= l
  for (int idx = t; idx < N * N; idx += N) {
    const int i = idx & (N - 1), l = idx >> 7;
    if (i <= l) Nl[pidx(i, l)] = (i < l && l < ncols) ? S[i + l * N] : 0.0;
  }
  __syncthreads();
  const int j = t;
  Xl[pidx(j, j)] = 1.0;
  // rows walked in lockstep by the whole wave (i uniform): the N[i][l] reads are LDS broadcasts
  for (int i = N - 2; i >= 0; --i) {
    if (i < j) {
      double acc0 = 0.0, acc1 = 0.0;
      if (j < ncols) {
        acc0 = Nl[pidx(i, j)];  // l = j term: N[i][j] * X[j][j]
        int l = i + 1;
        for (; l + 1 < j; l += 2) {
          acc0 = fma(Nl[pidx(i, l)], Xl[pidx(l, j)], acc0);
          acc1 = fma(Nl[pidx(i, l + 1)], Xl[pidx(l + 1, j)], acc1);
        }
        if (l < j) acc0 = fma(Nl[pidx(i, l)], Xl[pidx(l, j)], acc0);
      }
      Xl[pidx(i, j)] = -(acc0 + acc1);
    }
  }
  __syncthreads();
  for (int idx = t; idx < N * N; idx += N) {
    const int i = idx & (N - 1), l = idx >> 7;  // Tout[i + l*N] = T[i][l]
    Tout[idx] = (i <= l) ? Xl[pidx(i, l)] : 0.0;
    Ttout[idx] = (l <= i) ? Xl[pidx(l, i)] : 0.0;  // Tt[i + l*N] = T[l][i]
  }
}
