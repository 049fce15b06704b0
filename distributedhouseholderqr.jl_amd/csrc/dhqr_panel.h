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

#define DHQR_IB 64   // sub-panel width (measured on MI355X at 32768^2: 64 > 32 > 16)
#define PS_CPW 4     // columns per workgroup in k_panel_step
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
// row q); ncs = columns in this sub-panel.  grid = (nch - q/PS_RC, 1 + ceil((ncs-q-1)/CPW)):
// blockIdx.y = 0 is the pivot column itself (emits v and alpha); blockIdx.y = y >= 1 updates the
// CPW columns starting at q + 1 + (y-1)*CPW on row chunk blockIdx.x.  The pivot chunk v and the
// updated next-pivot chunk are formed once per workgroup and reused for its CPW columns.
template <int VEC, int CPW>
__global__ __launch_bounds__(256) void k_panel_step(
    double *__restrict__ Ps, int64_t ldp, int64_t rows, int q, int ncs,
    const double *__restrict__ piv, double *__restrict__ pivnext, const double *__restrict__ prow,
    double *__restrict__ prownext, const double *__restrict__ part, double *__restrict__ partnext,
    int nch, double *__restrict__ Vs, int64_t ldvs, double *__restrict__ Vw, int64_t ldvw,
    double *__restrict__ alpha_q) {
  __shared__ double red[4 * CPW];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int c0 = q / PS_RC;
  const int c = c0 + blockIdx.x;
  const int y = blockIdx.y;
  const int first = q + 1 + (y - 1) * CPW;              // first column of this group (y >= 1)
  const int ncol = (y == 0) ? 0 : ((ncs - first < CPW) ? ncs - first : CPW);
  const int64_t cbase = (int64_t)c * PS_RC;
  const bool havenext = (q + 1 < ncs);

  // (0) issue every chunk load of this workgroup first: pivot chunk, old next-pivot chunk and
  // the CPW columns -- they do not depend on the reductions below and overlap their latency.
  double v[4], an[4], a[CPW][4];
  ps_load<VEC>(piv, cbase, t, rows, v);
  if (y > 0) {
    ps_load<VEC>(Ps + (int64_t)(q + 1) * ldp, cbase, t, rows, an);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int kq = (i < ncol) ? first + i : first;  // clamp: load something valid, ignore it
      ps_load<VEC>(Ps + (int64_t)kq * ldp, cbase, t, rows, a[i]);
    }
  }

  // (1) finish this step's reductions (every wave redundantly; fixed order => deterministic)
  double dj = 0.0, dn = 0.0, dk[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) dk[i] = 0.0;
  for (int cc = c0 + lane; cc < nch; cc += 64) {
    dj += part[(int64_t)q * nch + cc];
    if (havenext) dn += part[(int64_t)(q + 1) * nch + cc];
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      if (i < ncol) dk[i] += part[(int64_t)(first + i) * nch + cc];
  }
  dj = wave_sum(dj);
  dn = wave_sum(dn);
#pragma unroll
  for (int i = 0; i < CPW; ++i) dk[i] = wave_sum(dk[i]);

  // (2) reflector scalars (src:129-131), identical in every workgroup
  const double h = prow[q];
  const double s = sqrt(dj);
  const double al = s * dhqr_alphafactor(h);
  const double f = 1.0 / sqrt(s * (s + fabs(h)));

  // (3) v on this chunk from the unscaled pivot column
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t r = ps_row<VEC>(cbase, t, e);
    v[e] = (r == q) ? (h - al) * f : (r > q ? v[e] * f : 0.0);  // src:132-135
  }

  if (y == 0) {  // (4) pivot column: publish v (packed buffers) and alpha
    ps_store<VEC>(Vs + (int64_t)q * ldvs, cbase, t, rows, v);
    ps_store<VEC>(Vw, cbase, t, rows, v);  // Vw already points at (row j0, column of this reflector)
    if (blockIdx.x == 0 && t == 0) *alpha_q = al;
    return;
  }

  // updated next-pivot chunk a_{q+1} - v (v'a_{q+1}), from the OLD column q+1 still in A
  {
    const double wn = f * (dn - al * prow[q + 1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) an[e] = fma(-v[e], wn, an[e]);
  }

  // (5)+(6) rank-1 update of each column of the group, then its partial dot for the next step
  double d[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    d[i] = 0.0;
    if (i < ncol) {
      const int kq = first + i;
      const double wk = f * (dk[i] - al * prow[kq]);  // v' a_k = f (a_j'a_k - alpha a_jk)
      double *colk = Ps + (int64_t)kq * ldp;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[i][e] = fma(-v[e], wk, a[i][e]);  // src:209
      if (kq == q + 1) {
        // next pivot column: staged in pivnext (other workgroups still read the old values from
        // A); only its final R entry (row q) goes in place
        ps_store<VEC>(pivnext, cbase, t, rows, a[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ps_row<VEC>(cbase, t, e) == q) colk[q] = a[i][e];
      } else {
        ps_store<VEC>(colk, cbase, t, rows, a[i]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t r = ps_row<VEC>(cbase, t, e);
        if (r > q) d[i] = fma(an[e], a[i][e], d[i]);   // a_{q+1}' a_k over rows >= q+1
        if (r == q + 1) prownext[kq] = a[i][e];        // row q+1 after this step
      }
    }
  }
  // block reduction of the CPW partial dots
#pragma unroll
  for (int i = 0; i < CPW; ++i) d[i] = wave_sum(d[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < CPW; ++i) red[wv * CPW + i] = d[i];
  }
  __syncthreads();
  if (t < ncol) partnext[(int64_t)(first + t) * nch + c] = red[t] + red[CPW + t] + red[2 * CPW + t] + red[3 * CPW + t];
}

// A[r, p] <- Vw[r, p] for r >= p (the reflectors, produced out of place by k_panel_step)
__global__ __launch_bounds__(256) void k_unpack_v(double *__restrict__ P, int64_t ldp, int64_t rows,
                                                  int64_t ncols, const double *__restrict__ Vw,
                                                  int64_t ldv, const int *__restrict__ stat, int epoch) {
  const int64_t p = blockIdx.y;
  if (p >= ncols) return;
  if (stat != nullptr && stat[0] <= epoch) return;  // commit predicate (see k_gemm_nn_sub)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = p + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride)
    P[r + p * ldp] = Vw[r + p * ldv];
}

// Row-split driver (dhqr_rowsplit.h): ranks that do not hold the panel's diagonal rows move whole rows.
// Vw[r + p*ldv] = P[r + p*ldp] for p < ncols, 0 in the padding columns and in the pad rows [rows, npad).
__global__ __launch_bounds__(256) void k_pack_rows(const double *__restrict__ P, int64_t ldp, int64_t rows,
                                                   int64_t ncols, double *__restrict__ Vw, int64_t ldv, int64_t npad) {
  const int64_t p = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < npad; r += stride)
    Vw[r + p * ldv] = (p < ncols && r < rows) ? P[r + p * ldp] : 0.0;
}
__global__ __launch_bounds__(256) void k_unpack_rows(double *__restrict__ P, int64_t ldp, int64_t rows, int64_t ncols,
                                                     const double *__restrict__ Vw, int64_t ldv,
                                                     const int *__restrict__ stat, int epoch) {
  const int64_t p = blockIdx.y;
  if (p >= ncols) return;
  if (stat != nullptr && stat[0] <= epoch) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) P[r + p * ldp] = Vw[r + p * ldv];
}
