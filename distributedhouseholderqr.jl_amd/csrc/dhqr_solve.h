// dhqr_solve.h -- back substitution, KAT dot hook and the ||A - QR|| metric kernels.
//
// Reference mapping (src/DistributedHouseholderQR.jl):
//   k_backsub_diag + k_backsub_update   src:244-254 / src:256-282 (_solve_householder2!), blocked
//       by 64 rows: the reference walks rows n..1 with a stride-m row dot per row (and one RPC +
//       sum-reduction per row when distributed); here each 64 x 64 diagonal block is solved by
//       one wavefront out of LDS and the rows above are updated by a column-oriented (coalesced)
//       GEMV.
//   k_partialdot_*                      src:42-49 partialdot (test/partialdot.jl hook)
//   Q'b (src:215-242) is the nrhs = 1 case of the blocked apply in dhqr_api.hip.
#pragma once
#include "dhqr_common.h"

#define BS_NB 64

// Solve the diagonal block rows/cols [lo, hi) (hi - lo <= 64):  x_i = (b_i - sum_{j>i} R_ij x_j)/alpha_i
// One wavefront; R block staged in LDS; column sweep so every step is one broadcast + one fma.
__global__ __launch_bounds__(64) void k_backsub_diag(const double *__restrict__ A, int64_t lda,
                                                     const double *__restrict__ alpha,
                                                     double *__restrict__ b, int64_t lo,
                                                     int64_t hi) {
  __shared__ double Rs[BS_NB * (BS_NB + 1)];
  const int t = threadIdx.x;
  const int nb = (int)(hi - lo);
  for (int c = 0; c < nb; ++c)
    if (t < c) Rs[c * (BS_NB + 1) + t] = A[(lo + t) + (lo + c) * lda];  // strict upper part
  __syncthreads();
  double bi = (t < nb) ? b[lo + t] : 0.0;
  const double ai = (t < nb) ? alpha[lo + t] : 1.0;
  for (int c = nb - 1; c >= 0; --c) {
    double xc = 0.0;
    if (t == c) xc = bi / ai;                       // src:251 b[i] = bi / alpha[i]
    xc = __shfl(xc, c, 64);
    if (t == c) bi = xc;
    if (t < c) bi = fma(-Rs[c * (BS_NB + 1) + t], xc, bi);  // src:248-250
  }
  if (t < nb) b[lo + t] = bi;
}

// b[0:lo] -= R[0:lo, lo:hi] * x[lo:hi]   (x already stored in b[lo:hi])
__global__ __launch_bounds__(256) void k_backsub_update(const double *__restrict__ A, int64_t lda,
                                                        double *__restrict__ b, int64_t lo,
                                                        int64_t hi) {
  __shared__ double xs[BS_NB];
  const int t = threadIdx.x;
  const int nb = (int)(hi - lo);
  if (t < nb) xs[t] = b[lo + t];
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + t;
  if (r >= lo) return;
  double acc = 0.0;
  for (int c = 0; c < nb; ++c) acc = fma(A[r + (lo + c) * lda], xs[c], acc);
  b[r] -= acc;
}

// b[top:rend] -= R[top:rend, xlo:xhi] * b[xlo:xhi]  (row-range variant used by the block-level
// entry point; rend <= xlo so the x entries are never modified)
__global__ __launch_bounds__(256) void k_backsub_update_range(const double *__restrict__ A,
                                                              int64_t lda, double *__restrict__ b,
                                                              int64_t top, int64_t rend, int64_t xlo,
                                                              int64_t xhi) {
  __shared__ double xs[BS_NB];
  const int t = threadIdx.x;
  const int nb = (int)(xhi - xlo);
  if (t < nb) xs[t] = b[xlo + t];
  __syncthreads();
  const int64_t r = top + (int64_t)blockIdx.x * blockDim.x + t;
  if (r >= rend) return;
  double acc = 0.0;
  for (int c = 0; c < nb; ++c) acc = fma(A[r + (xlo + c) * lda], xs[c], acc);
  b[r] -= acc;
}

// acc[0:rows] -= A[0:rows, 0:nb] * x[0:nb], nb <= 256: what a pair of panels solved on this rank contributes to the rows above
// it (cs_solve at P > 1, dhqr_dist.h); A = the pair's columns from row 0, x = the pair's solved entries
__global__ __launch_bounds__(256) void k_backsub_update_wide(const double *__restrict__ A, int64_t lda, double *__restrict__ acc,
                                                             int64_t rows, const double *__restrict__ x, int nb) {
  __shared__ double xs[256];
  const int t = threadIdx.x;
  xs[t] = (t < nb) ? x[t] : 0.0;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + t;
  if (r >= rows) return;
  double a0 = 0.0, a1 = 0.0;
  int c = 0;
  for (; c + 1 < nb; c += 2) {
    a0 = fma(A[r + (int64_t)c * lda], xs[c], a0);
    a1 = fma(A[r + (int64_t)(c + 1) * lda], xs[c + 1], a1);
  }
  if (c < nb) a0 = fma(A[r + (int64_t)c * lda], xs[c], a0);
  acc[r] -= a0 + a1;
}
// x[i] += s[i], i < n (any n)
__global__ __launch_bounds__(256) void k_axpy_n(double *__restrict__ x, const double *__restrict__ s, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += s[i];
}

// partialdot hook: partial sums per workgroup, then one workgroup finishes.
__global__ __launch_bounds__(256) void k_partialdot_partial(const double *__restrict__ a,
                                                            const double *__restrict__ b,
                                                            int64_t lo, int64_t hi,
                                                            double *__restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride)
    s = fma(a[i], b[i], s);
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sum_final(const double *__restrict__ part, int n,
                                                   double *__restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) *out = s;
}

// W = [R; 0] from the factor format: W[i,jl] = A[i,jl] (i<gj), alpha[gj] (i==gj), 0 (i>gj), with
// gj the GLOBAL index of local column jl in a block-cyclic column layout (identity for 1 rank).
__global__ __launch_bounds__(256) void k_form_r0(const double *__restrict__ A, int64_t lda,
                                                 const double *__restrict__ alpha, int64_t m,
                                                 int64_t ncols, double *__restrict__ W, int64_t ldw,
                                                 int64_t cb, int nranks, int rank) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t jl = blockIdx.y; jl < ncols; jl += gridDim.y) {  // grid.y is capped at 32768
    const int64_t j = ((jl / cb) * nranks + rank) * cb + jl % cb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      double x = 0.0;
      if (i < j) x = A[i + jl * lda];
      else if (i == j) x = alpha[j];
      W[i + jl * ldw] = x;
    }
  }
}

// part[2*b] = sum (X - Y)^2, part[2*b+1] = sum X^2 over this workgroup's grid-stride share
__global__ __launch_bounds__(256) void k_diff_norms(const double *__restrict__ X, int64_t ldx,
                                                    const double *__restrict__ Y, int64_t ldy,
                                                    int64_t m, int64_t n,
                                                    double *__restrict__ part) {
  __shared__ double red[4];
  double d2 = 0.0, x2 = 0.0;
  const int64_t total = m * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t j = e / m, i = e - j * m;
    const double x = X[i + j * ldx], y = Y[i + j * ldy];
    d2 = fma(x - y, x - y, d2);
    x2 = fma(x, x, x2);
  }
  d2 = block_sum<256>(d2, red);
  x2 = block_sum<256>(x2, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = d2;
    part[2 * blockIdx.x + 1] = x2;
  }
}
__global__ __launch_bounds__(256) void k_sum2_final(const double *__restrict__ part, int n,
                                                    double *__restrict__ out) {
  __shared__ double red[4];
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    s0 += part[2 * i];
    s1 += part[2 * i + 1];
  }
  s0 = block_sum<256>(s0, red);
  s1 = block_sum<256>(s1, red);
  if (threadIdx.x == 0) {
    out[0] = s0;
    out[1] = s1;
  }
}

// Row-split driver: this rank's rows of [R; 0] (local row i is global row row0 + i; alpha = diag(R))
__global__ __launch_bounds__(256) void k_form_r0_rows(const double *__restrict__ A, int64_t lda,
                                                      const double *__restrict__ alpha, int64_t mloc, int64_t n,
                                                      int64_t row0, double *__restrict__ B, int64_t ldb) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = blockIdx.y; j < n; j += gridDim.y)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mloc; i += stride) {
      const int64_t g = row0 + i;
      B[i + j * ldb] = (g < j) ? A[i + j * lda] : (g == j ? alpha[j] : 0.0);
    }
}
// Row-split back substitution: b[i] -= sum_{k<w} R[i, c0+k] x[k] for the first `nrows` local rows (all above the block)
__global__ __launch_bounds__(256) void k_rs_backsub_update(const double *__restrict__ A, int64_t lda,
                                                           double *__restrict__ b, int64_t nrows, int64_t c0, int w,
                                                           const double *__restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  double s = 0.0;
  for (int k = 0; k < w; ++k) s = fma(A[i + (c0 + k) * lda], x[k], s);
  b[i] -= s;
}
