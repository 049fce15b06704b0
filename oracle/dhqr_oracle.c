/*
 * dhqr_oracle.c -- CPU restatement of jwscook/DistributedHouseholderQR.jl's hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (libdhqr.so, HIP) never
 * links, imports or falls back to anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" at the element level.  The reference is pure Julia
 * (Julia 1.10.2, Manifest.toml:3); there is no julia binary in this image or on the GPU
 * box, and the reference's tests hold NO golden vectors (test/runtests.jl, test/partialdot.jl
 * only assert relative residuals / approx-equality against LinearAlgebra at run time).
 * What this oracle IS pinned against (tests/test_oracle.py):
 *   - the reference's own acceptance inequality  ||A'A x - A'b|| < 8 * (same for LAPACK QR)
 *     on its seven shapes m = 1.1 n (test/runtests.jl:42-63);
 *   - the partialdot == dot property for every start offset (test/partialdot.jl:12-20);
 *   - LAPACK dgeqrf (scipy): R equal, tau == v_jj^2, v_lapack == v / v_jj, and the known
 *     R[n,n] sign difference when m == n.
 *
 * Third-party arithmetic not under /root/reference: LinearAlgebra.norm at src:129 ->
 * BLAS dnrm2 from OpenBLAS_jll 0.3.23+4 (Manifest.toml:139-142).  On x86-64 OpenBLAS's
 * dnrm2 kernel accumulates the sum of squares in x87 extended precision (no scaling pass
 * needed); oracle_nrm2 below restates that with `long double`.
 *
 * Every function cites the reference lines (src/DistributedHouseholderQR.jl) it follows.
 * Column-major storage, 0-based indices here (the reference is 1-based).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define H_(i, j) H[(size_t)(i) + (size_t)(j) * (size_t)ldh]

/* ---- portable counter-based generator (SURVEY.md section 7 "same random inputs") ----
 * u(seed, idx) = (splitmix64_finalise(seed + (idx+1)*GOLDEN) >> 11) * 2^-53, uniform [0,1)
 * like Julia's rand(Float64) (test/runtests.jl:45-46).  Implemented identically in
 * oracle/dhqr_oracle.py (numpy) and csrc/dhqr_kernels.hip (device). */
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
double dhqr_oracle_u01(uint64_t seed, uint64_t idx) {
  uint64_t z = mix64(seed + (idx + 1ULL) * 0x9E3779B97F4A7C15ULL);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
/* A[i + j*lda] = u01(seed, i + j*m) for the full m x n matrix (global linear index). */
void dhqr_oracle_fill(double *A, int64_t m, int64_t n, int64_t lda, uint64_t seed) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      A[i + j * lda] = dhqr_oracle_u01(seed, (uint64_t)(i + j * m));
}

/* src:8  alphafactor(x::Real) = -sign(x)   (Julia sign(0.0) == 0.0, so alphafactor(0) = -0.0) */
double dhqr_oracle_alphafactor(double x) {
  if (x > 0.0) return -1.0;
  if (x < 0.0) return 1.0;
  return -x; /* +-0.0 -> -+0.0 ; NaN -> NaN like Julia's sign */
}

/* src:42-49  partialdot(a, b, is, ::Type{<:Real}): s += a[i]*b[i] over is = lo:hi (0-based,
 * inclusive lo, exclusive hi here).  The reference's @simd permits reassociation; the
 * `omp simd reduction` clause grants the C compiler the same licence. */
double dhqr_oracle_partialdot(const double *a, const double *b, int64_t lo, int64_t hi) {
  double s = 0.0;
#pragma omp simd reduction(+ : s)
  for (int64_t i = lo; i < hi; ++i) s += a[i] * b[i];
  return s;
}

/* src:156-160  hotloop!(Hl, Hj, s, is, jj, ::Type{<:Real}):  Hl[i,jj] -= Hj[i]*s */
static inline void oracle_hotloop(double *col, const double *Hj, double s, int64_t lo, int64_t hi) {
#pragma omp simd
  for (int64_t i = lo; i < hi; ++i) col[i] -= Hj[i] * s;
}

/* src:129  norm(view(Hl, j:m, j)) -> BLAS dnrm2 (OpenBLAS 0.3.23 x86-64: extended-precision
 * sum of squares, then sqrt). */
double dhqr_oracle_nrm2(const double *x, int64_t n) {
  long double s = 0.0L;
  for (int64_t i = 0; i < n; ++i) s += (long double)x[i] * (long double)x[i];
  return (double)sqrtl(s);
}

/* src:127-136  reflector construction for (global) column j held at `col` (length m):
 *   s = norm(H[j:m,j]); alpha[j] = s*alphafactor(H[j,j]); f = 1/sqrt(s*(s+abs(H[j,j])));
 *   H[j,j] -= alpha[j]; H[j:m,j] *= f.
 * src:138-140 then copies the WHOLE column into Hj (rows < j included, unused later). */
static void oracle_reflector(double *col, int64_t m, int64_t j, double *alpha_j, double *Hj) {
  double s = dhqr_oracle_nrm2(col + j, m - j);
  double hjj = col[j];
  double a = s * dhqr_oracle_alphafactor(hjj);
  double f = 1.0 / sqrt(s * (s + fabs(hjj)));
  *alpha_j = a;
  col[j] = hjj - a;
  for (int64_t i = j; i < m; ++i) col[i] *= f;
  memcpy(Hj, col, (size_t)m * sizeof(double));
}

/* src:198-213  _householder_inner!(H, j, Hj) on a LocalColumnBlock owning global columns
 * [col_lo, col_hi):  for jj in (j+1:n) ∩ colrange:  s = partialdot(Hj, H[:,jj], j:m);
 * hotloop!(H, Hj, s, j:m, jj).  Hl points at the block's first column (global col_lo).
 * Columns are chunked contiguously over threads exactly like src:203-207 (static schedule). */
void dhqr_oracle_householder_inner(double *Hl, int64_t m, int64_t ldh, int64_t j, const double *Hj,
                                   int64_t col_lo, int64_t col_hi, int64_t n) {
  int64_t lo = j + 1 > col_lo ? j + 1 : col_lo;
  int64_t hi = n < col_hi ? n : col_hi;
  if (lo >= hi) return; /* src:202 isempty(jjs) && return */
#pragma omp parallel for schedule(static)
  for (int64_t jj = lo; jj < hi; ++jj) {
    double *col = Hl + (size_t)(jj - col_lo) * (size_t)ldh;
    double s = dhqr_oracle_partialdot(Hj, col, j, m);
    oracle_hotloop(col, Hj, s, j, m);
  }
}

/* src:122-136 (first half of _householder! for ONE column j of a local block): builds the
 * reflector in place, writes alpha[j], fills Hj.  Exposed so the multi-process restatement
 * (oracle/dist_oracle.py, mirroring householder!(::DArray) src:115-120) can interleave the
 * per-column "broadcast" of Hj between owner and peers. */
void dhqr_oracle_reflector_step(double *Hl, int64_t m, int64_t ldh, int64_t j, int64_t col_lo,
                                double *alpha, double *Hj) {
  oracle_reflector(Hl + (size_t)(j - col_lo) * (size_t)ldh, m, j, &alpha[j], Hj);
}

/* src:113 + src:122-148  householder!(A, alpha) for a plain (non-distributed) matrix:
 * colrange = 1:n, a single "process".  In place: on return H[j:m,j] = v_j (||v_j||^2 = 2,
 * diagonal included), H[i,j] = R[i,j] for i<j, alpha[j] = R[j,j]  (src:296-309 format). */
void dhqr_oracle_householder(double *H, int64_t m, int64_t n, int64_t ldh, double *alpha) {
  double *Hj = (double *)malloc((size_t)m * sizeof(double)); /* src:125 */
  for (int64_t j = 0; j < n; ++j) {                          /* src:127 */
    oracle_reflector(&H_(0, j), m, j, &alpha[j], Hj);        /* src:129-140 */
    dhqr_oracle_householder_inner(H, m, ldh, j, Hj, 0, n, n); /* src:141-143, procs(H) == 1 */
  }
  free(Hj);
}

/* Same algorithm, but only the first `ncols_done` reflectors are processed (each applied to
 * ALL n columns).  Used by bench.py's cpu_baseline leg: a bounded sample of the 32768^2
 * workload ("time the first 256 columns and extrapolate"). */
void dhqr_oracle_householder_prefix(double *H, int64_t m, int64_t n, int64_t ldh, double *alpha,
                                    int64_t ncols_done) {
  double *Hj = (double *)malloc((size_t)m * sizeof(double));
  if (ncols_done > n) ncols_done = n;
  for (int64_t j = 0; j < ncols_done; ++j) {
    oracle_reflector(&H_(0, j), m, j, &alpha[j], Hj);
    dhqr_oracle_householder_inner(H, m, ldh, j, Hj, 0, n, n);
  }
  free(Hj);
}

/* src:215-224 / src:232-242  _solve_householder1!: b <- Q' b, reflectors in column order.
 * Restricted to the local columns [col_lo, col_hi) like _solve_householder1_inner!. */
void dhqr_oracle_solve1_inner(double *b, const double *Hl, int64_t m, int64_t ldh, int64_t col_lo,
                              int64_t col_hi, int64_t n) {
  int64_t hi = n < col_hi ? n : col_hi;
  for (int64_t j = col_lo; j < hi; ++j) {
    const double *v = Hl + (size_t)(j - col_lo) * (size_t)ldh;
    double s = dhqr_oracle_partialdot(v, b, j, m); /* src:237 */
    for (int64_t i = j; i < m; ++i) b[i] -= v[i] * s; /* src:238-240 */
  }
}

/* src:272-282  _solve_householder2_inner!(b, H, i): sum_{j in local, j>i} H[i,j]*b[j] */
double dhqr_oracle_solve2_inner(const double *b, const double *Hl, int64_t ldh, int64_t i,
                                int64_t col_lo, int64_t col_hi, int64_t n) {
  int64_t lo = i + 1 > col_lo ? i + 1 : col_lo;
  int64_t hi = n < col_hi ? n : col_hi;
  double bi = 0.0;
  for (int64_t j = lo; j < hi; ++j) bi += Hl[(size_t)i + (size_t)(j - col_lo) * (size_t)ldh] * b[j];
  return bi;
}

/* src:244-254  _solve_householder2!: back substitution, rows n..1, divide by alpha[i]. */
static void oracle_solve2(double *b, const double *H, int64_t ldh, int64_t n, const double *alpha) {
  for (int64_t i = n - 1; i >= 0; --i) {
    double bi = b[i];
    for (int64_t j = i + 1; j < n; ++j) bi -= H_(i, j) * b[j]; /* src:248-250 */
    b[i] = bi / alpha[i];                                      /* src:251 */
  }
}

/* src:284-294  solve_householder!(b, H, alpha): b <- Q'b, then back-sub; x = b[1:n]
 * (the caller reads the first n entries of b).  Mutates b like the reference. */
void dhqr_oracle_solve(double *b, const double *H, int64_t m, int64_t n, int64_t ldh,
                       const double *alpha) {
  dhqr_oracle_solve1_inner(b, H, m, ldh, 0, n, n);
  oracle_solve2(b, H, ldh, n, alpha);
}

/* ---- helpers used only by tests (not in the reference): form Q*R from the factor format to
 * measure ||A - QR||_F / ||A||_F (the north-star metric).  B = [R;0]; for j = n..1: B <- H_j B. */
void dhqr_oracle_form_qr(const double *H, int64_t m, int64_t n, int64_t ldh, const double *alpha,
                         double *B, int64_t ldb) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      B[i + j * ldb] = (i < j) ? H_(i, j) : (i == j ? alpha[j] : 0.0);
  for (int64_t j = n - 1; j >= 0; --j) {
    const double *v = &H_(0, j);
#pragma omp parallel for schedule(static)
    for (int64_t c = j; c < n; ++c) { /* columns < j of [R;0] are zero in rows >= j */
      double *col = B + (size_t)c * (size_t)ldb;
      double s = dhqr_oracle_partialdot(v, col, j, m);
      oracle_hotloop(col, v, s, j, m);
    }
  }
}

int dhqr_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
