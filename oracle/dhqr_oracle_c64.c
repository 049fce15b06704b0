/*
 * dhqr_oracle_c64.c -- CPU restatement of the ComplexF64 methods of the reference's hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as dhqr_oracle.c: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it).
 *
 * PARITY STATUS: "parity unpinned" at the element level (no Julia, no golden vectors).  Pinned
 * in tests/test_oracle_complex.py against what the reference's own tests assert for ComplexF64:
 *   - partialdot == dot(a[i:end], b[i:end]) for N = 1..20 and every offset (test/partialdot.jl:12-20
 *     -- the reference's only known-answer test, and it is ComplexF64-only);
 *   - ||A'A x - A'b|| < 8 * (same for LAPACK QR) on the seven shapes m = 1.1 n
 *     (test/runtests.jl:42-63 with T = ComplexF64);
 * and against LAPACK zgeqrf (scipy): |R| equal row by row up to the unit phase of alpha_j
 * (the reference's R is NOT phase-normalised: diag(R) = alpha_j = -e^{i arg a_jj} s is complex).
 *
 * Complex numbers are interleaved (re, im) pairs of doubles == Julia's ComplexF64 == C99
 * `double _Complex`.  Column-major, 0-based here.  `src:` = /root/reference/src/DistributedHouseholderQR.jl.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex zc;

#define HZ(i, j) H[(size_t)(i) + (size_t)(j) * (size_t)ldh]

static inline zc mk(double re, double im) {
  union { zc z; double d[2]; } u;
  u.d[0] = re;
  u.d[1] = im;
  return u.z;
}

/* src:9  alphafactor(x::Complex) = -exp(im * angle(x));  angle(0 + 0im) = 0  =>  -1 */
void dhqr_oracle_alphafactor_c64(const double *x, double *out) {
  const double th = atan2(x[1], x[0]);
  out[0] = -cos(th);
  out[1] = -sin(th);
}

/* src:51-59  partialdot(a, b, is, ::Type{<:Complex}):
 *   s += Complex(ar*br + ai*bi, ar*bi - ai*br)   (= conj(a[i]) * b[i]), plain sequential loop. */
void dhqr_oracle_partialdot_c64(const zc *a, const zc *b, int64_t lo, int64_t hi, double *out) {
  double sr = 0.0, si = 0.0;
  for (int64_t i = lo; i < hi; ++i) {
    const double ar = creal(a[i]), ai = cimag(a[i]);
    const double br = creal(b[i]), bi = cimag(b[i]);
    sr += ar * br + ai * bi;
    si += ar * bi - ai * br;
  }
  out[0] = sr;
  out[1] = si;
}

/* src:150-154 / src:171-196  hotloop!(Hl, Hj, s, is, jj, ::Type{ComplexF64}):  Hl[i,jj] -= Hj[i]*s.
 * The reference's SIMD form (muladd with (-sr,-si) and (si,-sr) on shuffled lanes) evaluates
 *   re -= sr*vr ; re += si*vi        im -= si*vr ; im -= sr*vi
 * as two fused multiply-adds per component; fma() below keeps that rounding behaviour. */
static inline void oracle_hotloop_c64(zc *col, const zc *Hj, double sr, double si, int64_t lo, int64_t hi) {
  double *c = (double *)col;
  const double *v = (const double *)Hj;
  for (int64_t i = lo; i < hi; ++i) {
    const double vr = v[2 * i], vi = v[2 * i + 1];
    double re = c[2 * i], im = c[2 * i + 1];
    re = fma(-sr, vr, re);
    im = fma(-si, vr, im);
    re = fma(si, vi, re);
    im = fma(-sr, vi, im);
    c[2 * i] = re;
    c[2 * i + 1] = im;
  }
}

/* src:129  norm(view(Hl, j:m, j)) for ComplexF64 -> BLAS dznrm2 (extended-precision sum of the
 * 2(m-j) squared parts on x86-64 OpenBLAS, see dhqr_oracle.c) */
static double oracle_nrm2_c64(const zc *x, int64_t n) {
  const double *d = (const double *)x;
  long double s = 0.0L;
  for (int64_t i = 0; i < 2 * n; ++i) s += (long double)d[i] * (long double)d[i];
  return (double)sqrtl(s);
}

/* src:127-140 for one complex column */
static void oracle_reflector_c64(zc *col, int64_t m, int64_t j, zc *alpha_j, zc *Hj) {
  const double s = oracle_nrm2_c64(col + j, m - j);
  const zc hjj = col[j];
  double h2[2] = {creal(hjj), cimag(hjj)}, af[2];
  dhqr_oracle_alphafactor_c64(h2, af);
  const zc a = mk(s * af[0], s * af[1]);                /* src:130 */
  const double f = 1.0 / sqrt(s * (s + cabs(hjj)));     /* src:131 */
  *alpha_j = a;
  col[j] = mk(creal(hjj) - creal(a), cimag(hjj) - cimag(a)); /* src:132 */
  double *c = (double *)col;
  for (int64_t i = 2 * j; i < 2 * m; ++i) c[i] *= f;    /* src:133-135 */
  memcpy(Hj, col, (size_t)m * sizeof(zc));              /* src:138-140 */
}

/* src:198-213 on columns [col_lo, col_hi) */
void dhqr_oracle_householder_inner_c64(zc *Hl, int64_t m, int64_t ldh, int64_t j, const zc *Hj,
                                       int64_t col_lo, int64_t col_hi, int64_t n) {
  int64_t lo = j + 1 > col_lo ? j + 1 : col_lo;
  int64_t hi = n < col_hi ? n : col_hi;
  if (lo >= hi) return;
#pragma omp parallel for schedule(static)
  for (int64_t jj = lo; jj < hi; ++jj) {
    zc *col = Hl + (size_t)(jj - col_lo) * (size_t)ldh;
    double s[2];
    dhqr_oracle_partialdot_c64(Hj, col, j, m, s);
    oracle_hotloop_c64(col, Hj, s[0], s[1], j, m);
  }
}

/* src:113 + src:122-148 for a plain ComplexF64 matrix */
void dhqr_oracle_householder_c64(zc *H, int64_t m, int64_t n, int64_t ldh, zc *alpha) {
  zc *Hj = (zc *)malloc((size_t)m * sizeof(zc));
  for (int64_t j = 0; j < n; ++j) {
    oracle_reflector_c64(&HZ(0, j), m, j, &alpha[j], Hj);
    dhqr_oracle_householder_inner_c64(H, m, ldh, j, Hj, 0, n, n);
  }
  free(Hj);
}

/* src:215-224 (b <- Q^H b; partialdot conjugates its FIRST argument = the reflector) then
 * src:244-254 (back substitution, complex division by alpha[i]); x = b[1:n] (src:284-294). */
void dhqr_oracle_solve_c64(zc *b, const zc *H, int64_t m, int64_t n, int64_t ldh, const zc *alpha) {
  for (int64_t j = 0; j < n; ++j) {
    const zc *v = &HZ(0, j);
    double s[2];
    dhqr_oracle_partialdot_c64(v, b, j, m, s);
    const zc sz = mk(s[0], s[1]);
    for (int64_t i = j; i < m; ++i) b[i] -= v[i] * sz;
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    zc bi = b[i];
    for (int64_t j = i + 1; j < n; ++j) bi -= HZ(i, j) * b[j];
    b[i] = bi / alpha[i];
  }
}

/* test helper (not in the reference): B = Q*R from the factor format, Q = H_1 ... H_n,
 * H_j = I - v_j v_j^H (Hermitian, so the same application as in the factorisation). */
void dhqr_oracle_form_qr_c64(const zc *H, int64_t m, int64_t n, int64_t ldh, const zc *alpha, zc *B,
                             int64_t ldb) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      B[i + j * ldb] = (i < j) ? HZ(i, j) : (i == j ? alpha[j] : mk(0.0, 0.0));
  for (int64_t j = n - 1; j >= 0; --j) {
    const zc *v = &HZ(0, j);
#pragma omp parallel for schedule(static)
    for (int64_t c = j; c < n; ++c) {
      zc *col = B + (size_t)c * (size_t)ldb;
      double s[2];
      dhqr_oracle_partialdot_c64(v, col, j, m, s);
      oracle_hotloop_c64(col, v, s[0], s[1], j, m);
    }
  }
}
