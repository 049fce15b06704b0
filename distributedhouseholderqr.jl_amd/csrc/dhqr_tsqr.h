// dhqr_tsqr.h -- tall-skinny panels by a TSQR tree with Householder reconstruction (Demmel et
// al.'s TSQR, Ballard et al.'s TSQR-HR).  Included by dhqr_api.hip.
//
// The fast panel path gets R from the Gram matrix (R = chol(P'P)) and the reflectors from V = (P - alpha E) M^{-1}:
// one pass over the panel, but both steps lose accuracy with kappa(P), so ill-conditioned panels are rejected by the
// verification.  Here the panel's rows are cut into 256-row leaves, every leaf is factored by an unblocked Householder
// QR in registers (k_tsqr_node, one workgroup per leaf), and pairs of 128 x 128 R factors are stacked and factored
// again, level by level, until one R is left:
//       leaves          R_0   R_1   R_2   R_3   ...          (rows / 256 workgroups)
//       level 1           R_01        R_23      ...
//       ...                     R_t
// Every node keeps its reflectors.  Going back down the tree (k_tsqr_apply: H_1 ... H_128 [C; 0] per node, starting
// from the identity at the root) gives the EXPLICIT orthonormal factor Q (rows x 128), P = Q R_t, with the accuracy of
// a Householder QR whatever kappa(P) is.  Q then goes through the replay / reconstruction of dhqr_recon.h with the
// trivially known R(Q) = I -- perfectly conditioned -- which yields the reference's reflectors V (src:122-148) and the
// signs D = diag(alpha(Q)) = +-1;  R = D R_t and alpha = diag(R) complete the reference's factor format.
// With the rows split over ranks (dhqr_rowsplit.h) every rank reduces its own leaves, the P local R factors are
// gathered (one all-reduce of a zero-padded P x 128 x 128 buffer), every rank runs the same tree over them (up and
// down: bit-identical everywhere, no further exchange) and continues down its own subtree from its block.
// Use: second rung of the fallback ladder (Gram/Cholesky -> TSQR-HR -> column by column), which avoids the
// per-column collectives of the last rung, and R source of every panel with DHQR_TSQR=1 / dhqr_set_r_source(ctx, 3).
#pragma once
#include "dhqr_common.h"

#define TSQR_LEAF 256  // rows per leaf = rows of a stacked pair of R factors

// One tree node: Rout[b] = R of X_b (<= 256 rows x 128 columns), Householder QR with the reference's rule applied to
// the node's own pivots (R_jj = -sign(a_jj) ||a_j||, src:129-131; a zero column gives alpha = 0 and no reflection,
// src:8).  The root's R therefore equals the panel's R up to the SIGN of each row; k_recon_top takes any row signs.
//   leaf mode (Rin == nullptr): X_b = rows [256 b, min(rows, 256 b + 256)) of A (leading dimension lda)
//   pair mode: X_b = [Rin[2b]; Rin[2b+1]] (128 x 128 each, leading dimension 128; the second one is absent when
//              2b + 1 == count)
// 1024 threads = 16 waves.  Wave w owns the columns w + 16 y (y < 8), lane l the rows l + 64 x (x < 4): a column's dot
// with the reflector is a wave reduction (no LDS), and the only workgroup barrier of a step is the one that publishes
// the next reflector.  The wave that owns column j+1 updates it first, builds reflector j+1 and publishes it, then
// updates its other columns: the next step's reflector is ready when the barrier of the current step opens.
// Yout != nullptr: the node's reflectors are kept (leaf b: rows [256 b, 256 b + 256) of the column-major matrix Yout
// with leading dimension ldy; pair node b: the 256 x 128 block Yout + b * 256 * 128, leading dimension 256).
__global__ __launch_bounds__(1024) void k_tsqr_node(const double *__restrict__ A, int64_t lda, int64_t rows,
                                                     const double *__restrict__ Rin, int count,
                                                     double *__restrict__ Rout, double *__restrict__ Yout, int64_t ldy) {
  __shared__ double vbuf[2][TSQR_LEAF];
  __shared__ double alpha_s[128];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t b = blockIdx.x;
  double a[4][8];
  double *Y = nullptr;
  int64_t ly = 0;
  if (Yout != nullptr) {
    Y = (Rin == nullptr) ? Yout + b * TSQR_LEAF : Yout + b * TSQR_LEAF * 128;
    ly = (Rin == nullptr) ? ldy : TSQR_LEAF;
  }
  if (Rin == nullptr) {
    const double *X = A + b * TSQR_LEAF;
    const int64_t nr = (rows - b * TSQR_LEAF < TSQR_LEAF) ? rows - b * TSQR_LEAF : TSQR_LEAF;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        const int i = lane + 64 * x, k = w + 16 * y;
        a[x][y] = (i < nr) ? X[i + (int64_t)k * lda] : 0.0;
      }
  } else {
    const double *R0 = Rin + (2 * b) * 128 * 128;
    const bool two = 2 * b + 1 < count;
    const double *R1 = Rin + (2 * b + (two ? 1 : 0)) * 128 * 128;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        const int i = lane + 64 * x, k = w + 16 * y;
        // an R factor is upper triangular: rows below the diagonal are zero by construction, not read
        if (i < 128) a[x][y] = (i <= k) ? R0[i + k * 128] : 0.0;
        else a[x][y] = (two && i - 128 <= k) ? R1[(i - 128) + k * 128] : 0.0;
      }
  }

  // reflector of column j (held by this wave as register column y): v -> vbuf[j & 1], alpha -> alpha_s[j]
  auto build = [&](int j, int y) {
    double s = 0.0;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int i = lane + 64 * x;
      const double z = (i >= j) ? a[x][y] : 0.0;
      s = fma(z, z, s);
    }
    s = wave_sum_dpp(s);
    double pj = 0.0;  // a_jj lives in lane j % 64, register row j / 64
#pragma unroll
    for (int x = 0; x < 4; ++x)
      if ((j >> 6) == x) pj = __shfl(a[x][y], j & 63, 64);
    const double nrm = sqrt(s);
    const double al = nrm * dhqr_alphafactor(pj);                             // src:130
    const double f = (nrm > 0.0) ? 1.0 / sqrt(nrm * (nrm + fabs(pj))) : 0.0;  // src:131; zero column: no reflection
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int i = lane + 64 * x;
      double v = (i > j) ? a[x][y] * f : 0.0;
      if (i == j) v = (pj - al) * f;                                          // src:132-135
      vbuf[j & 1][i] = v;
      if (Y != nullptr) Y[i + (int64_t)j * ly] = v;
    }
    if (lane == 0) alpha_s[j] = al;
  };
  // column y of this wave -= v (v' column)   (src:208-209)
  auto apply = [&](const double (&v)[4], int y) {
    double d = 0.0;
#pragma unroll
    for (int x = 0; x < 4; ++x) d = fma(v[x], a[x][y], d);
    d = wave_sum_dpp(d);
#pragma unroll
    for (int x = 0; x < 4; ++x) a[x][y] = fma(-v[x], d, a[x][y]);
  };

  if (w == 0) build(0, 0);
  __syncthreads();
  for (int j = 0; j < 128; ++j) {
    double v[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) v[x] = vbuf[j & 1][lane + 64 * x];
    const int jn = j + 1, wn = jn & 15, yn = jn >> 4;
    if (jn < 128 && w == wn) {  // owner of the next column: update it, publish its reflector, then the rest
#pragma unroll
      for (int y = 0; y < 8; ++y)
        if (y == yn) {
          apply(v, y);
          build(jn, y);
        }
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int k = w + 16 * y;
      if (k > j && !(w == wn && y == yn && jn < 128)) apply(v, y);
    }
    __syncthreads();
  }

  double *Ro = Rout + b * 128 * 128;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int i = lane + 64 * x, k = w + 16 * y;
      Ro[i + k * 128] = (k > i) ? a[x][y] : (k == i ? alpha_s[i] : 0.0);
    }
}

// One node on the way DOWN the tree: out = H_1 H_2 ... H_128 [C; 0] (256 x 128) with the node's stored reflectors
// (Y as written by k_tsqr_node) and C = Cin[b] (128 x 128, leading dimension 128; Cin == nullptr: the identity, the
// root).  pair mode (leaf == 0): the two 128-row halves of `out` are the C blocks of the children 2b and 2b+1 (the
// second only if it exists: 2b + 1 < nchild) -> Cout[2b], Cout[2b+1].  leaf mode: `out` is this leaf's 256 rows of
// the explicit Q -> rows [256 b, 256 b + 256) of Q (leading dimension ldq; may alias Y: every read precedes the
// writes).  Same thread map as k_tsqr_node; the waves are independent here (no barrier at all).
__global__ __launch_bounds__(1024) void k_tsqr_apply(const double *__restrict__ Cin, const double *Yin, int64_t ldy, int leaf,
                                                      int nchild, double *__restrict__ Cout, double *Q, int64_t ldq) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t b = blockIdx.x;
  const double *Y = leaf ? Yin + b * TSQR_LEAF : Yin + b * TSQR_LEAF * 128;
  const int64_t ly = leaf ? ldy : TSQR_LEAF;
  double a[4][8];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int i = lane + 64 * x, k = w + 16 * y;
      if (i >= 128) a[x][y] = 0.0;
      else a[x][y] = (Cin == nullptr) ? (i == k ? 1.0 : 0.0) : Cin[b * 128 * 128 + i + k * 128];
    }
  for (int j = 127; j >= 0; --j) {
    double v[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) v[x] = Y[(lane + 64 * x) + (int64_t)j * ly];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      double d = 0.0;
#pragma unroll
      for (int x = 0; x < 4; ++x) d = fma(v[x], a[x][y], d);
      d = wave_sum_dpp(d);
#pragma unroll
      for (int x = 0; x < 4; ++x) a[x][y] = fma(-v[x], d, a[x][y]);
    }
  }
  if (leaf) {
    __syncthreads();  // Q may alias Y: every wave has finished reading the reflectors
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) Q[b * TSQR_LEAF + (lane + 64 * x) + (int64_t)(w + 16 * y) * ldq] = a[x][y];
  } else {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        const int i = lane + 64 * x, k = w + 16 * y;
        const int64_t child = 2 * b + (i >> 7);
        if (child < nchild) Cout[child * 128 * 128 + (i & 127) + k * 128] = a[x][y];
      }
  }
}

// R and alpha of the panel from the tree's R_t and the signs of Q's reflection: R = D R_t, D = diag(alpha(Q)) (+-1).
// Rref: strict upper part (dense 128 x 128, zeros elsewhere), alpha_out: diag(R).
__global__ __launch_bounds__(256) void k_tsqr_final_r(const double *__restrict__ Rt, const double *__restrict__ alphaq,
                                                       double *__restrict__ Rref, double *__restrict__ alpha_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // 128 * 128 elements
  const int i = e & 127, k = e >> 7;
  const double d = alphaq[i];
  Rref[e] = (k > i) ? d * Rt[e] : 0.0;
  if (i == k) alpha_out[i] = d * Rt[e];
}
// The reflectors of P = Q R_t are those of Q up to sign: after j-1 reflections the active part of column j of P is
// (R_t)_jj times that of Q, so v_j(P) = sign((R_t)_jj) v_j(Q) (src:130-135: alpha and the pivot change sign together).
// Folded into the reconstruction operator: column j of -M^{-1} is scaled by that sign before V = Q M^{-1}.
__global__ __launch_bounds__(256) void k_tsqr_sign_cols(double *__restrict__ negMinv, const double *__restrict__ Rt) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int j = e >> 7;
  if (Rt[j + j * 128] < 0.0) negMinv[e] = -negMinv[e];
}
__global__ __launch_bounds__(256) void k_tsqr_identity(double *__restrict__ I) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  I[e] = ((e & 127) == (e >> 7)) ? 1.0 : 0.0;
}
