// dhqr_pack.h -- small streaming kernels shared by every driver: the synthetic fill (the portable counter-based generator
// of dhqr_common.h, identical on host, device and oracle) and the copy of a factored panel into its packed reflector
// operand V (R part zeroed).
#pragma once
#include "dhqr_common.h"

__global__ __launch_bounds__(256) void k_fill_uniform(double *__restrict__ A, int64_t rows,
                                                      int64_t cols, int64_t lda, uint64_t seed,
                                                      int64_t gm, int64_t row0, int64_t cb,
                                                      int nranks, int rank) {
  const int64_t total = rows * cols;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t jl = e / rows, il = e - jl * rows;
    const int64_t gj = ((jl / cb) * nranks + rank) * cb + jl % cb;
    A[il + jl * lda] = dhqr_u01(seed, (uint64_t)(row0 + il + gj * gm));
  }
}

// Pack a factored panel into the clean V operand of the MFMA GEMMs / the broadcast buffer:
// Vw[r + p*ldv] = P[r + p*ldp] for r >= p, p < ncols; 0 above the diagonal (that is R), in the
// zero-padded columns p >= ncols and in the pad rows [rows, ldv).
// `npad` (>= rows) rows of every column are written: rows [rows, npad) are zero padding.
__global__ __launch_bounds__(256) void k_pack_v(const double *__restrict__ P, int64_t ldp,
                                                int64_t rows, int64_t ncols,
                                                double *__restrict__ Vw, int64_t ldv, int64_t npad) {
  const int64_t p = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < npad; r += stride) {
    double x = 0.0;
    if (p < ncols && r >= p && r < rows) x = P[r + p * ldp];
    Vw[r + p * ldv] = x;
  }
}
