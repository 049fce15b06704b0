// dhqr_common.h -- device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double dhqr_d4 __attribute__((ext_vector_type(4)));

#define DHQR_NBV 128  // block-reflector width (== DHQR_NB in include/dhqr.h)

// ---- portable counter-based generator; bit-identical to oracle/dhqr_oracle.c::dhqr_oracle_u01
__host__ __device__ __forceinline__ uint64_t dhqr_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double dhqr_u01(uint64_t seed, uint64_t idx) {
  uint64_t z = dhqr_mix64(seed + (idx + 1ULL) * 0x9E3779B97F4A7C15ULL);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// ---- reductions: wavefront xor-shuffle butterfly (every lane ends with the total), then one
// LDS slot per wave.  This is the GPU form of the reference's partialdot accumulation
// (src:42-49, @simd => order unspecified).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// All THREADS threads must call; every thread returns the same total. `red` >= THREADS/64 doubles.
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  constexpr int NW = THREADS / 64;
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` against a previous use
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += red[i];
  return s;
}

// src:8  alphafactor(x::Real) = -sign(x)  (sign(0) == 0 in Julia: a zero pivot gives alpha = -0*s)
__device__ __forceinline__ double dhqr_alphafactor(double x) {
  return x > 0.0 ? -1.0 : (x < 0.0 ? 1.0 : -x);
}
