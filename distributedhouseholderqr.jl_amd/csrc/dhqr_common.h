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
// One DPP step on a double: every lane of the enabled rows receives the value of its source lane (0.0 in the disabled
// rows / where the pattern has no source), VALU only -- no LDS crossbar round trip like ds_bpermute (__shfl_xor).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  const int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// The same sum with DPP moves instead of ds_bpermute.  quad_perm [1,0,3,2] / [2,3,0,1] and row_ror:4 / :8 leave every
// lane of a 16-lane row with the row's sum; row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the sums
// across the rows into lane 63, which is read back with v_readlane.  For kernels that do many reductions per step
// (k_tsqr_node / k_tsqr_apply: 8 per step and wave -- the butterfly was LDS-throughput bound there, 1.4 - 1.7 x
// slower); the streaming kernels keep the butterfly (k_rank1_fused measured 2 % slower with this one).
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_f64<0xB1, 0xf>(v);
  v += dpp_f64<0x4E, 0xf>(v);
  v += dpp_f64<0x124, 0xf>(v);
  v += dpp_f64<0x128, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);
  v += dpp_f64<0x143, 0xc>(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
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

// The same total with ONE barrier: consecutive calls alternate between two halves of `red` (>= 2 * THREADS/64 doubles;
// `par` is the caller's call counter), so a wave that is already writing the next call's partial sums cannot overwrite
// those a slower wave is still reading -- it cannot be two calls ahead, each call has its barrier.  Wave sums by DPP.
// For kernels whose critical chain is dot product -> total -> update, several times per column (k_rankk_fused).
template <int THREADS>
__device__ __forceinline__ double block_sum_alt(double v, double *red, int &par) {
  v = wave_sum_dpp(v);
  constexpr int NW = THREADS / 64;
  if constexpr (NW == 1) return v;
  double *r = red + (par & 1) * NW;
  ++par;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += r[i];
  return s;
}

// block_sum_alt with a RAW barrier: `__syncthreads()` makes the compiler drain every outstanding vector-memory operation
// first when a direct global -> LDS load is in flight (it writes LDS), i.e. also the next column's prefetch from HBM.  The
// only data this barrier hands from wave to wave are the partial sums written just above (LDS: lgkmcnt); the direct loads
// in flight go to thread-private slots (k_rankk_tall).
template <int THREADS>
__device__ __forceinline__ double block_sum_alt_raw(double v, double *red, int &par) {
  v = wave_sum_dpp(v);
  constexpr int NW = THREADS / 64;
  if constexpr (NW == 1) return v;
  double *r = red + (par & 1) * NW;
  ++par;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += r[i];
  return s;
}

// ---- extended-precision sum of squares for the column norm.  The reference's norm (src:129) is BLAS
// dnrm2 / dznrm2, which OpenBLAS accumulates in x87 extended precision on x86-64 (the CPU test oracle
// restates that with long double).  The reflector of the dominant direction is what the reference's
// acceptance metric ||A^H (A x - b)|| is most sensitive to, so the device keeps the same accuracy:
// a double-double (hi, lo) accumulator built from error-free transforms (TwoSum, FMA TwoProd), carried
// through the wavefront butterfly and the cross-wave sum.  Only the pivot workgroup pays for it.
struct dhqr_dd { double hi, lo; };
// (contraction is switched off inside the transforms: fusing `a + x*x` into an fma would break them)
__device__ __forceinline__ void dd_two_sum(double a, double b, double &s, double &e) {
#pragma clang fp contract(off)
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);  // exact: a + b == s + e
}
__device__ __forceinline__ void dd_add_sq(dhqr_dd &acc, double x) {  // acc += x*x
#pragma clang fp contract(off)
  const double p = x * x;
  const double pe = fma(x, x, -p);  // exact: x*x == p + pe
  double s, e;
  dd_two_sum(acc.hi, p, s, e);
  acc.hi = s;
  acc.lo += e + pe;
}
__device__ __forceinline__ void dd_add_prod(dhqr_dd &acc, double a, double b) {  // acc += a*b
#pragma clang fp contract(off)
  const double p = a * b;
  const double pe = fma(a, b, -p);  // exact: a*b == p + pe
  double s, e;
  dd_two_sum(acc.hi, p, s, e);
  acc.hi = s;
  acc.lo += e + pe;
}
__device__ __forceinline__ void dd_renorm(dhqr_dd &a) {  // hi <- fl(hi + lo), lo <- the rest
#pragma clang fp contract(off)
  const double s = a.hi + a.lo;
  a.lo = a.lo - (s - a.hi);
  a.hi = s;
}
__device__ __forceinline__ dhqr_dd dd_add(const dhqr_dd a, const dhqr_dd b) {  // symmetric in (a, b)
#pragma clang fp contract(off)
  double s, e;
  dd_two_sum(a.hi, b.hi, s, e);
  e += a.lo + b.lo;
  dhqr_dd r;
  r.hi = s + e;
  r.lo = e - (r.hi - s);
  return r;
}
// All THREADS threads call; every thread returns the same (hi + lo rounded once). red >= 2*THREADS/64.
template <int THREADS>
__device__ __forceinline__ double dd_block_sum(dhqr_dd v, double *red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dhqr_dd o;
    o.hi = __shfl_xor(v.hi, off, 64);
    o.lo = __shfl_xor(v.lo, off, 64);
    v = dd_add(v, o);
  }
  constexpr int NW = THREADS / 64;
  if constexpr (NW > 1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) {
      red[2 * w] = v.hi;
      red[2 * w + 1] = v.lo;
    }
    __syncthreads();
    dhqr_dd s;
    s.hi = red[0];
    s.lo = red[1];
#pragma unroll
    for (int i = 1; i < NW; ++i) {
      dhqr_dd o;
      o.hi = red[2 * i];
      o.lo = red[2 * i + 1];
      s = dd_add(s, o);
    }
    v = s;
  }
  return v.hi + v.lo;
}


// sqrt(d) and 1/sqrt(d) of a positive, normally scaled d (the pivots of a panel's Gram matrix) without the range scaling,
// special-case selects and the second division chain of `sqrt(d); 1.0 / r`: v_rsq_f64, one coupled Goldschmidt step and two
// residual corrections -- the compiler's own sequence for the square root, same result -- and the reciprocal root falls
// out of it with one Newton step.  On the critical chain of k_panel_top (dhqr_recon.h) the row owners execute ~110
// dependent FP64 instructions per elimination step while fifteen waves wait at the barrier; this halves them.
// inf / NaN propagate (a broken panel is rejected by its verification).
__device__ __forceinline__ void dhqr_sqrt_rsqrt(double d, double &r, double &rinv) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-g, g, d);
  g = fma(e, h, g);
  e = fma(-g, g, d);
  g = fma(e, h, g);
  double x = h + h;
  e = fma(-g, x, 1.0);
  x = fma(x, e, x);
  r = g;
  rinv = x;
}
// 1/x to within an ulp for a normally scaled x: v_rcp_f64 + two Newton steps (no div_scale / div_fmas / div_fixup)
__device__ __forceinline__ double dhqr_rcp(double x) {
  double z = __builtin_amdgcn_rcp(x);
  double e = fma(-x, z, 1.0);
  z = fma(z, e, z);
  e = fma(-x, z, 1.0);
  z = fma(z, e, z);
  return z;
}

// src:8  alphafactor(x::Real) = -sign(x)  (sign(0) == 0 in Julia: a zero pivot gives alpha = -0*s)
__device__ __forceinline__ double dhqr_alphafactor(double x) {
  return x > 0.0 ? -1.0 : (x < 0.0 ? 1.0 : -x);
}

// ---- inter-workgroup hand-over flags of the column pipelines (k_zpanel_pipe, rankk_lead_pipe) ----------------------
// One array of 128 flags (one 128-byte line each) + one error word per context.  A flag holds the number (epoch) of the
// launch that last raised it, so the flags are never reset.  INVARIANTS the drivers keep and the kernels rely on:
//   * the flag-using launches of a context never overlap (they are all issued on ONE stream at a time: the look-ahead
//     lane for ComplexF64 panels, the caller's stream for the unblocked path) -- and should two ever overlap, a flag is
//     raised with an atomic MAX, so a late store of epoch e cannot take e + 1 back and strand a waiter;
//   * a workgroup only waits for LOWER-indexed workgroups of its own launch, which the hardware dispatches first (observed,
//     not promised by HIP).  Should that ever fail -- or a predecessor die -- the wait is BOUNDED: after
//     DHQR_PIPE_SPIN_LIMIT polls the waiter records the epoch in the error word and goes on (wrong numbers instead of a
//     hung GPU); the host reports it at its next synchronising entry point (pipe_error_check in dhqr_api.hip) and names
//     the kill switches DHQR_ZPIPE=0 / DHQR_RANKK_PIPE=0 (one launch per column, no inter-workgroup waits).
#define DHQR_PIPE_FLAG_STRIDE 32
#define DHQR_PIPE_NFLAGS 128
#define DHQR_PIPE_ERR_OFFSET (DHQR_PIPE_NFLAGS * DHQR_PIPE_FLAG_STRIDE)
#define DHQR_PIPE_LIMIT_OFFSET (DHQR_PIPE_ERR_OFFSET + 1)  // the waiters' poll bound (same 128-byte line as the error word)
#define DHQR_PIPE_INTS (DHQR_PIPE_ERR_OFFSET + DHQR_PIPE_FLAG_STRIDE)
#ifndef DHQR_PIPE_SPIN_LIMIT
#define DHQR_PIPE_SPIN_LIMIT (1 << 24)  // x (one L2 poll + s_sleep 1) ~ several seconds; a hand-over takes microseconds
#endif
// called by ONE thread of the waiting workgroup: relaxed polls, then one acquire fence (an acquire LOAD at agent scope
// would invalidate the XCD's L2 on every iteration)
// The bound is a word of the flag block (DHQR_PIPE_LIMIT_OFFSET; written once by dhqr_create: DHQR_PIPE_SPIN_LIMIT, or
// the environment variable of that name -- tests/test_gpu_kernels.py sets it to 1 to exercise the reporting path).
__device__ __forceinline__ void dhqr_pipe_wait(int *flags, int idx, int epoch) {
  int spins = 0;
  const int limit = flags[DHQR_PIPE_LIMIT_OFFSET];
  while (__hip_atomic_load(flags + idx * DHQR_PIPE_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > limit) {
      __hip_atomic_store(flags + DHQR_PIPE_ERR_OFFSET, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// called by ONE thread after the workgroup's stores (and a barrier): L2 write-back, then the flag
__device__ __forceinline__ void dhqr_pipe_raise(int *flags, int idx, int epoch) {
  __hip_atomic_fetch_max(flags + idx * DHQR_PIPE_FLAG_STRIDE, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
