// dhqr_rank1.h -- HBM-bound streaming kernels of the unblocked path (BASELINE config 2) and of
// the in-panel factorisation: synthetic fill, reflector construction, fused rank-1 update.
//
// Reference mapping (src/DistributedHouseholderQR.jl):
//   k_reflector        src:129-140  (norm, alpha, f, scale, copy column into Hj)
//   k_rank1_fused/_generic
//                      src:198-213 + src:42-49 + src:156-160 for every trailing column of step j,
//                      PLUS src:129-140 for column j+1 done by the workgroup that owns it, so one
//                      launch per column replaces the reference's norm / scale / copy / @spawnat
//                      sequence.  The trailing column stays in registers between the dot and the
//                      update: HBM sees one read and one write per element (16 B / element /
//                      reflector, the algorithmic traffic of SURVEY.md section 8d).
//   `vcur`/`vnext`     the reference's dense `Hj` staging vector (src:125,138-140), double
//                      buffered; entries above the diagonal are kept at 0.
#pragma once
#include "dhqr_common.h"

// One workgroup builds the reflector of column j from scratch (first column of a matrix/panel).
// col = &A[0 + j*lda].  Writes the scaled v in place, vnext[0:m] (zeros above the diagonal), alpha.
template <int T>
__global__ __launch_bounds__(T) void k_reflector(double *__restrict__ col, int64_t m, int64_t j,
                                                 double *__restrict__ vnext,
                                                 double *__restrict__ alpha_j) {
  __shared__ double red[2 * (T / 64) + 2];
  const int t = threadIdx.x;
  const double h = col[j];
  dhqr_dd acc = {0.0, 0.0};  // double-double sum of squares: the reference's dnrm2 is extended precision too
  for (int64_t i = j + t; i < m; i += T) dd_add_sq(acc, col[i]);
  const double s2 = dd_block_sum<T>(acc, red);
  const double s = sqrt(s2);                       // src:129
  const double al = s * dhqr_alphafactor(h);       // src:130
  const double f = 1.0 / sqrt(s * (s + fabs(h)));  // src:131
  for (int64_t i = t; i < m; i += T) {
    double val = 0.0;
    if (i >= j) {
      val = (i == j ? h - al : col[i]) * f;  // src:132-135
      col[i] = val;
    }
    vnext[i] = val;  // src:138-140
  }
  if (t == 0) *alpha_j = al;
}

// Fused step j, register-resident variant: workgroup b owns trailing column c = j+1+b.
// T threads x EPT doubles cover rows [r0, r0 + T*EPT) with r0 = j (VEC=1) or j rounded down to
// even (VEC=2, 16-byte loads; needs lda, m even and 16-byte aligned bases).  Loads are
// unconditional (clamped address + select) so all EPT/VEC loads of a thread are in flight at once.
template <int T, int EPT, int VEC>
__global__ __launch_bounds__(T) void k_rank1_fused(double *__restrict__ A, int64_t lda, int64_t m,
                                                   int64_t j, const double *__restrict__ vcur,
                                                   double *__restrict__ vnext,
                                                   double *__restrict__ alpha) {
  __shared__ double red[2 * (T / 64) + 2];
  constexpr int HSLOT = 2 * (T / 64);  // pivot element of the next column
  const int t = threadIdx.x;
  const int64_t c = j + 1 + blockIdx.x;
  double *__restrict__ col = A + c * lda;
  const int64_t r0 = (VEC == 2) ? (j & ~(int64_t)1) : j;
  const int64_t mlast = m - VEC;  // last valid (pair) start
  double a[EPT], v[EPT];

  if constexpr (VEC == 2) {
#pragma unroll
    for (int i = 0; i < EPT / 2; ++i) {
      const int64_t row = r0 + 2 * ((int64_t)t + (int64_t)i * T);
      const bool ok = row < m;
      const int64_t rc = ok ? row : mlast;
      const double2 x = *reinterpret_cast<const double2 *>(col + rc);
      const double2 y = *reinterpret_cast<const double2 *>(vcur + rc);
      a[2 * i] = ok ? x.x : 0.0; a[2 * i + 1] = ok ? x.y : 0.0;
      v[2 * i] = ok ? y.x : 0.0; v[2 * i + 1] = ok ? y.y : 0.0;
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = r0 + t + (int64_t)e * T;
      const bool ok = row < m;
      const int64_t rc = ok ? row : mlast;
      const double x = col[rc], y = vcur[rc];
      a[e] = ok ? x : 0.0;
      v[e] = ok ? y : 0.0;
    }
  }

  double dot = 0.0;  // src:208 partialdot(Hj, view(Hl,:,jj), j:m)
#pragma unroll
  for (int e = 0; e < EPT; ++e) dot = fma(a[e], v[e], dot);
  const double s = block_sum<T>(dot, red);
#pragma unroll
  for (int e = 0; e < EPT; ++e) a[e] = fma(-v[e], s, a[e]);  // src:209 hotloop!

  const bool pivot = (blockIdx.x == 0);  // this workgroup owns column j+1: build its reflector
  if (pivot) {
    const int64_t jp = j + 1;
    dhqr_dd acc = {0.0, 0.0};  // extended-precision column norm (src:129: dnrm2), pivot workgroup only
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = (VEC == 2) ? r0 + 2 * ((int64_t)t + (int64_t)(e >> 1) * T) + (e & 1)
                                     : r0 + t + (int64_t)e * T;
      if (row == jp) red[HSLOT] = a[e];
      if (row >= jp && row < m) dd_add_sq(acc, a[e]);
    }
    const double s2 = dd_block_sum<T>(acc, red);  // barriers inside also publish red[HSLOT]
    const double h = red[HSLOT];
    const double sn = sqrt(s2);                        // src:129
    const double al = sn * dhqr_alphafactor(h);        // src:130
    const double f = 1.0 / sqrt(sn * (sn + fabs(h)));  // src:131
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = (VEC == 2) ? r0 + 2 * ((int64_t)t + (int64_t)(e >> 1) * T) + (e & 1)
                                     : r0 + t + (int64_t)e * T;
      if (row == jp) a[e] = (h - al) * f;  // src:132-135
      else if (row > jp) a[e] *= f;
      v[e] = (row >= jp) ? a[e] : 0.0;  // reuse v[] as the outgoing Hj (src:138-140)
    }
    if (t == 0) alpha[jp] = al;
  }

  if constexpr (VEC == 2) {
#pragma unroll
    for (int i = 0; i < EPT / 2; ++i) {
      const int64_t row = r0 + 2 * ((int64_t)t + (int64_t)i * T);
      if (row < m) {
        *reinterpret_cast<double2 *>(col + row) = make_double2(a[2 * i], a[2 * i + 1]);
        if (pivot) *reinterpret_cast<double2 *>(vnext + row) = make_double2(v[2 * i], v[2 * i + 1]);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = r0 + t + (int64_t)e * T;
      if (row < m) {
        col[row] = a[e];
        if (pivot) vnext[row] = v[e];
      }
    }
  }
}

// LDS slots for the reflectors the lead workgroup of k_rankk_fused builds (beside the K - 3 reflectors of the pass that
// live in LDS): as many of the K - 1 as fit 144 KiB, for columns of at most 6144 rows -- the launches whose duration is
// the lead's chain.  Such launches run ONE workgroup per CU (launch_rankk).  Measured at 8192^2 (ms): none 147.7,
// <= 3072 rows 147.0, <= 4096 144.3, <= 6144 143.7.
#ifndef DHQR_RK_LEAD_ROWS
#define DHQR_RK_LEAD_ROWS 6144
#endif
// Most reflectors a pass of k_rankk_fused<T, EPT> can keep on the CU: three in registers + as many as fit 152 KiB of LDS
// (r4: columns of at most 6144 rows take 6, of at most 4096 rows 7, of at most 3072 rows 8 -- the bytes per element and
// reflector fall from 16/5 to 16/K there).
#define DHQR_RK_KMAX 8
// reflectors of the pass held in REGISTERS: three beside three column buffers of 8 elements per thread (1024 threads, 128
// registers each); four in the 16-elements-per-thread instantiations (<= 512 threads: 256 registers each), which exist so
// that columns of 6145 ... 8192 rows -- 58 % of the traffic of an 8192^2 factorisation -- get a sixth reflector per pass
constexpr int rankk_kr(int EPT, int K) { return EPT >= 16 ? (K < 4 ? K : 4) : (K < 3 ? K : 3); }
constexpr int rankk_fit(int T, int EPT) {
  const int k = rankk_kr(EPT, DHQR_RK_KMAX) + (152 * 1024) / (T * EPT * 8);
  return k > DHQR_RK_KMAX ? DHQR_RK_KMAX : k;
}
constexpr int rankk_lead_slots(int T, int EPT, int K) {
  const int KL = K - rankk_kr(EPT, K);
  if (T * EPT > DHQR_RK_LEAD_ROWS) return 0;
  const int avail = (144 * 1024) / (T * EPT * 8) - KL;
  return avail < 0 ? 0 : (avail < K - 1 ? avail : K - 1);
}

// The LEAD workgroup of k_rankk_fused (see there).  NOT inlined: inside the kernel the register allocation is shaped by the
// bulk (three column buffers + three reflectors), and with the norm's temporaries on top the 1024-thread instantiations
// kept the reflectors in scratch and re-read them for every apply (26 us per lead column at 8192 rows).  Here the old
// reflectors are not held at all: the first three stream from `vold` (L2) through two buffers, one load ahead of the
// apply that uses them; reflectors 4, 5 and -- where they fit -- the ones built here sit in LDS (`vl`, passed in
// together with the reduction scratch; generic pointers, so LDS is reached by flat instructions).
template <int T, int EPT, int VEC, int K>
__device__ __forceinline__ void rankk_lead_body(double *__restrict__ A, int64_t lda, int64_t m, int64_t ncols,
                                                int64_t c0, int64_t rtop, int kold, const double *vold,
                                                double *vnew, int64_t vlen, double *__restrict__ alpha,
                                                double *red, double *reda, double *vl) {
  constexpr int KR = rankk_kr(EPT, K);
  constexpr int KL = K - KR;
  constexpr int NN = rankk_lead_slots(T, EPT, K);
  constexpr int HSLOT = 2 * (T / 64);
  const int t = threadIdx.x;
  const int64_t mlast = m - VEC;
  int par = 0;
  double a[EPT], w0[EPT], w1[EPT];

  auto row_of = [&](int e) -> int64_t {
    return (VEC == 2) ? rtop + 2 * ((int64_t)t + (int64_t)(e >> 1) * T) + (e & 1) : rtop + t + (int64_t)e * T;
  };
  auto load = [&](const double *src, double *dst, bool mask) {  // as in the kernel: reflectors masked, columns not
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const int64_t row = rtop + 2 * ((int64_t)t + (int64_t)i * T);
        const bool ok = row < m;
        const double2 x = *reinterpret_cast<const double2 *>(src + (ok ? row : mlast));
        dst[2 * i] = (ok || !mask) ? x.x : 0.0;
        dst[2 * i + 1] = (ok || !mask) ? x.y : 0.0;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t row = rtop + t + (int64_t)e * T;
        const bool ok = row < m;
        const double x = src[ok ? row : mlast];
        dst[e] = (ok || !mask) ? x : 0.0;
      }
    }
  };
  auto store = [&](double *dst, const double *src) {
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const int64_t row = rtop + 2 * ((int64_t)t + (int64_t)i * T);
        if (row < m) *reinterpret_cast<double2 *>(dst + row) = make_double2(src[2 * i], src[2 * i + 1]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t row = rtop + t + (int64_t)e * T;
        if (row < m) dst[row] = src[e];
      }
    }
  };
  auto lds_at = [&](int q, int e) -> double * {
    return (VEC == 2) ? vl + (size_t)q * T * EPT + 2 * ((size_t)t + (size_t)(e >> 1) * T) + (e & 1)
                      : vl + (size_t)q * T * EPT + (size_t)t + (size_t)e * T;
  };
  auto apply = [&](const double *x) {  // src:208 partialdot, src:209 hotloop! on the column in a[]
    double dot = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) dot = fma(a[e], x[e], dot);
    const double s = block_sum_alt<T>(dot, reda, par);
#pragma unroll
    for (int e = 0; e < EPT; ++e) a[e] = fma(-x[e], s, a[e]);
  };
  auto apply_lds = [&](int q) {
    double dot = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) dot = fma(a[e], *lds_at(q, e), dot);
    const double s = block_sum_alt<T>(dot, reda, par);
#pragma unroll
    for (int e = 0; e < EPT; ++e) a[e] = fma(-*lds_at(q, e), s, a[e]);
  };

  const int nown = (int)((ncols - c0 < K) ? (ncols - c0) : K);
  load(A + c0 * lda, a, false);
#pragma unroll
  for (int q = 0; q < KL; ++q)
    if (KR + q < kold) {
      load(vold + (int64_t)(KR + q) * vlen, w0, true);
#pragma unroll
      for (int e = 0; e < EPT; ++e) *lds_at(q, e) = w0[e];  // read back by the same thread only
    }
  const int kreg = kold < KR ? kold : KR;  // old reflectors that are not in LDS
  for (int q = 0; q < nown; ++q) {
    const int64_t c = c0 + q;
    double *col = A + c * lda;
    if (q > 0) load(col, a, false);
    if (kreg > 0) load(vold, w0, true);
    for (int p = 0; p < kreg; p += 2) {
      if (p + 1 < kreg) load(vold + (int64_t)(p + 1) * vlen, w1, true);
      apply(w0);
      if (p + 1 < kreg) {
        if (p + 2 < kreg) load(vold + (int64_t)(p + 2) * vlen, w0, true);
        apply(w1);
      }
    }
#pragma unroll
    for (int ql = 0; ql < KL; ++ql)
      if (KR + ql < kold) apply_lds(ql);
    for (int p = 0; p < q; ++p) {  // the reflectors built in this launch
      if (p < NN) {
        apply_lds(KL + p);
      } else {
        load(vnew + (int64_t)p * vlen, w0, true);
        apply(w0);
      }
    }
    dhqr_dd acc = {0.0, 0.0};  // extended-precision column norm (src:129: dnrm2)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = row_of(e);
      if (row == c) red[HSLOT] = a[e];
      if (row >= c && row < m) dd_add_sq(acc, a[e]);
    }
    const double sq = dd_block_sum<T>(acc, red);  // barriers inside also publish red[HSLOT]
    const double h = red[HSLOT];
    const double sn = sqrt(sq);                        // src:129
    const double al = sn * dhqr_alphafactor(h);        // src:130
    const double f = 1.0 / sqrt(sn * (sn + fabs(h)));  // src:131
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = row_of(e);
      if (row == c) a[e] = (h - al) * f;  // src:132-135
      else if (row > c) a[e] *= f;
      w0[e] = (row >= c && row < m) ? a[e] : 0.0;  // outgoing Hj (src:138-140), zero beyond the column
    }
    if (t == 0) alpha[c] = al;
    store(vnew + (int64_t)q * vlen, w0);
    if (q < NN) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) *lds_at(KL + q, e) = w0[e];  // read back by the same thread only
    }
    store(col, a);
    __syncthreads();  // red[HSLOT] is rewritten for the next column
  }
}

// (workgroups of at most 512 threads have 256 registers per thread and no such problem: they inline the body and save the
// call's register saves -- 1024^2: 5.75 -> 5.4 ms)
template <int T, int EPT, int VEC, int K>
__device__ __attribute__((noinline)) void rankk_lead(double *__restrict__ A, int64_t lda, int64_t m, int64_t ncols,
                                                     int64_t c0, int64_t rtop, int kold, const double *vold,
                                                     double *vnew, int64_t vlen, double *__restrict__ alpha,
                                                     double *red, double *reda, double *vl) {
  rankk_lead_body<T, EPT, VEC, K>(A, lda, m, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, red, reda, vl);
}

__device__ __forceinline__ uint32_t rk_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// The thread index as a value the optimiser cannot hoist: in k_rankk_tall every load address is loop invariant, LICM formed
// 16 + 16 full 64-bit addresses in front of the column loop, the register allocator spilled them, and every global load
// then waited (s_waitcnt vmcnt(0)) for the scratch reload of ITS address -- one load in flight at a time (ISA of the
// 512 x 32 instantiation: 152 scratch loads in the loop; 16384-row columns ran at 1.8 - 3.1 TB/s against 4.5 at 12288 rows).
// With an opaque copy per step the offsets are recomputed where they are used: two VALU instructions per 16-byte load.
__device__ __forceinline__ uint32_t rk_opaque(uint32_t t) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(t));
#endif
  return t;
}

// The lead of k_rankk_tall, PIPELINED over K workgroups: workgroup q owns column c0 + q.  All K apply the pass's old
// reflectors at the same time (streamed from L2 through two buffers), then workgroup q waits for the reflectors
// v_c0 .. v_c0+q-1 of its predecessors (flag p = epoch: stored with release at agent scope once the owner's stores have
// reached its L2, polled with acquire by one thread -- the mechanism of k_zpanel_pipe, dhqr_complex.h; a workgroup only
// waits for lower-indexed ones), applies them, builds its own and publishes it.  The arithmetic and its order are those of
// the one-workgroup lead (rankk_lead_body): one workgroup needed K x (K + q) applies of 128 KiB each in sequence (~300 us
// per launch at 16384 rows, three times the bulk's traffic time); here the chain is K old applies + K hand-overs.
#define DHQR_RK_FLAG_STRIDE DHQR_PIPE_FLAG_STRIDE  // ints between two flags: one 128-byte line each
template <int T, int EPT, int VEC, int K>
__device__ __attribute__((noinline)) void rankk_lead_pipe(double *__restrict__ A, int64_t lda, int64_t m, int64_t ncols, int64_t c0,
                                                int64_t rtop, int kold, const double *vold, double *vnew, int64_t vlen,
                                                double *__restrict__ alpha, double *red, double *reda, int *flags,
                                                int epoch, int q) {
  constexpr int HSLOT = 2 * (T / 64);
  const int64_t c = c0 + q;
  if (c >= ncols) return;
  const uint32_t t = threadIdx.x;
  const uint32_t span = (uint32_t)(m - rtop), olast = span - VEC;
  int par = 0;
  double a[EPT], w0[EPT], w1[EPT];
  auto row_of = [&](int e) -> int64_t {
    return (VEC == 2) ? rtop + 2 * ((int64_t)t + (int64_t)(e >> 1) * T) + (e & 1) : rtop + t + (int64_t)e * T;
  };
  auto load_col = [&](const double *src, double *dst) {  // clamped, not masked (see k_rankk_fused)
    const double *base = src + rtop;
    const uint32_t tt = rk_opaque(t);  // offsets recomputed per call, not hoisted and spilled (rk_opaque)
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const double2 x = *reinterpret_cast<const double2 *>(base + rk_umin(2u * (tt + (uint32_t)i * T), olast));
        dst[2 * i] = x.x;
        dst[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) dst[e] = base[rk_umin(tt + (uint32_t)e * T, olast)];
    }
  };
  auto load_refl = [&](const double *src, double *dst) {  // zero-padded slots: no clamp, no mask
    const double *base = src + rtop;
    const uint32_t tt = rk_opaque(t);
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const double2 x = *reinterpret_cast<const double2 *>(base + 2u * (tt + (uint32_t)i * T));
        dst[2 * i] = x.x;
        dst[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) dst[e] = base[tt + (uint32_t)e * T];
    }
  };
  auto store = [&](double *dst, const double *src) {
    double *base = dst + rtop;
    const uint32_t tt = rk_opaque(t);
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const uint32_t o = 2u * (tt + (uint32_t)i * T);
        if (o < span) *reinterpret_cast<double2 *>(base + o) = make_double2(src[2 * i], src[2 * i + 1]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const uint32_t o = tt + (uint32_t)e * T;
        if (o < span) base[o] = src[e];
      }
    }
  };
  auto apply = [&](const double *x) {  // src:208 partialdot, src:209 hotloop! on the column in a[]
    double dot = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) dot = fma(a[e], x[e], dot);
    const double s = block_sum_alt<T>(dot, reda, par);
#pragma unroll
    for (int e = 0; e < EPT; ++e) a[e] = fma(-x[e], s, a[e]);
  };
  double *col = A + c * lda;
  load_col(col, a);
  if (kold > 0) load_refl(vold, w0);
  for (int p = 0; p < kold; p += 2) {  // the pass's reflectors, one load ahead of the apply that uses it
    if (p + 1 < kold) load_refl(vold + (int64_t)(p + 1) * vlen, w1);
    apply(w0);
    if (p + 1 < kold) {
      if (p + 2 < kold) load_refl(vold + (int64_t)(p + 2) * vlen, w0);
      apply(w1);
    }
  }
  for (int p = 0; p < q; ++p) {  // the reflectors of this launch's earlier columns, as their owners publish them
    if (t == 0) dhqr_pipe_wait(flags, p, epoch);  // relaxed polls, one acquire fence, bounded (dhqr_common.h)
    __syncthreads();
    load_refl(vnew + (int64_t)p * vlen, w0);
    apply(w0);
  }
  dhqr_dd acc = {0.0, 0.0};  // extended-precision column norm (src:129: dnrm2)
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t row = row_of(e);
    if (row == c) red[HSLOT] = a[e];
    if (row >= c && row < m) dd_add_sq(acc, a[e]);
  }
  const double sq = dd_block_sum<T>(acc, red);  // barriers inside also publish red[HSLOT]
  const double h = red[HSLOT];
  const double sn = sqrt(sq);                        // src:129
  const double al = sn * dhqr_alphafactor(h);        // src:130
  const double f = 1.0 / sqrt(sn * (sn + fabs(h)));  // src:131
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t row = row_of(e);
    if (row == c) a[e] = (h - al) * f;  // src:132-135
    else if (row > c) a[e] *= f;
    w0[e] = (row >= c && row < m) ? a[e] : 0.0;  // outgoing Hj (src:138-140), zero beyond the column
  }
  if (t == 0) alpha[c] = al;
  store(vnew + (int64_t)q * vlen, w0);
  store(col, a);
  __syncthreads();  // every wave's stores have reached the L2
  if (t == 0) dhqr_pipe_raise(flags, q, epoch);
}

// K steps in ONE pass over the trailing columns.  The reflectors v_jlo .. v_jlo+kold-1 already exist (`vold`, one every
// `vlen` doubles) and stay on the CU for the whole launch (three in registers, the others in LDS).  A workgroup loads a
// column once, applies them one after the
// other -- each with its own dot product over the column as updated so far, i.e. exactly the arithmetic of `kold`
// consecutive k_rank1_fused launches (src:208-209 per step) -- and stores it once: 16/K bytes of HBM traffic per
// (element, reflector) instead of 16.
//   * blockIdx 0, the LEAD workgroup (rankk_lead_body above), owns the next K columns c0 .. c0+K-1: column by column
//     it also applies the reflectors it has just built (from LDS where they fit, otherwise re-read from `vnew`: each
//     thread reads back only elements it wrote itself) and builds the column's own reflector (src:129-140), so the
//     launch hands v_c0 .. v_c0+K-1 to the next one and no single-workgroup launch sits between two passes.
//   * blockIdx b >= 1, the BULK workgroups, are persistent: b owns columns c0+K + (b-1) + i (gridDim-1), and requests
//     the next column's loads before it works on the current one (one workgroup per CU holds three column buffers and
//     three reflectors in registers, so nothing else hides the load latency while it computes).
// kold = 0 with a grid of ONE workgroup builds the first K reflectors of a matrix / panel from scratch; kold = 1
// continues from the one-reflector kernels of the tall-column phase.  Rows covered: [rtop, rtop + T*EPT), rtop = jlo
// (rounded down to even for VEC = 2); every reflector is zero above its diagonal.
template <int T, int EPT, int VEC, int K>
__global__ __launch_bounds__(T) void k_rankk_fused(double *__restrict__ A, int64_t lda, int64_t m, int64_t ncols,
                                                   int64_t c0, int64_t rtop, int kold,
                                                   const double *__restrict__ vold, double *vnew, int64_t vlen,
                                                   double *__restrict__ alpha, int *flags, int epoch) {
  constexpr int KR = rankk_kr(EPT, K);  // reflectors held in registers ...
  constexpr int KL = K - KR;         // ... and in LDS (one workgroup per CU: 64 KiB each at 8192 rows)
  __shared__ double red[2 * (T / 64) + 2];   // the lead's double-double sums + the pivot slot
  __shared__ double reda[2 * (T / 64)];      // block_sum_alt: two halves in alternation
  int par = 0;
  // ... and, where a column is short enough to leave room (<= 6144 rows), the reflectors the LEAD builds in this launch:
  // it re-reads each of them for every later column of its K, on the chain that bounds the launch once the trailing
  // matrix is small (such launches run one workgroup per CU: launch_rankk)
  constexpr int NN = rankk_lead_slots(T, EPT, K);
  __shared__ __attribute__((aligned(16))) double vl[(KL + NN) > 0 ? (KL + NN) * T * EPT : 2];
  const int t = threadIdx.x;
  const int64_t mlast = m - VEC;
  double a[EPT], an[EPT], v[KR][EPT];
  double ax[EPT];  // third column buffer of the bulk rotation

  // rows beyond m read a clamped address; a REFLECTOR is zeroed there (mask), a COLUMN is left as loaded: times the
  // reflector's zero it adds nothing to a dot product, its update is a - 0 s, and it is never stored -- so a column's
  // registers have no use between the load and the first dot product, and the bulk loop's early loads stay in flight
  auto load = [&](const double *src, double *dst, bool mask) {
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const int64_t row = rtop + 2 * ((int64_t)t + (int64_t)i * T);
        const bool ok = row < m;
        const double2 x = *reinterpret_cast<const double2 *>(src + (ok ? row : mlast));
        dst[2 * i] = (ok || !mask) ? x.x : 0.0;
        dst[2 * i + 1] = (ok || !mask) ? x.y : 0.0;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t row = rtop + t + (int64_t)e * T;
        const bool ok = row < m;
        const double x = src[ok ? row : mlast];
        dst[e] = (ok || !mask) ? x : 0.0;
      }
    }
  };
  // the bulk's column stores are non-temporal: a column is not read again before the next launch, and dirty lines left in
  // the L2s are written back at the kernel boundary, on the chain of dependent launches (8192^2: 154 -> 148 ms;
  // non-temporal LOADS of the columns: no gain)
  typedef double dhqr_d2 __attribute__((ext_vector_type(2)));
  auto store_nt = [&](double *dst, const double *src) {
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const int64_t row = rtop + 2 * ((int64_t)t + (int64_t)i * T);
        if (row < m) {
          const dhqr_d2 x = {src[2 * i], src[2 * i + 1]};
          __builtin_nontemporal_store(x, reinterpret_cast<dhqr_d2 *>(dst + row));
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int64_t row = rtop + t + (int64_t)e * T;
        if (row < m) __builtin_nontemporal_store(src[e], dst + row);
      }
    }
  };
  auto apply = [&](double *y, const double *x) {  // one step on the column in y[]: src:208 partialdot, src:209 hotloop!
    double dot = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) dot = fma(y[e], x[e], dot);
    const double s = block_sum_alt<T>(dot, reda, par);
#pragma unroll
    for (int e = 0; e < EPT; ++e) y[e] = fma(-x[e], s, y[e]);
  };
  // thread t keeps element e of an LDS-resident reflector at the position of its own 16-byte (8-byte) accesses
  auto lds_at = [&](int q, int e) -> double * {
    return (VEC == 2) ? vl + (size_t)q * T * EPT + 2 * ((size_t)t + (size_t)(e >> 1) * T) + (e & 1)
                      : vl + (size_t)q * T * EPT + (size_t)t + (size_t)e * T;
  };
  auto apply_lds = [&](double *y, int q) {
    double dot = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) dot = fma(y[e], *lds_at(q, e), dot);
    const double s = block_sum_alt<T>(dot, reda, par);
#pragma unroll
    for (int e = 0; e < EPT; ++e) y[e] = fma(-*lds_at(q, e), s, y[e]);
  };
  auto load_old = [&]() {  // the launch's reflectors: registers, then LDS (staged through ax[])
#pragma unroll
    for (int p = 0; p < KR; ++p)
      if (p < kold) load(vold + (int64_t)p * vlen, v[p], true);
#pragma unroll
    for (int q = 0; q < KL; ++q)
      if (KR + q < kold) {
        load(vold + (int64_t)(KR + q) * vlen, ax, true);
#pragma unroll
        for (int e = 0; e < EPT; ++e) *lds_at(q, e) = ax[e];  // read back by the same thread only
      }
  };
  auto apply_old = [&](double *y) {
#pragma unroll
    for (int p = 0; p < KR; ++p)
      if (p < kold) apply(y, v[p]);
#pragma unroll
    for (int q = 0; q < KL; ++q)
      if (KR + q < kold) apply_lds(y, q);
  };

  // flags != nullptr: the lead is K workgroups, one column each, handing their reflectors on (rankk_lead_pipe: built for
  // k_rankk_tall, where it took 16384 x 4096 from 312 to 257 ms); flags == nullptr: one lead workgroup (rankk_lead_body)
  const int nlead = flags ? K : 1;
  if (flags && (int)blockIdx.x < K) {
    rankk_lead_pipe<T, EPT, VEC, K>(A, lda, m, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, red, reda, flags, epoch,
                                    (int)blockIdx.x);
    return;
  }
  if ((int)blockIdx.x >= nlead) {  // ---- bulk: persistent, the next column's loads in flight behind the current column's work
    const int64_t stride = (int64_t)gridDim.x - nlead;
    int64_t c = c0 + K + ((int64_t)blockIdx.x - nlead);
    if (c >= ncols) return;
    load(A + c * lda, a, false);
    load_old();
    // THREE column buffers in rotation: a store holds its data registers until it completes, so the early load goes to
    // the buffer stored one step earlier, not to the one stored a moment ago.  (The early load is unconditional -- past
    // the last column it re-reads the current one -- so that the wait counters the compiler derives are those of
    // straight-line code: a branch around the loads would make it wait for them at once.)
#define DHQR_RK_STEP(CUR, NXT)                       \
    {                                                  \
      const int64_t cn = c + stride;                   \
      const bool more = cn < ncols;                    \
      load(A + (more ? cn : c) * lda, NXT, false);     \
      apply_old(CUR);                                  \
      store_nt(A + c * lda, CUR);                      \
      if (!more) break;                                \
      c = cn;                                          \
    }
    for (;;) {
      DHQR_RK_STEP(a, an)
      DHQR_RK_STEP(an, ax)
      DHQR_RK_STEP(ax, a)
    }
#undef DHQR_RK_STEP
    return;
  }

  // ---- lead: the next K columns and their reflectors -- a function of its own (own register allocation)
  if constexpr (T > 512) rankk_lead<T, EPT, VEC, K>(A, lda, m, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, red, reda, vl);
  else rankk_lead_body<T, EPT, VEC, K>(A, lda, m, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, red, reda, vl);
}

// k_rankk_fused for columns of 8192 < rows <= 16384: the same pass -- every trailing column loaded once, K reflectors
// applied one after the other in the reference's order (src:208-209 per step), stored once, the LEAD workgroup building
// the next K -- but a column of this height leaves neither registers nor LDS to keep the pass's reflectors on the CU:
//   * 512 threads x 24 / 32 elements (ONE workgroup per CU, 256 registers per thread; at 1024 threads the three buffers
//     of 16 elements spilled 150 registers and the pass ran SLOWER than one reflector per launch);
//   * the bulk STREAMS each reflector from `vold` (L2: K x 128 KiB per column beside 256 KiB of HBM traffic) into one
//     buffer, requested right behind its previous use;
//   * two column buffers in alternation: the next column's loads are issued after the FIRST reflector of the current
//     one -- the buffer they go to was stored at the end of the previous column, and a load into registers a store is
//     still reading from waits for that store; one reflector step later it has drained, and K - 1 steps remain to cover
//     the HBM latency;
//   * the lead is K workgroups, one column each, handing their reflectors on through flags (rankk_lead_pipe above).
// HBM traffic 16 / K bytes per element and reflector instead of the 16 of k_rank1_generic.
template <int T, int EPT, int VEC, int K>
__global__ __launch_bounds__(T) void k_rankk_tall(double *__restrict__ A, int64_t lda, int64_t m, int64_t ncols,
                                                  int64_t c0, int64_t rtop, int kold, const double *__restrict__ vold,
                                                  double *vnew, int64_t vlen, double *__restrict__ alpha, int *flags,
                                                  int epoch) {
  static_assert(T <= 512, "three buffers of a tall column need the 256 registers of a <= 512-thread workgroup");
  __shared__ double red[2 * (T / 64) + 2];
  __shared__ double reda[2 * (T / 64)];
  __shared__ __attribute__((aligned(16))) double wl[VEC == 2 ? T * EPT : 2];  // the next reflector (bulk, 16-byte path)
  if (blockIdx.x < K) {  // the K lead workgroups: one column each (rankk_lead_pipe)
    rankk_lead_pipe<T, EPT, VEC, K>(A, lda, m, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, red, reda, flags, epoch,
                                    (int)blockIdx.x);
    return;
  }
  int par = 0;
  const uint32_t t = threadIdx.x;
  // 32-bit element offsets from the (uniform) base `src + rtop`, clamped to the last valid access: the addresses are one
  // scalar base plus a recomputed lane offset, no 64-bit address registers stay live beside the three buffers
  const uint32_t span = (uint32_t)(m - rtop);          // valid rows from rtop
  const uint32_t olast = span - VEC;                   // offset of the last valid (pair of) element(s)
  double a[EPT], an[EPT], w[EPT];
  // columns: offsets clamped to the last valid access (never masked: see k_rankk_fused); reflectors: the host pads every
  // reflector slot with zeros up to rtop + T * EPT (factor_unblocked_cols: vlen), so their loads need neither
  auto load_col = [&](const double *src, double *dst, uint32_t tt) {
    const double *base = src + rtop;
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const uint32_t o = rk_umin(2u * (tt + (uint32_t)i * T), olast);
        const double2 x = *reinterpret_cast<const double2 *>(base + o);
        dst[2 * i] = x.x;
        dst[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) dst[e] = base[rk_umin(tt + (uint32_t)e * T, olast)];
    }
  };
  auto load_refl = [&](const double *src, double *dst, uint32_t tt) {
    const double *base = src + rtop;
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const double2 x = *reinterpret_cast<const double2 *>(base + 2u * (tt + (uint32_t)i * T));
        dst[2 * i] = x.x;
        dst[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) dst[e] = base[tt + (uint32_t)e * T];
    }
  };
  typedef double dhqr_d2 __attribute__((ext_vector_type(2)));
  auto store_nt = [&](double *dst, const double *src, uint32_t tt) {
    double *base = dst + rtop;
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const uint32_t o = 2u * (tt + (uint32_t)i * T);
        if (o < span) {
          const dhqr_d2 x = {src[2 * i], src[2 * i + 1]};
          __builtin_nontemporal_store(x, reinterpret_cast<dhqr_d2 *>(base + o));
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const uint32_t o = tt + (uint32_t)e * T;
        if (o < span) __builtin_nontemporal_store(src[e], base + o);
      }
    }
  };
  const int64_t stride = (int64_t)gridDim.x - K;
  int64_t c = c0 + K + ((int64_t)blockIdx.x - K);
  if (c >= ncols) return;
  load_col(A + c * lda, a, t);
  // one column: CUR holds it, NXT receives the next one (unconditional early load, see k_rankk_fused), requested behind
  // the first reflector's step
#define DHQR_RKT_APPLY(CUR)                                                              \
  {                                                                                      \
    double dot = 0.0; /* src:208 partialdot */                                           \
    _Pragma("unroll") for (int e = 0; e < EPT; ++e) dot = fma(CUR[e], w[e], dot);        \
    const double sdot = (VEC == 2) ? block_sum_alt_raw<T>(dot, reda, par) : block_sum_alt<T>(dot, reda, par); \
    _Pragma("unroll") for (int e = 0; e < EPT; ++e) CUR[e] = fma(-w[e], sdot, CUR[e]); /* src:209 hotloop! */ \
  }
#define DHQR_RKT_STEP(CUR, NXT)                                                          \
  {                                                                                      \
    const int64_t cn = c + stride;                                                       \
    const bool more = cn < ncols;                                                        \
    const uint32_t tt = rk_opaque(t); /* offsets recomputed per step, never hoisted */   \
    load_refl(vold, w, tt);                                                              \
    DHQR_RKT_APPLY(CUR)                                                                  \
    if (kold > 1) load_refl(vold + vlen, w, tt);                                         \
    load_col(A + (more ? cn : c) * lda, NXT, tt);                                        \
    for (int p = 1; p < kold; ++p) {                                                     \
      DHQR_RKT_APPLY(CUR)                                                                \
      if (p + 1 < kold) load_refl(vold + (int64_t)(p + 1) * vlen, w, rk_opaque(t));      \
    }                                                                                    \
    store_nt(A + c * lda, CUR, rk_opaque(t));                                            \
    if (!more) break;                                                                    \
    c = cn;                                                                              \
  }
  if constexpr (VEC == 2) {
    // 16-byte path: the NEXT reflector streams from L2 into LDS (direct global -> LDS loads: no registers, 16 x 16 B per
    // thread into thread-private slots wl[i][t]) while the current one, in registers, is applied; a reflector is then 16
    // ds_read_b128 away instead of an exposed L2 round trip of 128 KiB per CU (with one register buffer each of a column's
    // K reflector loads was exposed: 3.1 TB/s of the bytes as implemented at 16384 rows against 3.9-4.5 below).  Every thread
    // reads back only the slots its own lanes wrote: no barrier.
    auto issue_lds = [&](const double *src, uint32_t tt) {
      const double *base = src + rtop;
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + 2u * (tt + (uint32_t)i * T)),
                                         (__attribute__((address_space(3))) void *)(wl + 2 * (i * T + (int)(t & ~63u))), 16, 0, 0);
#else
        wl[2 * (i * T + (int)t)] = base[2u * (tt + (uint32_t)i * T)];
        wl[2 * (i * T + (int)t) + 1] = base[2u * (tt + (uint32_t)i * T) + 1];
#endif
      }
    };
    auto lds_to_regs = [&](double *dst) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const double2 x = *reinterpret_cast<const double2 *>(wl + 2 * (i * T + (int)t));
        dst[2 * i] = x.x;
        dst[2 * i + 1] = x.y;
      }
#if defined(__HIP_DEVICE_COMPILE__)
      // the reads have RETURNED before the next direct load is issued into the same slots: it is not ordered with the LDS
      // queue, and 128 KiB of ds_read_b128 take ~0.45 us to drain -- an L2 hit came back sooner and overwrote slots not yet
      // read (wrong factorisations once a workgroup owned more than one column; the dot product needs the data here anyway)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    };
#define DHQR_RKT_STEP_LDS(CUR, NXT)                                                      \
  {                                                                                      \
    const int64_t cn = c + stride;                                                       \
    const bool more = cn < ncols;                                                        \
    const uint32_t tt = rk_opaque(t);                                                    \
    lds_to_regs(w); /* reflector 0, requested during the previous column's last apply */ \
    issue_lds(vold + (kold > 1 ? vlen : 0), tt);                                         \
    DHQR_RKT_APPLY(CUR)                                                                  \
    load_col(A + (more ? cn : c) * lda, NXT, tt);                                        \
    for (int p = 1; p < kold; ++p) {                                                     \
      lds_to_regs(w);                                                                    \
      issue_lds(vold + (int64_t)(p + 1 < kold ? p + 1 : 0) * vlen, rk_opaque(t)); /* the next one, or reflector 0 for the next column */ \
      DHQR_RKT_APPLY(CUR)                                                                \
    }                                                                                    \
    store_nt(A + c * lda, CUR, rk_opaque(t));                                            \
    if (!more) break;                                                                    \
    c = cn;                                                                              \
  }
    issue_lds(vold, t);
    for (;;) {
      DHQR_RKT_STEP_LDS(a, an)
      DHQR_RKT_STEP_LDS(an, a)
    }
#undef DHQR_RKT_STEP_LDS
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0);  // the last request (never read) has landed before the workgroup's LDS is released
#endif
  } else {
    for (;;) {
      DHQR_RKT_STEP(a, an)
      DHQR_RKT_STEP(an, a)
    }
  }
#undef DHQR_RKT_APPLY
#undef DHQR_RKT_STEP
}

// The K-reflector pass for columns of 16384 < rows <= 32768 (a column = up to 256 KiB: what ONE workgroup's registers hold,
// 512 threads x 48 / 64 elements = 96 / 128 VGPRs, and nothing else).  Neither a second column buffer nor a whole
// reflector fits beside it (k_rankk_tall keeps two columns and a reflector in registers and the next reflector in LDS), so
// a reflector is STREAMED from `vold` (L2: K x 256 KiB per launch, shared by every CU) in chunks of 8 elements per thread,
// twice per step -- once for the dot product (src:208 partialdot), once for the update (src:209 hotloop!) -- two chunks in
// flight.  Per column and CU: 512 KiB of HBM traffic and 2 K x 256 KiB from L2; 16 / K bytes of HBM traffic per element
// and reflector instead of the 16 of one k_rank1_generic launch per reflector, which is what columns of this height took
// until round 4.  The arithmetic per thread (element order of the dot product, one fma per element of the update) is that
// of k_rankk_fused / k_rankk_tall with the same row map.  The lead is K pipelined workgroups, one column each (the scheme
// of rankk_lead_pipe), with the same streamed apply.
template <int T, int EPT, int VEC, int K>
__global__ __launch_bounds__(T) void k_rankk_xtall(double *__restrict__ A, int64_t lda, int64_t m, int64_t ncols,
                                                   int64_t c0, int64_t rtop, int kold, const double *__restrict__ vold,
                                                   double *vnew, int64_t vlen, double *__restrict__ alpha, int *flags,
                                                   int epoch) {
  static_assert(T <= 512 && EPT % 8 == 0, "one column in the 256 registers of a <= 512-thread workgroup, chunks of 8 elements");
  constexpr int CH = 8, NCH = EPT / CH, HSLOT = 2 * (T / 64);
  __shared__ double red[2 * (T / 64) + 2];
  __shared__ double reda[2 * (T / 64)];
  int par = 0;
  const uint32_t t = threadIdx.x;
  const uint32_t span = (uint32_t)(m - rtop);  // valid rows from rtop
  const uint32_t olast = span - VEC;           // offset of the last valid (pair of) element(s)
  double a[EPT];
  auto row_of = [&](int e) -> int64_t {
    return (VEC == 2) ? rtop + 2 * ((int64_t)t + (int64_t)(e >> 1) * T) + (e & 1) : rtop + t + (int64_t)e * T;
  };
  auto load_col = [&](const double *src) {  // clamped, never masked (see k_rankk_fused)
    const double *base = src + rtop;
    const uint32_t tt = rk_opaque(t);
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const double2 x = *reinterpret_cast<const double2 *>(base + rk_umin(2u * (tt + (uint32_t)i * T), olast));
        a[2 * i] = x.x;
        a[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) a[e] = base[rk_umin(tt + (uint32_t)e * T, olast)];
    }
  };
  typedef double dhqr_d2 __attribute__((ext_vector_type(2)));
  auto store_col = [&](double *dst, bool nt) {
    double *base = dst + rtop;
    const uint32_t tt = rk_opaque(t);
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < EPT / 2; ++i) {
        const uint32_t o = 2u * (tt + (uint32_t)i * T);
        if (o < span) {
          const dhqr_d2 x = {a[2 * i], a[2 * i + 1]};
          if (nt) __builtin_nontemporal_store(x, reinterpret_cast<dhqr_d2 *>(base + o));
          else *reinterpret_cast<dhqr_d2 *>(base + o) = x;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const uint32_t o = tt + (uint32_t)e * T;
        if (o < span) {
          if (nt) __builtin_nontemporal_store(a[e], base + o);
          else base[o] = a[e];
        }
      }
    }
  };
  // chunk j of a reflector slot (zero-padded up to rtop + T * EPT by the host: neither clamp nor mask)
  auto load_chunk = [&](const double *base, int j, double (&w)[CH], uint32_t tt) {
    if constexpr (VEC == 2) {
#pragma unroll
      for (int i = 0; i < CH / 2; ++i) {
        const double2 x = *reinterpret_cast<const double2 *>(base + 2u * (tt + (uint32_t)((CH / 2) * j + i) * T));
        w[2 * i] = x.x;
        w[2 * i + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < CH; ++e) w[e] = base[tt + (uint32_t)(CH * j + e) * T];
    }
  };
  // D chunks of a reflector in flight (the one being consumed included).  The order is pinned by scheduling barriers: left
  // alone the scheduler hoists every chunk load of the unrolled loop to the front.
  // Every CU streams the SAME reflectors at about the same time, so the pass is bound by what an XCD's L2 delivers to its 32
  // CUs (2 x 256 KiB per step and CU: 9.6 us per step at 32768 rows, ~53 GB/s per CU), not by HBM.  The otherwise unused LDS
  // takes a part of that: the first NL chunks of the dot-product pass are parked there (128 KiB, thread-private 16-byte
  // slots: no barrier) and the update pass reads them back instead of asking the L2 again.
  constexpr int D = (EPT >= 64) ? 5 : 6;
  constexpr int NL = (VEC == 2) ? (NCH < 4 ? NCH : 4) : 0;
  __shared__ __attribute__((aligned(16))) double wl[NL > 0 ? NL * CH * T : 2];
  auto pin = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto park = [&](int j, const double (&w)[CH]) {
#pragma unroll
    for (int i = 0; i < CH / 2; ++i)
      *reinterpret_cast<double2 *>(wl + 2 * (((CH / 2) * j + i) * T + (int)t)) = make_double2(w[2 * i], w[2 * i + 1]);
  };
  auto unpark = [&](int j, double (&w)[CH]) {
#pragma unroll
    for (int i = 0; i < CH / 2; ++i) {
      const double2 x = *reinterpret_cast<const double2 *>(wl + 2 * (((CH / 2) * j + i) * T + (int)t));
      w[2 * i] = x.x;
      w[2 * i + 1] = x.y;
    }
  };
  auto apply_stream = [&](const double *src) {  // one step on the column in a[]: src:208 partialdot, src:209 hotloop!
    const double *base = src + rtop;
    double w[D][CH];
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < D - 1 && j < NCH; ++j) load_chunk(base, j, w[j % D], rk_opaque(t));
    pin();
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (j + D - 1 < NCH) load_chunk(base, j + D - 1, w[(j + D - 1) % D], rk_opaque(t));
#pragma unroll
      for (int e = 0; e < CH; ++e) dot = fma(a[CH * j + e], w[j % D][e], dot);
      if (j < NL) park(j, w[j % D]);
      pin();
    }
    // the update's chunks from the L2 (NL .. NCH - 1) travel during the reduction
#pragma unroll
    for (int j = NL; j < NL + D - 1 && j < NCH; ++j) load_chunk(base, j, w[j % D], rk_opaque(t));
    pin();
    const double sdot = block_sum_alt<T>(dot, reda, par);
#pragma unroll
    for (int j = NL; j < NCH; ++j) {
      if (j + D - 1 < NCH) load_chunk(base, j + D - 1, w[(j + D - 1) % D], rk_opaque(t));
#pragma unroll
      for (int e = 0; e < CH; ++e) a[CH * j + e] = fma(-w[j % D][e], sdot, a[CH * j + e]);
      pin();
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {  // the parked chunks: every thread reads back what it wrote itself
      double u[CH];
      unpark(j, u);
#pragma unroll
      for (int e = 0; e < CH; ++e) a[CH * j + e] = fma(-u[e], sdot, a[CH * j + e]);
    }
  };

  if ((int)blockIdx.x < K) {  // ---- the K lead workgroups: column c0 + q each, reflectors handed on through flags
    const int q = (int)blockIdx.x;
    const int64_t c = c0 + q;
    if (c >= ncols) return;
    double *col = A + c * lda;
    load_col(col);
    for (int p = 0; p < kold; ++p) apply_stream(vold + (int64_t)p * vlen);
    for (int p = 0; p < q; ++p) {  // the reflectors of this launch's earlier columns, as their owners publish them
      if (t == 0) dhqr_pipe_wait(flags, p, epoch);
      __syncthreads();
      apply_stream(vnew + (int64_t)p * vlen);
    }
    dhqr_dd acc = {0.0, 0.0};  // extended-precision column norm (src:129: dnrm2)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = row_of(e);
      if (row == c) red[HSLOT] = a[e];
      if (row >= c && row < m) dd_add_sq(acc, a[e]);
    }
    const double sq = dd_block_sum<T>(acc, red);  // barriers inside also publish red[HSLOT]
    const double h = red[HSLOT];
    const double sn = sqrt(sq);                        // src:129
    const double al = sn * dhqr_alphafactor(h);        // src:130
    const double f = 1.0 / sqrt(sn * (sn + fabs(h)));  // src:131
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int64_t row = row_of(e);
      if (row == c) a[e] = (h - al) * f;  // src:132-135
      else if (row > c) a[e] *= f;
    }
    if (t == 0) alpha[c] = al;
    store_col(col, false);
    // outgoing Hj (src:138-140): the column from its diagonal down, zero above (rows rtop .. c - 1: at most K + 1 of them)
#pragma unroll
    for (int e = 0; e < EPT; ++e)
      if (row_of(e) < c) a[e] = 0.0;
    store_col(vnew + (int64_t)q * vlen, false);
    __syncthreads();  // every wave's stores have reached the L2
    if (t == 0) dhqr_pipe_raise(flags, q, epoch);
    return;
  }
  // ---- bulk: persistent, one column at a time
  const int64_t stride = (int64_t)gridDim.x - K;
  for (int64_t c = c0 + K + ((int64_t)blockIdx.x - K); c < ncols; c += stride) {
    load_col(A + c * lda);
    for (int p = 0; p < kold; ++p) apply_stream(vold + (int64_t)p * vlen);
    store_col(A + c * lda, true);
  }
}

// Fused step j for columns taller than 1024*8 rows: same contract, the column is streamed twice
// (the second pass hits L2: a 32768-row column is 256 KiB).
template <int T, int VEC>
__global__ __launch_bounds__(T) void k_rank1_generic(double *__restrict__ A, int64_t lda,
                                                     int64_t m, int64_t j,
                                                     const double *__restrict__ vcur,
                                                     double *__restrict__ vnext,
                                                     double *__restrict__ alpha) {
  __shared__ double red[2 * (T / 64) + 2];
  const int t = threadIdx.x;
  const int64_t c = j + 1 + blockIdx.x;
  double *col = A + c * lda;
  const int64_t r0 = (VEC == 2) ? (j & ~(int64_t)1) : j;

  double dot = 0.0;
  if constexpr (VEC == 2) {
    for (int64_t row = r0 + 2 * (int64_t)t; row < m; row += 2 * T) {
      const double2 x = *reinterpret_cast<const double2 *>(col + row);
      const double2 y = *reinterpret_cast<const double2 *>(vcur + row);
      dot = fma(x.x, y.x, dot);
      dot = fma(x.y, y.y, dot);
    }
  } else {
    for (int64_t row = r0 + t; row < m; row += T) dot = fma(col[row], vcur[row], dot);
  }
  const double s = block_sum<T>(dot, red);
  if constexpr (VEC == 2) {
    for (int64_t row = r0 + 2 * (int64_t)t; row < m; row += 2 * T) {
      double2 x = *reinterpret_cast<const double2 *>(col + row);
      const double2 y = *reinterpret_cast<const double2 *>(vcur + row);
      x.x = fma(-y.x, s, x.x);
      x.y = fma(-y.y, s, x.y);
      *reinterpret_cast<double2 *>(col + row) = x;
    }
  } else {
    for (int64_t row = r0 + t; row < m; row += T) col[row] = fma(-vcur[row], s, col[row]);
  }
  if (blockIdx.x != 0) return;

  const int64_t jp = j + 1;
  __syncthreads();  // column j+1 fully updated and visible inside this workgroup
  const double h = col[jp];
  dhqr_dd acc = {0.0, 0.0};  // extended-precision column norm (src:129: dnrm2)
  if constexpr (VEC == 2) {
    for (int64_t row = r0 + 2 * (int64_t)t; row < m; row += 2 * T) {
      const double2 x = *reinterpret_cast<const double2 *>(col + row);
      if (row >= jp) dd_add_sq(acc, x.x);
      if (row + 1 >= jp) dd_add_sq(acc, x.y);
    }
  } else {
    for (int64_t row = r0 + t; row < m; row += T)
      if (row >= jp) dd_add_sq(acc, col[row]);
  }
  const double s2 = dd_block_sum<T>(acc, red);
  const double sn = sqrt(s2);
  const double al = sn * dhqr_alphafactor(h);
  const double f = 1.0 / sqrt(sn * (sn + fabs(h)));
  // every thread read h before block_sum's barriers; the owner of row jp overwrites it below.
  for (int64_t row = r0 + t; row < m; row += T) {
    double val = 0.0;
    if (row >= jp) {
      val = (row == jp ? h - al : col[row]) * f;
      col[row] = val;
    }
    vnext[row] = val;
  }
  if (t == 0) alpha[jp] = al;
}
