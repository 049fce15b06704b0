// dhqr_qtb.h -- the solve written for the machine ("device-side solve PERFORMANCE").
//
// Reference mapping (src/DistributedHouseholderQR.jl):
//   _solve_householder1!  src:215-242   b <- Q'b: for every reflector j, s = partialdot(v_j, b, j:m); b[j:m] -= v_j s.
//   _solve_householder2!  src:244-282   back substitution from row n up: b[i] = (b[i] - sum_{j>i} R[i,j] b[j]) / alpha[i].
// Both are O(mn) passes over the factored matrix: HBM / latency bound, no MFMA tile for one right-hand side.
//
// Q'b, one launch per 128-column panel (k_qtb_step).  Panel k's 128 reflectors act as I - V_k T_k' V_k' (compact WY,
// T_k^{-1} = I + striu(V_k'V_k)); V_k is read IN PLACE (the strict upper part of its top block holds R and counts as zero).
// Launch k does, per slab of rows, (a) the update by panel k-1, b -= V_{k-1} w_{k-1}, and -- on the rows it has just
// brought up to date -- (b) the partial dots of panel k, y_k = V_k' b; (c) the LAST workgroup to arrive (one atomic
// counter per launch) sums the slabs' partial dots in slab order and forms w_k = T_k' y_k.  So V is streamed twice (once
// as the dot operand, once -- one launch later, out of the L2 / MALL -- as the update operand), b never leaves the
// workgroup between the two, and one dependent launch per PANEL replaces the reference's two passes per COLUMN.
// T_k' for every panel comes from one batched pre-pass that does not depend on b: k_gemm_tn_gram_batch (dhqr_gemm.h,
// FP64 MFMA, one pass over V) -> k_qtb_sum_gram -> k_build_t_batch (the blocked inverse of dhqr_recon.h).
//
// Back substitution, ONE launch (k_backsub_pipe): workgroup i owns the 128 rows of block s = nblk-1-i.  It streams its
// row block of R from the right, R[s, J] for J = nblk-1 .. s+1, subtracting R[s, J] x_J as soon as the owner of block J has
// published x_J (release / acquire flag per block), then solves its own triangular block and publishes x_s.  A workgroup
// only waits for LOWER-indexed workgroups (dispatched first; the CPU emulator runs them in index order), the next R
// block is in flight while the current one waits for its x, and what sits on the critical chain per block is one flag
// hand-over, one 128 x 128 matrix-vector product and one 128-step triangular solve inside a single wave.
#pragma once
#include "dhqr_common.h"
#include "dhqr_recon.h"

#define QTB_NB DHQR_NBV
#define QTB_NB2 (DHQR_NBV * DHQR_NBV)

// Kept T factors (dhqr_api.hip): the context holds T_k' of every panel of its last blocked factorisation of THIS matrix
// (same pointer, shape, leading dimension -- checked on the host) together with a copy of alpha.  *same = 1 iff the caller's
// alpha still is that copy bit for bit: a factor that was overwritten by another one since (the one way the kept T could be
// stale) comes with another diag(R).  Then the pre-pass kernels return at once and the Q'b kernels read the kept T'.
__global__ __launch_bounds__(1024) void k_qtb_same_alpha(const double *__restrict__ alpha, const double *__restrict__ kept,
                                                          int64_t n, int *__restrict__ same) {
  __shared__ int diff;
  if (threadIdx.x == 0) diff = 0;
  __syncthreads();
  int d = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double a = alpha[i], b = kept[i];  // bit patterns (NaN != NaN must not count as "changed")
    d |= (__double2hiint(a) != __double2hiint(b) || __double2loint(a) != __double2loint(b)) ? 1 : 0;
  }
  if (d) diff = 1;  // (benign race: every writer stores 1)
  __syncthreads();
  if (threadIdx.x == 0) *same = diff ? 0 : 1;
}

// S_k = sum over panel k's slab partials (fixed order); columns beyond the panel's width are zeroed.  grid (np, 16).
__global__ __launch_bounds__(256) void k_qtb_sum_gram(const double *__restrict__ part, const int *__restrict__ unit_first,
                                                      int64_t n, double *__restrict__ S, const int *__restrict__ skip) {
  if (*skip) return;
  const int k = blockIdx.x;
  const int u0 = unit_first[k], u1 = unit_first[k + 1];
  const int64_t c0 = (int64_t)k * QTB_NB;
  const int w = (int)((n - c0 < QTB_NB) ? n - c0 : QTB_NB);
  for (int e = blockIdx.y * 256 + threadIdx.x; e < QTB_NB2; e += gridDim.y * 256) {
    double s = 0.0;
    if ((e >> 7) < w) {
      int u = u0;
      for (; u + 3 < u1; u += 4) {
        const double a0 = part[(int64_t)u * QTB_NB2 + e], a1 = part[(int64_t)(u + 1) * QTB_NB2 + e];
        const double a2 = part[(int64_t)(u + 2) * QTB_NB2 + e], a3 = part[(int64_t)(u + 3) * QTB_NB2 + e];
        s += a0; s += a1; s += a2; s += a3;
      }
      for (; u < u1; ++u) s += part[(int64_t)u * QTB_NB2 + e];
    }
    S[(int64_t)k * QTB_NB2 + e] = s;
  }
}

// Tt_k = T_k' with T_k = (I + striu(S_k))^{-1}, for every panel: one 1024-thread workgroup per panel (k_build_t's inverse).
// Stored column-major, Tt[i + 128 j] = T'[i][j] = T[j][i]: lower triangular.
__global__ __launch_bounds__(1024) void k_build_t_batch(const double *__restrict__ S_all, int64_t n,
                                                         double *__restrict__ Tt_all, const int *__restrict__ skip) {
  __shared__ rc5_lds L;
  if (*skip) return;
  const int64_t k = blockIdx.x;
  const int64_t c0 = k * QTB_NB;
  const int w = (int)((n - c0 < QTB_NB) ? n - c0 : QTB_NB);
  double x12[4];
  rc_upper_inverse_blocked(S_all + k * QTB_NB2, w, true, L, x12);
  double *Tt = Tt_all + k * QTB_NB2;
  rc5_emit(L, x12, [&](int i, int j, double v) { Tt[j + i * QTB_NB] = v; });
}

// Sums of 32 per-lane values over the 64 lanes of a wave by halving: after the exchange with lane ^ 32 a lane keeps 16
// of the 32 sums, after lane ^ 16 eight, ... -- 32 exchanges instead of 32 x 6.  Returns, in every lane, the total of value
// index qtb_red_index(lane) (both lanes of a pair (l, l ^ 1) hold the same one).
__device__ __forceinline__ int qtb_red_index(int lane) {
  return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}
__device__ __forceinline__ double qtb_wave_reduce32(double (&y)[32], int lane) {
#define QTB_HALVE(N_, BIT_)                                                  \
  {                                                                          \
    const bool up = (lane & (BIT_)) != 0;                                    \
    _Pragma("unroll") for (int q = 0; q < (N_); ++q) {                       \
      const double keep = up ? y[q + (N_)] : y[q];                           \
      const double send = up ? y[q] : y[q + (N_)];                           \
      y[q] = keep + __shfl_xor(send, (BIT_), 64);                            \
    }                                                                        \
  }
  QTB_HALVE(16, 32)
  QTB_HALVE(8, 16)
  QTB_HALVE(4, 8)
  QTB_HALVE(2, 4)
  QTB_HALVE(1, 2)
#undef QTB_HALVE
  return y[0] + __shfl_xor(y[0], 1, 64);
}

// ---- b <- Q'b (file header) --------------------------------------------------------------------------------------------
// Shared memory of the Q'b kernels.  Tp: T_k' packed by columns (column j holds rows j .. 127 at Tp[j * 128 - j (j - 1) / 2 ..]),
// staged by the reducing workgroup BEFORE its own slab work so that the w_k = T_k' y_k product at the end of a panel step --
// on the critical chain of every step -- reads LDS instead of waiting for 128 KB from the other end of the chip.
#define QTB_TP_ELEMS (QTB_NB * (QTB_NB + 1) / 2)
template <int VEC>
struct qtb_lds {
  double w_s[QTB_NB];
  double red[2][4][64 * VEC];
  double y_s[2][QTB_NB];
  double Tp[QTB_TP_ELEMS];
};
__device__ __forceinline__ int qtb_tp_off(int j) { return j * QTB_NB - (j * (j - 1)) / 2; }

// Bounded wait (one thread): until *word >= want.  Relaxed polls + one acquire fence; an expired wait sets the context's
// pipeline error word (err[0]; err[1] = the poll bound, DHQR_PIPE_LIMIT_OFFSET).
__device__ __forceinline__ void qtb_wait_ge(int *word, int want, int *err) {
  int spins = 0;
  const int limit = err[1];
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(1);
    // (once any wait of the call has expired its results are void: the others give up at once instead of one bound each)
    if (++spins > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      __hip_atomic_store(err, 0x7ffffffe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// T_k' (dense, column-major, lower triangular) -> L.Tp (all 256 threads; the caller's next barrier publishes it)
template <int VEC>
__device__ __forceinline__ void qtb_stage_t(const double *__restrict__ Tt, qtb_lds<VEC> &L) {
  const int t = threadIdx.x, i = t & 127, h = t >> 7;
#pragma unroll 1
  for (int ub = 0; ub < 64; ub += 32) {  // 32 loads in flight per thread, then their LDS stores: two memory latencies
    double tv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) tv[u] = Tt[i + (2 * (ub + u) + h) * QTB_NB];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int j = 2 * (ub + u) + h;
      if (i >= j) L.Tp[qtb_tp_off(j) + i - j] = tv[u];
    }
  }
}

// The slab phase of panel step k for rows [r_lo, r_hi) (r_lo a multiple of 64 VEC, >= the step's first active row):
// (a) k >= 1: b -= V_{k-1} w_{k-1} (w in L.w_s), (b) k < np: this slab's partial dots of panel k -> yrow[0..128).
// 256 threads: wave g takes the panel's columns 32 g .. 32 g + 31, lane l the rows r0 + VEC l .. of every 64 VEC-row
// sub-slab.  VEC = 2: 16-byte loads (lda, m even, 16-byte aligned A, b).
// goff (r6, the row split at P > 1): the GLOBAL row of local row 0 -- A, b and every row index here are local to a rank that
// holds rows [goff, goff + m); a panel's first row (and the triangle of R above its diagonal) are global quantities.
template <int VEC>
__device__ __forceinline__ void qtb_slab_phase(const double *__restrict__ A, int64_t lda, int64_t m, int64_t n, int k, int np,
                                               int64_t r_lo, int64_t r_hi, double *__restrict__ b, qtb_lds<VEC> &L,
                                               double *__restrict__ yrow, int &par, int64_t goff = 0) {
  constexpr int SS = 64 * VEC;
  double (&w_s)[QTB_NB] = L.w_s;
  double (&red)[2][4][SS] = L.red;
  const int t = threadIdx.x, lane = t & 63, g = t >> 6;
  const bool upd = k >= 1, dot = k < np;
  const int64_t cu = (int64_t)(k - 1) * QTB_NB;  // first column (= first row) of the panel that updates
  const int64_t cd = (int64_t)k * QTB_NB;        // ... of the panel whose dot products are formed
  const int64_t ru = cu - goff, rd = cd - goff;  // the same as LOCAL row numbers (negative: the panel starts above this rank's rows)
  const int wu = upd ? (int)((n - cu < QTB_NB) ? n - cu : QTB_NB) : 0;
  const int wd = dot ? (int)((n - cd < QTB_NB) ? n - cd : QTB_NB) : 0;
  double yacc[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) yacc[q] = 0.0;
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += SS) {
    const int64_t r = r0 + (int64_t)lane * VEC;
    const int64_t ra = (r + VEC <= m) ? r : (m - VEC);  // address row: never beyond the matrix (values masked below)
    double bv[VEC];
    if constexpr (VEC == 2) {
      bv[0] = bv[1] = 0.0;
      if (r < m) {  // m even: the pair is inside or outside as a whole
        const double2 x = *reinterpret_cast<const double2 *>(b + r);
        bv[0] = x.x; bv[1] = x.y;
      }
    } else {
      bv[0] = (r < m) ? b[r] : 0.0;
    }
    const bool tail = r0 + SS > m;
    if (upd) {
      double acc[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = 0.0;
      const int jn = (wu - g * 32 < 32) ? wu - g * 32 : 32;  // this wave's columns inside the panel (wave-uniform)
      const double *Ac = A + ra + (cu + (jn > 0 ? g * 32 : 0)) * lda;  // (a wave beyond a partial panel reads column cu, masked)
      if (!tail && r0 >= ru + QTB_NB && jn == 32) {          // below the top block: no masks
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double v[16][VEC];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if constexpr (VEC == 2) {
              const double2 x = *reinterpret_cast<const double2 *>(Ac + (int64_t)(h * 16 + q) * lda);
              v[q][0] = x.x; v[q][1] = x.y;
            } else {
              v[q][0] = Ac[(int64_t)(h * 16 + q) * lda];
            }
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const double wj = w_s[g * 32 + h * 16 + q];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fma(v[q][e], wj, acc[e]);
          }
        }
      } else {  // top block (R above the diagonal counts as zero), last rows, partial panel: every load is issued
                // unconditionally at a clamped address, the selects follow (a loop of dependent load -> fma iterations
                // cost one memory latency per column: 25 us per launch on the slab that holds the top block)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double v[16][VEC];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int qq = (h * 16 + q < jn) ? h * 16 + q : 0;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[q][e] = Ac[(int64_t)qq * lda + ((r + e < m) ? (r + e - ra) : 0)];
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int j = g * 32 + h * 16 + q;
            const double wj = w_s[j];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const bool ok = (h * 16 + q < jn) && (r + e < m) && (r + e >= ru + j);
              acc[e] = fma(ok ? v[q][e] : 0.0, wj, acc[e]);
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[par][g][lane * VEC + e] = acc[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int i = lane * VEC + e;
        bv[e] -= (red[par][0][i] + red[par][1][i]) + (red[par][2][i] + red[par][3][i]);  // src:219-221 for 128 reflectors
      }
      par ^= 1;
      if (g == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (r + e < m) b[r + e] = bv[e];
      }
    }
    if (dot && r0 + SS > rd) {
      const int jn = (wd - g * 32 < 32) ? wd - g * 32 : 32;
      const double *Ac = A + ra + (cd + (jn > 0 ? g * 32 : 0)) * lda;
      if (!tail && r0 >= rd + QTB_NB && jn == 32) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double v[16][VEC];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if constexpr (VEC == 2) {
              const double2 x = *reinterpret_cast<const double2 *>(Ac + (int64_t)(h * 16 + q) * lda);
              v[q][0] = x.x; v[q][1] = x.y;
            } else {
              v[q][0] = Ac[(int64_t)(h * 16 + q) * lda];
            }
          }
#pragma unroll
          for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) yacc[h * 16 + q] = fma(v[q][e], bv[e], yacc[h * 16 + q]);  // src:218
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double v[16][VEC];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int qq = (h * 16 + q < jn) ? h * 16 + q : 0;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[q][e] = Ac[(int64_t)qq * lda + ((r + e < m) ? (r + e - ra) : 0)];
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int j = g * 32 + h * 16 + q;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const bool ok = (h * 16 + q < jn) && (r + e < m) && (r + e >= rd + j);
              yacc[h * 16 + q] = fma(ok ? v[q][e] : 0.0, bv[e], yacc[h * 16 + q]);
            }
          }
        }
      }
    }
  }
  if (!dot) return;
  const double tot = qtb_wave_reduce32(yacc, lane);
  if ((lane & 1) == 0) yrow[g * 32 + qtb_red_index(lane)] = tot;
}

// The reducing workgroup's part of panel step k: y_k = the sum of the ns slabs' partial dots in slab order (two interleaved
// halves, then the halves), w_k = T_k' y_k out of L.Tp -> wk[0..128).  All 256 threads; ypart rows are 128 doubles.
template <int VEC>
__device__ __forceinline__ void qtb_reduce_phase(const double *__restrict__ ypart, int ns, qtb_lds<VEC> &L,
                                                 double *__restrict__ wk, bool sum_only = false) {
  const int t = threadIdx.x, i = t & 127, h = t >> 7;
  {
    double s = 0.0;
    int q = h;
    for (; q + 6 < ns; q += 8) {
      const double a0 = ypart[(int64_t)q * QTB_NB + i], a1 = ypart[(int64_t)(q + 2) * QTB_NB + i];
      const double a2 = ypart[(int64_t)(q + 4) * QTB_NB + i], a3 = ypart[(int64_t)(q + 6) * QTB_NB + i];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; q < ns; q += 2) s += ypart[(int64_t)q * QTB_NB + i];
    L.y_s[h][i] = s;
  }
  __syncthreads();
  if (t < QTB_NB) L.y_s[0][t] += L.y_s[1][t];
  if (sum_only) {  // (uniform) the dots leave for an all-reduce: no T product here
    if (t < QTB_NB) wk[t] = L.y_s[0][t];
    return;
  }
  __syncthreads();
  // thread (i, h): columns j = h, h + 2, ... <= i of row i (src:218-221 for the panel's 128 reflectors at once)
  double a0 = 0.0;
  for (int j = h; j <= i; j += 2) a0 = fma(L.Tp[qtb_tp_off(j) + i - j], L.y_s[0][j], a0);
  if (h == 1) L.y_s[1][i] = a0;
  __syncthreads();
  if (h == 0) wk[i] = a0 + L.y_s[1][i];
}

// Row slabs of the Q'b kernels: `sl` rows (a multiple of 64 VEC); panel step k touches the slabs from floor(rfirst / sl) on,
// rfirst = the first row of panel k-1 (k >= 1: its update) or 0.
__device__ __forceinline__ int64_t qtb_rfirst(int k) { return (int64_t)(k >= 1 ? k - 1 : 0) * QTB_NB; }

// One launch per panel step k = 0 .. np (the form the CPU emulator runs, and the fallback when several contexts share a
// device: no workgroup of it waits for a HIGHER-indexed one).  Workgroup x owns slab floor(rfirst / sl) + x; the LAST
// workgroup reduces: it stages T_k' first, does its slab, waits until the others have arrived (counter[k], one release
// increment each), sums and writes w_k.  ypart: gridDim.x x 128; counter zeroed before the first launch.
template <int VEC>
__global__ __launch_bounds__(256) void k_qtb_step(const double *__restrict__ A, int64_t lda, int64_t m, int64_t n, int k,
                                                  int np, int64_t sl, double *__restrict__ b,
                                                  const double *__restrict__ Tt_new, const double *__restrict__ Tt_kept,
                                                  const int *__restrict__ use_kept, double *__restrict__ wbuf,
                                                  double *__restrict__ ypart, int *__restrict__ counter,
                                                  int *__restrict__ err, int64_t goff = 0, double *__restrict__ ydist = nullptr) {
  // goff / ydist (r6, the row split at P > 1, rs_solve): this rank holds rows [goff, goff + m) of the matrix; the reducer
  // leaves the sum of the LOCAL slabs' partial dots in ydist[k] -- the ranks all-reduce it and k_qtb_tw forms w_k -- instead
  // of applying T_k' itself.  wbuf / ypart / counter as below.
  __shared__ qtb_lds<VEC> L;
  const int t = threadIdx.x;
  const bool dist = ydist != nullptr;
  const double *Tt_all = *use_kept ? Tt_kept : Tt_new;
  const bool dot = k < np, reducer = dot && blockIdx.x == gridDim.x - 1;
  const int64_t rglob = qtb_rfirst(k);
  const int64_t rfirst = rglob > goff ? rglob - goff : 0;  // first active LOCAL row
  const int64_t slab = rfirst / sl + blockIdx.x;
  const int64_t r_lo = (slab * sl > rfirst) ? slab * sl : rfirst;
  const int64_t r_hi = ((slab + 1) * sl < m) ? (slab + 1) * sl : m;
  if (reducer && !dist) qtb_stage_t<VEC>(Tt_all + (int64_t)k * QTB_NB2, L);
  if (k >= 1 && t < QTB_NB) L.w_s[t] = wbuf[(int64_t)(k - 1) * QTB_NB + t];
  __syncthreads();
  int par = 0;
  qtb_slab_phase<VEC>(A, lda, m, n, k, np, r_lo, r_hi, b, L, ypart + (int64_t)blockIdx.x * QTB_NB, par, goff);
  if (!dot) return;
  __syncthreads();  // every wave's partial dots are stored (the barrier waits for the stores)
  if (!reducer) {
    if (t == 0) __hip_atomic_fetch_add(counter + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (t == 0) qtb_wait_ge(counter + k, (int)gridDim.x - 1, err);
  __syncthreads();
  qtb_reduce_phase<VEC>(ypart, (int)gridDim.x, L, dist ? ydist + (int64_t)k * QTB_NB : wbuf + (int64_t)k * QTB_NB, dist);
}

// w = T' y for one panel (T' dense, column-major, lower triangular): what the reducer of k_qtb_step does when the dots
// need no all-reduce in between.  One workgroup of 256 threads: thread (i, h) takes the columns j = h, h + 2, ... <= i of row i.
__global__ __launch_bounds__(256) void k_qtb_tw(const double *__restrict__ Tt, const double *__restrict__ y,
                                                double *__restrict__ w) {
  __shared__ double ys[QTB_NB], half[QTB_NB];
  const int t = threadIdx.x, i = t & 127, h = t >> 7;
  if (t < QTB_NB) ys[t] = y[t];
  __syncthreads();
  double a0 = 0.0, a1 = 0.0;
  int j = h;
  for (; j + 2 <= i; j += 4) {
    a0 = fma(Tt[i + j * QTB_NB], ys[j], a0);
    a1 = fma(Tt[i + (j + 2) * QTB_NB], ys[j + 2], a1);
  }
  for (; j <= i; j += 2) a0 = fma(Tt[i + j * QTB_NB], ys[j], a0);
  a0 += a1;
  if (h == 1) half[i] = a0;
  __syncthreads();
  if (h == 0) w[i] = a0 + half[i];
}

// ---- the same panel steps in ONE launch ----------------------------------------------------------------------------------
// Workgroup s owns the 64 VEC rows of slab s for the whole of Q'b and retires when the panels have moved below it; the last
// workgroup (the bottom slab, active to the end) also reduces every step and publishes w_k by raising wflag[k].  What this
// form buys over one launch per step (measured with time stamps in the kernel, 8192^2: 20 us per step before, see DESIGN):
//   * a workgroup's operands never leave its registers: its rows of b stay in bv, and the slab of V_k it loads for the dot
//     products of step k IS the update operand of step k + 1 -- every element of V crosses the memory system ONCE;
//   * the loads of step k's slab are issued BEFORE the wait for w_{k-1}, so the chain per step is flag -> w (128 doubles)
//     -> 2 x CPW fma per lane -> LDS sum -> partial dots -> arrive; the reducer's T_k' loads are in flight while the last
//     workgroups arrive, its gather of the partial dots is one batch of loads.
// The workgroups wait for each other in BOTH directions (everybody for the reducer's w, the reducer for everybody's
// arrival), so all of them must be resident: the host launches at most one workgroup per CU, and only while this context
// has the device to itself (dhqr_api.hip).  NW waves per workgroup: wave g takes the panel's columns CPW g .. CPW g + CPW - 1.
template <int NW>
struct qtbp_lds {
  double w_s[QTB_NB];
  double red[2][NW][128];   // [parity][wave][row of the slab] (64 VEC <= 128 rows)
  double y4[NW][QTB_NB];
  double y_s[QTB_NB];
};
template <int N>
__device__ __forceinline__ double qtb_wave_reduce_n(double (&y)[N], int lane);
template <>
__device__ __forceinline__ double qtb_wave_reduce_n<32>(double (&y)[32], int lane) { return qtb_wave_reduce32(y, lane); }
// 16 values: four halvings, then the two remaining lane bits (index = bits 5..2 of the lane)
template <>
__device__ __forceinline__ double qtb_wave_reduce_n<16>(double (&y)[16], int lane) {
#define QTB_HALVE(N_, BIT_)                                                  \
  {                                                                          \
    const bool up = (lane & (BIT_)) != 0;                                    \
    _Pragma("unroll") for (int q = 0; q < (N_); ++q) {                       \
      const double keep = up ? y[q + (N_)] : y[q];                           \
      const double send = up ? y[q] : y[q + (N_)];                           \
      y[q] = keep + __shfl_xor(send, (BIT_), 64);                            \
    }                                                                        \
  }
  QTB_HALVE(8, 32)
  QTB_HALVE(4, 16)
  QTB_HALVE(2, 8)
  QTB_HALVE(1, 4)
#undef QTB_HALVE
  double v = y[0] + __shfl_xor(y[0], 2, 64);
  return v + __shfl_xor(v, 1, 64);
}
template <int N>
__device__ __forceinline__ int qtb_red_index_n(int lane) {
  if constexpr (N == 32) return qtb_red_index(lane);
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

template <int VEC, int NW>
__global__ __launch_bounds__(64 * NW) void k_qtb_persist(const double *__restrict__ A, int64_t lda, int64_t m, int64_t n,
                                                         int np, double *__restrict__ b, const double *__restrict__ Tt_new,
                                                         const double *__restrict__ Tt_kept, const int *__restrict__ use_kept,
                                                         double *__restrict__ wbuf, double *__restrict__ ypart,
                                                         int *__restrict__ counter, int *__restrict__ wflag,
                                                         int *__restrict__ err) {
  constexpr int SS = 64 * VEC, CPW = QTB_NB / NW, NT = 64 * NW;
  __shared__ qtbp_lds<NW> L;
  const double *Tt_all = *use_kept ? Tt_kept : Tt_new;
  const int t = threadIdx.x, lane = t & 63, g = t >> 6;
  const int64_t slab = blockIdx.x, nsl = gridDim.x;
  const bool reducer = slab == nsl - 1;
  const int64_t s_lo = slab * SS, s_hi = (s_lo + SS < m) ? s_lo + SS : m;
  const int64_t r = s_lo + (int64_t)lane * VEC;
  const int64_t ra = (r + VEC <= m) ? r : (m - VEC);  // address row, never beyond the matrix (values masked at use)
  double bv[VEC], va[CPW][VEC], vb[CPW][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = (r + e < m) ? b[r + e] : 0.0;
#pragma unroll
  for (int q = 0; q < CPW; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) va[q][e] = vb[q][e] = 0.0;
  int par = 0;
  for (int k = 0; k <= np; ++k) {
    const int64_t rfirst = qtb_rfirst(k);
    if (s_hi <= rfirst) return;  // the panels have moved below this slab
    const int64_t first_slab = rfirst / SS;
    const bool upd = k >= 1, dot = k < np;
    const int64_t cu = (int64_t)(k - 1) * QTB_NB, cd = (int64_t)k * QTB_NB;
    const int wu = upd ? (int)((n - cu < QTB_NB) ? n - cu : QTB_NB) : 0;
    const int wd = dot ? (int)((n - cd < QTB_NB) ? n - cd : QTB_NB) : 0;
    const bool dots_here = dot && s_hi > cd;  // (a slab that ends inside panel k-1's top block only takes the update)
    // ---- this step's slab of V_k, requested before anything is waited for
    if (dots_here) {
      const int jn = (wd - g * CPW < CPW) ? wd - g * CPW : CPW;  // this wave's columns inside the panel (wave-uniform)
      const double *Ac = A + ra + (cd + (jn > 0 ? g * CPW : 0)) * lda;
#pragma unroll
      for (int q = 0; q < CPW; ++q) {
        const double *p = Ac + (int64_t)((q < jn) ? q : 0) * lda;
        if constexpr (VEC == 2) {
          const double2 x = *reinterpret_cast<const double2 *>(p);
          vb[q][0] = x.x; vb[q][1] = x.y;
        } else {
          vb[q][0] = p[0];
        }
      }
    }
    // ---- w_{k-1}
    if (upd) {
      if (!reducer) {  // (the reducer wrote w_{k-1} itself)
        if (t == 0) qtb_wait_ge(wflag + (k - 1), 1, err);
        __syncthreads();
      }
      if (t < QTB_NB) L.w_s[t] = wbuf[(int64_t)(k - 1) * QTB_NB + t];
      __syncthreads();
      // ---- b -= V_{k-1} w_{k-1} on this slab: va holds the slab of V_{k-1} loaded one step ago (src:219-221)
      double acc[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = 0.0;
#pragma unroll
      for (int q = 0; q < CPW; ++q) {
        const int j = g * CPW + q;
        const double wj = L.w_s[j];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const bool ok = (j < wu) && (r + e < m) && (r + e >= cu + j);  // rows of the top block above the diagonal hold R
          acc[e] = fma(ok ? va[q][e] : 0.0, wj, acc[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) L.red[par][g][lane * VEC + e] = acc[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += L.red[par][w][lane * VEC + e];
        bv[e] -= sum;
      }
      par ^= 1;
      if (g == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (r + e < m) b[r + e] = bv[e];
      }
    }
    if (!dot) return;
    // ---- partial dots of panel k on the rows just brought up to date (src:218)
    double yq[CPW];
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
      const int j = g * CPW + q;
      double y = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const bool ok = dots_here && (j < wd) && (r + e < m) && (r + e >= cd + j);
        y = fma(ok ? vb[q][e] : 0.0, bv[e], y);
      }
      yq[q] = y;
#pragma unroll
      for (int e = 0; e < VEC; ++e) va[q][e] = vb[q][e];  // the update operand of the next step
    }
    {
      const double tot = qtb_wave_reduce_n<CPW>(yq, lane);
      const bool writer = (CPW == 32) ? ((lane & 1) == 0) : ((lane & 3) == 0);
      if (writer) ypart[(slab - first_slab) * QTB_NB + g * CPW + qtb_red_index_n<CPW>(lane)] = tot;
    }
    __syncthreads();  // the partial dots are stored
    if (!reducer) {
      if (t == 0) __hip_atomic_fetch_add(counter + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    // ---- reducer: T_k' row pieces first (in flight while the last workgroups arrive), then the gather, then w_k = T_k' y_k
    constexpr int NH = NT / QTB_NB, JW = QTB_NB / NH;  // thread (i, hh): row i, columns JW hh .. JW hh + JW - 1
    const int i = t & 127, hh = t >> 7;
    double tv[JW];
    {
      const double *Tt = Tt_all + (int64_t)k * QTB_NB2 + i + (int64_t)(JW * hh) * QTB_NB;
#pragma unroll
      for (int u = 0; u < JW; ++u) tv[u] = Tt[(int64_t)u * QTB_NB];
    }
    const int ns = (int)(nsl - first_slab);
    if (t == 0) qtb_wait_ge(counter + k, ns - 1, err);
    __syncthreads();
    {  // y_k: thread (columns 2 c2, 2 c2 + 1; row group rg): rows rg, rg + NW, ... in order, then the groups in order
      const int c2 = t & 63, rg = t >> 6;
      double s0 = 0.0, s1 = 0.0;
      constexpr int PB = (NW == 8) ? 16 : 32;  // loads in flight per thread (512-thread workgroups have 256 registers a thread)
      for (int q0 = rg; q0 < ns; q0 += PB * NW) {
        double2 pv[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int q = q0 + u * NW;
          pv[u] = *reinterpret_cast<const double2 *>(ypart + (int64_t)((q < ns) ? q : rg) * QTB_NB + 2 * c2);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u)
          if (q0 + u * NW < ns) { s0 += pv[u].x; s1 += pv[u].y; }
      }
      L.y4[rg][2 * c2] = s0;
      L.y4[rg][2 * c2 + 1] = s1;
    }
    __syncthreads();
    if (t < QTB_NB) {
      double y = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) y += L.y4[w][t];
      L.y_s[t] = y;
    }
    __syncthreads();
    {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int u = 0; u < JW; u += 2) {
        a0 = fma(tv[u], L.y_s[JW * hh + u], a0);
        a1 = fma(tv[u + 1], L.y_s[JW * hh + u + 1], a1);
      }
      L.y4[hh][i] = a0 + a1;  // (y4 is free again: every thread is past the sum above)
    }
    __syncthreads();
    if (t < QTB_NB) {
      double w = 0.0;
#pragma unroll
      for (int h2 = 0; h2 < NH; ++h2) w += L.y4[h2][t];
      wbuf[(int64_t)k * QTB_NB + t] = w;
    }
    __syncthreads();  // w_k is stored by this workgroup's threads ...
    if (t == 0) __hip_atomic_store(wflag + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // ... then the flag
  }
}

// value of lane `src` (wave-uniform) in every lane: two v_readlane_b32, no LDS crossbar round trip
__device__ __forceinline__ double qtb_lane_bcast(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// ---- pipelined back substitution (file header) -----------------------------------------------------------------------
// Bounded wait for block `idx`'s flag (one thread of the workgroup): relaxed polls + one acquire fence, like dhqr_pipe_wait;
// a waiter that gives up records it in the context's pipeline error word (reported by the next synchronising entry point).
__device__ __forceinline__ void qtb_flag_wait(int *flags, int idx, int *err) {
  int spins = 0;
  const int limit = err[1];  // DHQR_PIPE_LIMIT_OFFSET: the word behind the error word
  while (__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > limit) {
      __hip_atomic_store(err, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

#define BSP_THREADS 512
// x = R^{-1} b[0:n] in place (b[0:n] <- x).  grid = nblk = ceil(n / 128) workgroups of 512 threads; flags: nblk ints, zero.
// Thread (r = t & 127, q = t >> 7): row r of the block, columns 32 q .. 32 q + 31 of every 128-column block to the right.
// Rs: the workgroup's own diagonal block, column c scaled by 1 / alpha_c and zero on and below the diagonal, so that the
// triangular solve's chain per column is one lane broadcast and one fma: b_t -= (R[t,c] / alpha_c) b_c, x_c = b_c / alpha_c.
__global__ __launch_bounds__(BSP_THREADS) void k_backsub_pipe(const double *__restrict__ A, int64_t lda,
                                                              const double *__restrict__ alpha, double *__restrict__ b,
                                                              int64_t n, int *__restrict__ flags, int *__restrict__ err) {
  __shared__ double Rs[QTB_NB * QTB_NB];  // 128 KiB: column c at Rs[128 c ..]
  __shared__ double part[4][QTB_NB];
  __shared__ double ainv[QTB_NB];
  const int t = threadIdx.x, r = t & 127, q = t >> 7;
  const int64_t nblk = (n + QTB_NB - 1) / QTB_NB;
  const int64_t s = nblk - 1 - (int64_t)blockIdx.x;
  const int64_t row0 = s * QTB_NB;
  const int hb = (int)((n - row0 < QTB_NB) ? n - row0 : QTB_NB);  // rows (= columns) of the own block
  const bool rok = r < hb;
  const int64_t ra = row0 + (rok ? r : 0);

  // the first block to the right is requested before anything else
  double cur[32], nxt[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) cur[c] = nxt[c] = 0.0;
  auto load_block = [&](int64_t J, double (&dst)[32]) {
    const int64_t c0 = J * QTB_NB + q * 32;
    const int wj = (int)((n - c0 < 32) ? ((n - c0 > 0) ? n - c0 : 0) : 32);
    const double *p = A + ra + ((wj > 0) ? c0 : 0) * lda;  // a column group beyond the matrix reads column 0 (times x = 0)
#pragma unroll
    for (int c = 0; c < 32; ++c) dst[c] = p[(int64_t)((c < wj) ? c : 0) * lda];
  };
  if (s + 1 < nblk) load_block(nblk - 1, cur);
  // own diagonal block -> LDS, scaled; 1 / alpha by reciprocal + two Newton steps (dhqr_rcp)
  if (t < QTB_NB) ainv[t] = (t < hb) ? dhqr_rcp(alpha[row0 + t]) : 1.0;
  __syncthreads();
  for (int e = t; e < QTB_NB * QTB_NB; e += BSP_THREADS) {
    const int i = e & 127, c = e >> 7;
    double v = 0.0;
    if (i < c && c < hb) v = A[(row0 + i) + (row0 + c) * lda] * ainv[c];
    Rs[e] = v;
  }
  double acc = 0.0;  // this thread's part of sum_J R[s, J] x_J (its 32 columns of every block)
  for (int64_t J = nblk - 1; J > s; --J) {
    if (J - 1 > s) load_block(J - 1, nxt);
    if (t == 0) qtb_flag_wait(flags, (int)J, err);
    __syncthreads();
    const int64_t c0 = J * QTB_NB + q * 32;
    const int wj = (int)((n - c0 < 32) ? ((n - c0 > 0) ? n - c0 : 0) : 32);
    const double xv = ((t & 31) < wj) ? b[c0 + (t & 31)] : 0.0;  // lanes 0..31 / 32..63 of a wave hold the same 32 x
#pragma unroll
    for (int c = 0; c < 32; ++c) acc = fma(cur[c], qtb_lane_bcast(xv, c), acc);
#pragma unroll
    for (int c = 0; c < 32; ++c) cur[c] = nxt[c];
  }
  part[q][r] = acc;
  __syncthreads();  // also: Rs and ainv are complete
  if (t < 64) {     // one wave: rows t and t + 64
    double b0 = (t < hb) ? b[row0 + t] : 0.0, b1 = (t + 64 < hb) ? b[row0 + t + 64] : 0.0;
    b0 -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    b1 -= (part[0][t + 64] + part[1][t + 64]) + (part[2][t + 64] + part[3][t + 64]);
    // src:248-251, column-oriented, 16 columns at a time: their entries come out of LDS first (independent of the chain),
    // the chain itself is lane broadcast -> fma per column.  Columns >= hb of a partial block are zero in Rs.
#pragma unroll 1
    for (int cb = 112; cb >= 64; cb -= 16) {
      double ra[16], rb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        ra[u] = Rs[(cb + u) * QTB_NB + t];
        rb[u] = Rs[(cb + u) * QTB_NB + 64 + t];
      }
#pragma unroll
      for (int u = 15; u >= 0; --u) {
        const double bc = qtb_lane_bcast(b1, cb + u - 64);
        b0 = fma(-ra[u], bc, b0);
        b1 = fma(-rb[u], bc, b1);
      }
    }
#pragma unroll 1
    for (int cb = 48; cb >= 0; cb -= 16) {
      double ra[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) ra[u] = Rs[(cb + u) * QTB_NB + t];
#pragma unroll
      for (int u = 15; u >= 0; --u) b0 = fma(-ra[u], qtb_lane_bcast(b0, cb + u), b0);
    }
    // x_c = b_c / alpha_c: reciprocal product + one correction step (the quotient to within an ulp)
    if (t < hb) {
      const double al = alpha[row0 + t];
      double x = b0 * ainv[t];
      x = fma(fma(-al, x, b0), ainv[t], x);
      b[row0 + t] = x;
    }
    if (t + 64 < hb) {
      const double al = alpha[row0 + t + 64];
      double x = b1 * ainv[t + 64];
      x = fma(fma(-al, x, b1), ainv[t + 64], x);
      b[row0 + t + 64] = x;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the wave's stores of x, then the flag
    if (t == 0) __hip_atomic_store(flags + s, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
