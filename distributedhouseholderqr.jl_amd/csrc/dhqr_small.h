// dhqr_small.h -- qr!(A) and H \ b of a SMALL matrix, each in ONE single-workgroup launch.
//
// The shapes the reference's own test file times (test/runtests.jl:42: n = 100, 200, ... with m = 1.1 n, ratio against
// LAPACK printed at :87-89) start far below anything a blocked driver is made for: at 110 x 100 the look-ahead driver's
// group buffers, status read-backs and ~250 launches cost 1.2 ms where LAPACK needs 0.35 ms.  A matrix of up to
// 224 x 224 doubles (392 KB) fits into the REGISTERS of one compute unit (512 KB), so the reference's algorithm
// (src:122-148, 198-213) runs here exactly as written -- one reflector after the other, every trailing column updated by
// every reflector -- with the whole matrix resident in VGPRs and ONE workgroup barrier per column:
//
//   k_small_qr_d<NR, NQ> 512 threads = 8 waves (two per SIMD, 256 registers per lane; + a ninth for m <= 128).  Lane (rg, cs) of
//                        wave w holds rows rg + 16 r (r < NR) of the columns 32 q + 4 w + cs (q < NQ): a 16-lane DPP row spans
//                        16 consecutive matrix rows of one column, the four rows of a wave are four adjacent columns.  The dot
//                        product v_j' a_c (partialdot, src:42-49) is NR fma per lane + a 4-step DPP reduction inside the
//                        16-lane row (VALU only, four columns per instruction); the update (hotloop!, src:156-160) NR fma.
//                        Reflector j + 1 is built by a BUILDER wave from a copy of column j + 1 handed over through LDS
//                        (norm in double-double like every other path, src:129-135) while the matrix waves apply reflector
//                        j; v travels through 2 x 16 NR doubles of LDS (double buffered by column parity).  The kernels are
//                        bound by instruction issue (one or two waves per SIMD, 6.3 cycles per FP64 instruction), not by
//                        latency or memory.
//   k_small_ldiv<RPL>    b <- Q'b (src:215-224) and the back substitution (src:244-254) in one launch: wave 0 keeps b in
//                        registers and walks the columns, waves 1-3 stream the factor in 16-column chunks into a
//                        double-buffered LDS stage ahead of it (once left to right for Q'b, once right to left -- upper
//                        triangle only -- for R).
//
// Both kernels take plain pointers that may be device memory OR pinned host memory: the host-array entry points
// (dhqr_qr_f64 / dhqr_ldiv_f64) hand them their pinned staging buffer, so a call is memcpy -> one launch -> one
// synchronisation -> memcpy, with no hipMemcpy on the path at all (the kernel's first loads / last stores cross PCIe
// themselves: 88 KB at 110 x 100).
#pragma once
#include <type_traits>
#include "dhqr_common.h"

// End of a small-route kernel launched by a host-array entry point: `done` (pinned host memory, may be nullptr) <- epoch
// once every store of the workgroup is visible to the host.  The host polls that word instead of waiting in
// hipStreamSynchronize, whose wake-up (an interrupt and a thread switch) costs more than the copy of a 110 x 100 matrix.
// EVERY thread of the workgroup calls it.
__device__ __forceinline__ void small_signal_done(unsigned long long *done, unsigned long long epoch) {
  if (done == nullptr) return;  // (uniform)
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define SMQ_THREADS 512  // k_small_qr: 8 waves, two per SIMD, 256 registers per lane
#define SMQ_GW 32        // columns per group: 8 waves x 4 DPP rows
#define SML_THREADS 256  // k_small_ldiv

// sum over the 16 lanes of a DPP row; every lane of the row ends with the row's total
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x124, 0xf>(v);  // row_ror:4
  v += dpp_f64<0x128, 0xf>(v);  // row_ror:8
  return v;
}
__device__ __forceinline__ double smq_readlane(double v, int lane) {  // lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Wave sum of a double-double (lane 63's total, broadcast; the DPP pattern of wave_sum_dpp, dhqr_common.h) with the TREE in
// plain double, separately for the high and the low parts.  What is lost is the rounding of six additions of the high parts
// (<= 3 eps of the sum) -- the products and the per-lane sums stay exact to twice the working precision.  Used for the dot
// products of the solve (rounded to double anyway) and for the reflector norms: a cascaded tree (TwoSum per step, 14
// instead of 6 instructions) was measured and bought nothing -- the acceptance statistic over 16-24 draws at 110 x 100:
// median 0.6-0.9 x LAPACK's either way (the restated reference: 1.4-1.7 x, up to 11.6 x) -- while both kernels are bound by
// instruction issue.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ dhqr_dd dd_plain_step(const dhqr_dd v) {
  dhqr_dd r;
  r.hi = v.hi + dpp_f64<CTRL, ROW_MASK>(v.hi);
  r.lo = v.lo + dpp_f64<CTRL, ROW_MASK>(v.lo);
  return r;
}
__device__ __forceinline__ dhqr_dd wave_sum_dd_plain(dhqr_dd v) {
  v = dd_plain_step<0xB1, 0xf>(v);
  v = dd_plain_step<0x4E, 0xf>(v);
  v = dd_plain_step<0x124, 0xf>(v);
  v = dd_plain_step<0x128, 0xf>(v);
  v = dd_plain_step<0x142, 0xa>(v);
  v = dd_plain_step<0x143, 0xc>(v);
  dhqr_dd r;
  r.hi = smq_readlane(v.hi, 63);
  r.lo = smq_readlane(v.lo, 63);
  return r;
}

#define SMB_THREADS (SMQ_THREADS + 64)
#ifdef DHQR_BENCH_BUILD
// phase clock of k_small_qr_d (libdhqr_bench.so only; dhqr_debug_smq_phases): shader cycles summed over the steps of the last
// launches, per wave: [0] reflector read + trailing update (take-back and hand-over inside), [1] reflector construction
// (builder), [2] wait at the barrier, [5] steps
__device__ unsigned long long g_smq_phase[9][6];
#define SMQ_CLK(var) const long long var = clock64()
#else
#define SMQ_CLK(var) do { } while (0)
#endif
// householder!(A, alpha) (src:113, 122-148, 198-213) for m <= 16 NR, n <= 32 NQ, m >= n.  Asrc / Adst may alias.
// ONE barrier per column.  History of the step, because each version's measurement is why the next looks as it does
// (110 x 100 / 220 x 200, one launch):
//   1. the owner of column j + 1 updates it, builds reflector j + 1 (norm, square roots, scaling), then does its share of the
//      trailing update: ~1000 instructions at 6.3 cycles each where the other seven waves had 350 and waited: 169 / 602 us;
//   2. the construction moves to a BUILDER wave (m <= 128: a ninth wave that holds no part of the matrix; otherwise the wave
//      four places from the owner, after its own share of the update); the owner only brings its column up to date and hands
//      it over through LDS; two barriers per column ("column in xcol", "reflector in vb"); the matrix copy of a finished
//      column is scaled at the very end: 125 / 560, then 100 / 495 us with the refinement chains shortened.  Its phase clock
//      (tools/smq_phases.py, cycles per column: owner's update + hand-over 500 / 1300 in front of the first barrier, then the
//      construction 1713 / 380 beside a trailing update of 700-1050 / 1770-2070) shows two things on the critical chain that
//      need not be there: the owner's special update of the look-ahead column, and the barrier behind it;
//   3. (this kernel) the BUILDER applies reflector j to column j + 1 itself: the owner hands the column over one step
//      EARLIER -- updated through reflector j - 1, inside its ordinary update of step j - 1 -- and the builder (64 lanes x RBL
//      rows) forms v_j' x, updates its copy, and goes on to norm / alpha / f / the scaled reflector j + 1.  The eight matrix
//      waves apply reflector j to every column meanwhile -- column j + 1 included, no special case -- and the owner of column
//      j + 2 hands that one over as it updates it.  One barrier.
// The builder's copy of a column and the owner's differ in the rounding of one dot product (16-lane rows against 64 lanes),
// so the owner TAKES THE REFLECTOR BACK: at step j + 1 it reads reflector j + 1 from LDS like every wave, in exactly the
// layout of its matrix registers, and the lanes of column j + 1 overwrite rows >= j + 1 with it.  What is stored is therefore
// bit for bit what was applied, and there is no deferred scaling and no LDS image of the reflectors.
// (A builder in the 16-lane layout of the matrix waves -- bitwise the owner's arithmetic, no take-back -- was measured too:
//  135 / 647 us against 115 / 440: four times the per-lane work on the one chain that matters, and 100 more spilled registers.)
// One pass of reflector j (vr: its rows rg + 16 r, zeros above row j) over the column groups of this wave, rows r >= R0
// only (the caller knows that every row below 16 R0 lies above j).  Straight-line code per group -- and ONE uniform branch
// per group skips the groups that are finished (blocks of 2 and 4 groups per branch were measured: with two waves per SIMD
// hiding each other's latencies the finer skipping wins).
//   qt / tlane: the group / lanes of column j itself when this wave holds it: its rows >= j become reflector j (the take-back);
//   qh / hlane: the group / lanes of column j + 2 when this wave holds it: handed to the builder (xo) as it is updated.
// Both happen INSIDE the group's update: as separate passes over the registers the compiler kept a shadow copy of the group
// in scratch memory.
template <int NR, int NQ, int R0>
__device__ __forceinline__ void smq_pass_d(double (&a)[NQ][NR], const double (&vr)[NR], int j, int cbase, double *xo, int qh,
                                           bool hlane, int qt, bool tlane) {
  const int rg = threadIdx.x & 15;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (SMQ_GW * (q + 1) - 1 >= j) {  // (>=: the group of column j itself still takes its reflector back)
      if (q == qt) {                  // this wave holds column j: rows >= j of it become reflector j (src:133-140)
#pragma unroll
        for (int r = R0; r < NR; ++r) a[q][r] = (tlane && rg + 16 * r >= j) ? vr[r] : a[q][r];
      }
      double p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int r = R0; r < NR; r += 2) {
        p0 = fma(vr[r], a[q][r], p0);  // src:42-49
        if (r + 1 < NR) p1 = fma(vr[r + 1], a[q][r + 1], p1);
      }
      const double p = row16_sum(p0 + p1);
      const double d = (SMQ_GW * q + cbase > j) ? p : 0.0;  // columns <= j are finished, columns >= n hold zeros
#pragma unroll
      for (int r = R0; r < NR; ++r) a[q][r] = fma(-vr[r], d, a[q][r]);  // src:156-160, src:209
      if (q == qh) {  // column j + 2 leaves for the builder as it is updated
#pragma unroll
        for (int r = 0; r < NR; ++r)
          if (hlane) xo[rg + 16 * r] = a[q][r];
      }
    }
  }
}
// FLAGS (the instantiations above 128 rows, where a matrix wave builds): NO barrier in the column loop.  With a barrier the
// step is (update + the builder's chain): the seven other waves wait while the builder, after its own share of the update,
// runs the chain (~4100 cycles every eighth step for each wave against ~2000 of update).  Here the waves run free: v_j is
// awaited through an LDS word (vready), a handed-over column through another (xready), eight buffers of each instead of two,
// and a builder may not overwrite reflector j - 8's buffer before every wave has finished with it (one progress byte per
// wave, polled as one 64-bit word): a wave that has just built is one chain behind and catches up over the next seven
// steps, so a column costs max(chain, update + chain / 8).  The builder is a different wave for every column and builds
// BEFORE its own share of the update (otherwise that update sits between two builds of the chain); one column
// group of its matrix registers waits in LDS meanwhile (the chain needs ~40 registers; left to the allocator they went to
// scratch memory).  A writer publishes with a workgroup release fence (s_waitcnt lgkmcnt(0)) and an LDS store by lane 0.
// Measured at 220 x 200 (us per qr! call): barrier 440 | flags, builder after its update, same builder for four columns 724
// | builder first 544 | one poll for all progress bytes, no s_sleep 475 | a new builder every column 410 | eight buffers 383
// | s_setprio 3 while building 375 | polls as LDS atomics instead of volatile (= flat, system-coherent) loads 334.
// The waits are BOUNDED (a single-workgroup kernel that never ends would take the device with it): after `limit` polls
// (SMQ_SPIN_LIMIT: tens of milliseconds; a hand-over takes a microsecond; DHQR_TUNE small_spin_limit, which the tests set to 0
// to see the answer) the waiter sets `broken` and goes on, and the kernel returns NaN in alpha instead of a factorisation;
// the host-array entry point then factors once more with the barrier form (dhqr_qr_f64).
#define SMQ_SPIN_LIMIT (1 << 20)
// (relaxed workgroup-scope atomics, not `volatile`: a volatile access through a pointer loses the LDS address space and
// becomes a FLAT load with system-coherence bits -- measurably slower polls on the chain)
__device__ __forceinline__ void smq_wait_ge(int *p, int target, int *broken, int limit) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(0);  // (no pause on the device: the poll is on the chain; the CPU emulator yields here)
    if (++spins > limit) {
      __hip_atomic_store(broken, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// every wave's progress in ONE 64-bit word (byte w = 1 + the last reflector wave w is done with): one LDS round trip per poll
__device__ __forceinline__ void smq_wait_all_ge(unsigned long long *p, int target, int *broken, int limit) {
  for (int spins = 0;; ++spins) {
    const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 8; ++q) ok = ok && (int)((v >> (8 * q)) & 0xffull) >= target + 1;
    if (ok) break;
    __builtin_amdgcn_s_sleep(0);
    if (spins >= limit) {
      __hip_atomic_store(broken, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <typename T>
__device__ __forceinline__ void smq_post(T *p, T value, bool leader) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();  // (a wave runs in lockstep; on the CPU emulator its lanes do not: they meet here)
  if (leader) __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int NR, int NQ, bool EXTRA, bool FLAGS = false>
__global__ __launch_bounds__(EXTRA ? SMB_THREADS : SMQ_THREADS) void k_small_qr_d(const double *Asrc, int64_t lds, double *Adst,
                                                                                   int64_t ldd, int m, int n,
                                                                                   double *__restrict__ alpha, int spin_limit,
                                                                                   unsigned long long *done,
                                                                                   unsigned long long epoch) {
  constexpr int RBL = (16 * NR + 63) / 64;  // rows of a column per lane of the builder
  constexpr int NT = EXTRA ? SMB_THREADS : SMQ_THREADS;
  constexpr int NBUF = FLAGS ? 8 : 2, BM = NBUF - 1;  // reflector / column buffers (by column index)
  static_assert(!(FLAGS && EXTRA), "the flag form is for the instantiations whose builder is a matrix wave");
  __shared__ double vb[NBUF][64 * RBL];
  __shared__ double xcol[NBUF][64 * RBL];
  __shared__ double als[SMQ_GW * NQ];
  constexpr int NPARK = 1;  // FLAGS: column groups of the builder's registers parked in LDS while it builds
  __shared__ double park[FLAGS ? 8 : 1][FLAGS ? NPARK * NR * 64 : 1];  // (one area per wave)
  __shared__ unsigned long long sync_words[3];  // FLAGS: [0] = {vready, xready} (two ints), [1] = the eight waves' progress bytes, [2] = broken
  int *const vready = reinterpret_cast<int *>(sync_words), *const xready = vready + 1;
  unsigned char *const prog = reinterpret_cast<unsigned char *>(sync_words + 1);
  unsigned long long *const progall = sync_words + 1;
  int *const broken = reinterpret_cast<int *>(sync_words + 2);
  const int t = threadIdx.x, w = t >> 6, l = t & 63, rg = l & 15, cs = l >> 4;
  const bool xwave = EXTRA && w == 8;  // holds no part of the matrix
  const int cbase = 4 * w + cs;        // this lane's column of group q: 32 q + cbase (matrix waves)
  double a[NQ][NR];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = rg + 16 * r, col = SMQ_GW * q + cbase;
      a[q][r] = 0.0;
      if (!xwave && row < m && col < n) a[q][r] = Asrc[(int64_t)row + (int64_t)col * lds];
    }
  if (xwave) __builtin_amdgcn_s_setprio(3);  // the builder's dependent chain IS the step: its instructions go first
  for (int i = 16 * NR + t; i < 64 * RBL; i += NT)  // rows beyond the matrix registers: never written again
#pragma unroll
    for (int q = 0; q < NBUF; ++q) xcol[q][i] = vb[q][i] = 0.0;
  if (FLAGS && t == 0) {
    *vready = -1;
    *xready = -1;
    sync_words[1] = 0ull;
    sync_words[2] = 0ull;
  }
  // the prologue's hand-over (a step hands over inside its update of the group: see smq_pass_d)
  auto hand_over = [&](int jc) __attribute__((always_inline)) {
    const int q1 = jc / SMQ_GW, cs1 = jc & 3;
    double *xo = xcol[jc & BM];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q == q1) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
          if (cs == cs1) xo[rg + 16 * r] = a[q][r];
      }
  };
  // builder: column jn (in xcol, updated through reflector jn - 2) -> apply reflector jn - 1 (in vb) -> reflector jn
  // (src:129-135) -> the other vb buffer, alpha
  auto build = [&](int jn) __attribute__((always_inline)) {
    const double *xi = xcol[jn & BM], *vi = vb[(jn - 1) & BM];
    double x[RBL];
#pragma unroll
    for (int r = 0; r < RBL; ++r) x[r] = xi[l + 64 * r];
    if (jn > 0) {  // src:198-213 for this one column (vi holds zeros above row jn - 1)
      double vp[RBL], p = 0.0;
#pragma unroll
      for (int r = 0; r < RBL; ++r) {
        vp[r] = vi[l + 64 * r];
        p = fma(vp[r], x[r], p);
      }
      const double d = wave_sum_dpp(p);
#pragma unroll
      for (int r = 0; r < RBL; ++r) x[r] = fma(-vp[r], d, x[r]);
    }
    double hc = 0.0;
    dhqr_dd acc = {0.0, 0.0};
#pragma unroll
    for (int r = 0; r < RBL; ++r) {
      if (r == (jn >> 6)) hc = x[r];
      dd_add_sq(acc, l + 64 * r >= jn ? x[r] : 0.0);  // rows >= m hold zeros
    }
    const double h = smq_readlane(hc, jn & 63);
    // (the squares and the per-lane sums in double-double, the tree across the lanes in plain double: see wave_sum_dd_plain)
    const dhqr_dd ss = wave_sum_dd_plain(acc);
    const double s2 = ss.hi + ss.lo;
    double sn, f;
    if (s2 > 0.0 && s2 < 1e300) {
      double rinv, sq;
      dhqr_sqrt_rsqrt(s2, sn, rinv);                 // src:129
      dhqr_sqrt_rsqrt(fma(sn, fabs(h), s2), sq, f);  // src:131: s (s + |h|) = s^2 + s |h|
    } else {
      sn = sqrt(s2);
      f = 1.0 / sqrt(sn * (sn + fabs(h)));
    }
    const double al = sn * dhqr_alphafactor(h);      // src:130
    const double piv = (h - al) * f;                 // src:132
    double *vo = vb[jn & BM];
#pragma unroll
    for (int r = 0; r < RBL; ++r) {
      const int row = l + 64 * r;
      if (row < 16 * NR) vo[row] = row > jn ? x[r] * f : (row == jn ? piv : 0.0);  // src:133-140
    }
    if (l == 0) als[jn] = al;
  };
  auto step = [&](auto r0c, int j) __attribute__((always_inline)) {
    constexpr int R0 = decltype(r0c)::value;
    const int jn = j + 1, wn = (jn & (SMQ_GW - 1)) >> 2;  // wn: the wave that holds column j + 1
    SMQ_CLK(tq0);
    SMQ_CLK(tf0);
    SMQ_CLK(tf1);
    if constexpr (FLAGS) {
      smq_wait_ge(vready, j, broken, spin_limit);
#ifdef DHQR_BENCH_BUILD
      const long long tf1b = clock64();
#endif
      // the builder of reflector j + 1 builds FIRST (the chain of the whole factorisation runs through the builds) and owes
      // its share of reflector j's update afterwards.  Two column groups of its matrix registers wait in LDS meanwhile: the
      // chain needs ~40 registers and the allocator otherwise spills three times as many to scratch memory around it.
      // (the builder changes with EVERY column here -- wn + 2, + 4, + 6, + 8 inside a block of four columns, the odd ones in the
      // next block: eight different waves in eight steps, never the wave that is about to hand a column over -- so that a
      // builder's own update is not what the next build waits for)
      if (jn < n && w == ((wn + 2 + 2 * (jn & 3)) & 7)) {
        smq_wait_ge(xready, jn, broken, spin_limit);
        if (jn >= NBUF) smq_wait_all_ge(progall, jn - NBUF, broken, spin_limit);  // the buffer reflector jn goes to still holds reflector jn - 8
        __builtin_amdgcn_s_setprio(3);  // the chain goes first on its SIMD (the other wave there is in its update)
        double *pk = park[w];
#pragma unroll
        for (int q = NQ - NPARK; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < NR; ++r) pk[((q - (NQ - NPARK)) * NR + r) * 64 + l] = a[q][r];
        build(jn);
        smq_post(vready, jn, l == 0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int q = NQ - NPARK; q < NQ; ++q)
#pragma unroll
          for (int r = 0; r < NR; ++r) a[q][r] = pk[((q - (NQ - NPARK)) * NR + r) * 64 + l];
      }
#ifdef DHQR_BENCH_BUILD
      if (l == 0) {  // FLAGS: [3] wait for reflector j, [4] the build block (waits, parking, chain, publication)
        g_smq_phase[w][3] += (unsigned long long)(tf1b - tf0);
        g_smq_phase[w][4] += (unsigned long long)(clock64() - tf1b);
      }
#endif
    }
    if (!xwave) {
      double vr[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) vr[r] = (r >= R0) ? vb[j & BM][rg + 16 * r] : 0.0;
      const int jc = j + 2;
      const bool mine = jc < n && w == ((jc & (SMQ_GW - 1)) >> 2), back = w == ((j & (SMQ_GW - 1)) >> 2);
      smq_pass_d<NR, NQ, R0>(a, vr, j, cbase, xcol[jc & BM], mine ? jc / SMQ_GW : -1, cs == (jc & 3), back ? j / SMQ_GW : -1,
                             cs == (j & 3));
      if constexpr (FLAGS) {
        if (mine) smq_post(xready, jc, l == 0);  // column j + 2 is in its buffer
        smq_post(prog + w, (unsigned char)(j + 1), l == 0);  // this wave is done with reflector j's buffer
      }
    }
    SMQ_CLK(tq1);
    if constexpr (!FLAGS) {
      if (jn < n && w == (EXTRA ? 8 : ((wn + 4) & 7))) build(jn);
    }
    SMQ_CLK(tq2);
    if constexpr (!FLAGS) __syncthreads();  // reflector j + 1 is in vb, column j + 2 in xcol, reflector j applied everywhere
#ifdef DHQR_BENCH_BUILD
    if (l == 0) {
      const long long tq3 = clock64();
      g_smq_phase[w][0] += (unsigned long long)(tq1 - tq0);
      g_smq_phase[w][1] += (unsigned long long)(tq2 - tq1);
      g_smq_phase[w][2] += (unsigned long long)(tq3 - tq2);
      g_smq_phase[w][5] += 1ull;
    }
#endif
  };
  if (w == 0) hand_over(0);
  __syncthreads();
  if (w == (EXTRA ? 8 : 4)) build(0);
  if (n > 1 && w == 0) hand_over(1);  // (as it came: the builder applies reflector 0 to it)
  __syncthreads();
  if (FLAGS && t == 0) {
    *vready = 0;
    *xready = 1;
  }
  __syncthreads();
  {
    constexpr int RA = NR / 4, RB = NR / 2, RC = (3 * NR) / 4;
    int j = 0;  // (the last step, j = n - 1, only takes reflector n - 1 back -- and updates nothing: no column beyond it)
    for (; j < n && j < 16 * RA; ++j) step(std::integral_constant<int, 0>{}, j);
    for (; j < n && j < 16 * RB; ++j) step(std::integral_constant<int, RA>{}, j);
    for (; j < n && j < 16 * RC; ++j) step(std::integral_constant<int, RB>{}, j);
    for (; j < n; ++j) step(std::integral_constant<int, RC>{}, j);
  }
  if constexpr (FLAGS) __syncthreads();  // alpha of the last columns
  if (!xwave) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int col = SMQ_GW * q + cbase;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int row = rg + 16 * r;
        if (row < m && col < n) Adst[(int64_t)row + (int64_t)col * ldd] = a[q][r];
      }
    }
  }
  for (int i = t; i < n; i += NT) alpha[i] = (FLAGS && *broken != 0) ? __builtin_nan("") : als[i];
  small_signal_done(done, epoch);
}

// solve_householder!(b, H, alpha) (src:284-294) for m <= 256: b (m) <- [x; tail of Q'b], xout (n, may be nullptr) <- x.
// b is carried in DOUBLE-DOUBLE through Q'b and the back substitution, in the reference's operation order (src:215-224
// reflectors in column order, src:244-254 from the last row up), x rounded to double once per entry -- what the
// ComplexF64 solve does (dhqr_complex.h, k_zqtb_col_dd): the reference's acceptance statistic ||A'(A x - b)||
// (test/runtests.jl:51,62) sees the rounding of the O(mn) solve next to that of the O(mn^2) factorisation, and a
// plain-double solve in this order scores like the restated reference itself, up to 11 x LAPACK on some draws
// (docs/DESIGN_rounds1-4.md, section 1).
// Awork != nullptr: the factor is first copied there (m x n, leading dimension m; device memory) with every load in
// flight at once -- A is then pinned HOST memory, and the chunk pipeline below would pay a PCIe round trip per chunk.
// The solving wave is bound by instruction issue (~150 double-double instructions per reflector at four rows per lane, one
// wave on its SIMD: 6.3 cycles each): RPL = ceil(m / 64) rows per lane, not always four.  Up to 128 rows the chunks are 64
// columns wide (the same LDS) and come straight from the caller's memory: two or three chunk hand-overs per pass instead of
// seven to sixteen, and no copy of the factor first.
#define SML_LDR 256  // most rows of a matrix the solve takes
template <int RPL, int CH>  // RPL: rows of b per lane of the solving wave, m <= 64 RPL; CH: columns per LDS chunk (2 x CH x 64 RPL doubles of LDS)
__global__ __launch_bounds__(SML_THREADS) void k_small_ldiv(const double *__restrict__ A, int64_t lda, int m, int n,
                                                            const double *__restrict__ alpha, const double *bin, double *bout,
                                                            double *xout, double *__restrict__ Awork, unsigned long long *done,
                                                            unsigned long long epoch) {
  constexpr int LDR = 64 * RPL;  // rows of a staged column
  __shared__ double buf[2][CH][LDR];
  __shared__ double als[2][CH];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int nch = (n + CH - 1) / CH;
  if (Awork) {
    const int total = m * n;
    for (int base = 0; base < total; base += 8 * SML_THREADS) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * SML_THREADS + t;
        x[u] = 0.0;
        if (idx < total) x[u] = A[(int64_t)(idx % m) + (int64_t)(idx / m) * lda];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * SML_THREADS + t;
        if (idx < total) Awork[idx] = x[u];
      }
    }
    __threadfence_block();
    __syncthreads();
    A = Awork;
    lda = m;
  }
  // waves 1..3: columns [16 c, 16 c + 16) rows [0, rtop) -> buf[c & 1] (zeros beyond the matrix); every load of a thread is
  // issued before its first LDS store
  auto stage = [&](int c, int rtop) {
    double(*B)[LDR] = buf[c & 1];
    constexpr int PER = (CH * (LDR / 2) + SML_THREADS - 64 - 1) / (SML_THREADS - 64);
    double x0[PER], x1[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = t - 64 + u * (SML_THREADS - 64);
      const int jj = e / (LDR / 2), row = 2 * (e % (LDR / 2)), col = CH * c + jj;
      x0[u] = 0.0;
      x1[u] = 0.0;
      if (jj < CH && col < n && row < rtop) {
        if (row < m) x0[u] = A[(int64_t)row + (int64_t)col * lda];
        if (row + 1 < m) x1[u] = A[(int64_t)row + 1 + (int64_t)col * lda];
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = t - 64 + u * (SML_THREADS - 64);
      const int jj = e / (LDR / 2), row = 2 * (e % (LDR / 2));
      if (jj < CH) {
        B[jj][row] = x0[u];
        B[jj][row + 1] = x1[u];
      }
    }
    if (t >= 64 && t < 64 + CH) als[c & 1][t - 64] = (CH * c + t - 64 < n) ? alpha[CH * c + t - 64] : 1.0;
  };
  dhqr_dd b[RPL];
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      b[r].hi = (l + 64 * r < m) ? bin[l + 64 * r] : 0.0;
      b[r].lo = 0.0;
    }
  } else {
    stage(0, m);
  }
  __syncthreads();
  // ---- b <- Q'b: reflectors left to right (src:215-224)
  for (int c = 0; c < nch; ++c) {
    if (w == 0) {
      const double(*B)[LDR] = buf[c & 1];
#pragma unroll 4
      for (int jj = 0; jj < CH; ++jj) {
        const int j = CH * c + jj;  // (columns >= n are staged as zeros: no-ops)
        double v[RPL];
        dhqr_dd p = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const int row = l + 64 * r;
          v[r] = row >= j ? B[jj][row] : 0.0;  // rows < j of a factored column hold R
          dd_add_prod(p, v[r], b[r].hi);       // src:217: sum v_i b_i, b_i = hi + lo
          p.lo = fma(v[r], b[r].lo, p.lo);
        }
        const dhqr_dd sd = wave_sum_dd_plain(p);
        const double s = sd.hi + sd.lo;
#pragma unroll
        for (int r = 0; r < RPL; ++r) dd_add_prod(b[r], -s, v[r]);  // src:218-220: b_i -= v_i s
      }
#pragma unroll
      for (int r = 0; r < RPL; ++r) dd_renorm(b[r]);  // once per chunk: the low parts stay small against the high ones
    } else if (c + 1 < nch) {
      stage(c + 1, m);
    }
    __syncthreads();
  }
  // ---- back substitution, columns right to left (src:244-254): x_j = b_j / alpha_j, b[0:j] -= R[0:j, j] x_j
  // chunk nch - 1 is staged first; its parity may collide with the buffer wave 0 read last, which the barrier above released
  if (w != 0) stage(nch - 1, n);
  __syncthreads();
  for (int c = nch - 1; c >= 0; --c) {
    if (w == 0) {
      const double(*B)[LDR] = buf[c & 1];
      const double rinv = dhqr_rcp(als[c & 1][l & (CH - 1)]);  // lane jj: 1 / alpha of the chunk's column jj
#pragma unroll 4
      for (int jj = CH - 1; jj >= 0; --jj) {
        const int j = CH * c + jj;
        if (j < n) {  // wave-uniform
          const int rj = j >> 6;
          double bj = 0.0;
#pragma unroll
          for (int r = 0; r < RPL; ++r)
            if (r == rj) bj = b[r].hi + b[r].lo;
          // src:251 b_j / alpha_j: reciprocal (computed for the whole chunk before the loop, off the chain) times b_j and one
          // correction step -- the quotient to the last bit in all but rare half-way cases, three dependent operations
          // instead of the ~25 of a division
          const double bq = smq_readlane(bj, j & 63), aj = als[c & 1][jj], ri = smq_readlane(rinv, jj);
          double xj = bq * ri;
          xj = fma(fma(-aj, xj, bq), ri, xj);
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const int row = l + 64 * r;
            if (row == j) {
              b[r].hi = xj;
              b[r].lo = 0.0;
            } else if (row < j) {
              dd_add_prod(b[r], -B[jj][row], xj);  // src:248-250
            }
          }
        }
      }
    } else if (c > 0) {
      stage(c - 1, CH * c);  // only rows above the chunk's last column are needed
    }
    __syncthreads();
  }
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int row = l + 64 * r;
      const double val = b[r].hi + b[r].lo;
      if (row < m) bout[row] = val;
      if (xout && row < n) xout[row] = val;
    }
  }
  small_signal_done(done, epoch);
}
