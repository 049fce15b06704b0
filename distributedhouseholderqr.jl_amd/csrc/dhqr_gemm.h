// dhqr_gemm.h -- FP64 MFMA kernels of the blocked (compact-WY) trailing update
//     C <- C - V * (op(T) * (V' * C))                         (BASELINE config 3/4)
// which is the blocked form of the reference's per-column partialdot + hotloop!
// (src/DistributedHouseholderQR.jl:198-213): k_gemm_tn is `s = partialdot(Hj, H[:,jj])` for 128
// reflectors at once, k_gemm_nn_sub is `H[:,jj] -= Hj*s`.
//
// Both kernels use v_mfma_f64_16x16x4_f64 (2048 flop / instruction, 64 cycles on a gfx950 SIMD):
//   D[i][j] += sum_k A[i][k] * B[k][j],  i,j in 0..15, k in 0..3
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, register g holds D[i = (l>>4) + 4*g][j = l&15]
// (f64 C/D map differs from the f32 family: cdna_hip_programming.md section 3; verified on the
// device by tests/test_gpu_kernels.py::test_mfma_layout_probe.)
// The MFMA "j" index is always mapped to the memory-contiguous direction of the output so every
// 16-lane group stores/loads one contiguous 128-byte segment.
//
// Tiles: 256 threads = 4 waves, output tile 128 x 128, each wave 64 x 64 = 4 x 4 MFMA tiles
// (16 accumulators x 4 doubles = 128 VGPRs), K-tile 16, one barrier per K-tile.  Two operand paths:
//  * the WIDE kernels (k_gemm_tn2; interior tiles of k_gemm_nn_quad / k_gemm_nn_sub) load global -> LDS directly
//    (glds16 below: no staging registers, unpadded images, the bank swizzle in the per-lane global address);
//  * everything else (k_gemm_tn, edge tiles, unaligned operands, the lane's 64-row tiles) stages global -> registers ->
//    LDS (double buffered, next tile's global loads in flight during the 64 MFMAs of the current one) with padded
//    strides: 18 doubles (36 dwords) for k-contiguous tiles, 130 doubles for the row-contiguous V tile of
//    k_gemm_nn_sub (read with ds_read_b128) -- conflict-free fragment reads (ds_read_b64, 64 banks).
#pragma once
#include "dhqr_common.h"
#include <type_traits>

#define G_KT 16          // K-tile depth
#define G_LDK 18         // LDS stride (doubles) of a k-contiguous tile column
#define G_LDR 130        // LDS stride (doubles) of a row-contiguous 128-row tile column (see k_gemm_nn_sub)

__device__ __forceinline__ dhqr_d4 mfma_f64(double a, double b, dhqr_d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- direct global -> LDS operand loads (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave instruction) -------------
// The LDS destination of lane l is  M0 (wave-uniform byte address) + 16 l  -- a linear image; the global source is per lane,
// so a swizzled LDS layout is made by permuting which 16-byte chunk each lane fetches.  The load is issued from inline
// assembly ON PURPOSE: with __builtin_amdgcn_global_load_lds hipcc orders every later ds_read behind it (s_waitcnt vmcnt(0)
// in front of the first fragment read of the CURRENT tile, i.e. no prefetch at all; distinct LDS objects per buffer did not
// change that).  Hidden from the compiler, the load is ordered by the caller: gemm_lds_landed() (vmcnt + s_barrier) between
// the issue and the first ds_read of that buffer by any wave.  Compiler-visible global loads issued BEFORE a batch of these
// are still waited for correctly (its vmcnt(N) under-counts what is in flight, which only waits longer); consume them before
// the next batch is issued (sched_barrier) or they drag the whole batch into their wait.  m0 is a reserved register (a clobber declaration is not honoured for it), so the
// statement saves and restores it: whatever the compiler keeps there (movrel, LDS-direct, sendmsg) survives; the load
// latches m0 when it issues (consecutive loads with different m0 were always issued back to back).  `lds` = any pointer into the workgroup's LDS, `byte_off` wave-uniform.
// Measured (tools/gemm_lab.hip, 16384^2, K = 512): 60.3 -> 67.9 TFLOP/s, bit-identical results; what goes away is the staging
// registers, 8 ds_write_b128 + ~30 masking VALU per K-tile and wave, and the vmcnt wait in front of them.
__device__ __forceinline__ void glds16(const double *g, double *lds, uint32_t byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + byte_off;
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "v"(g), "s"(__builtin_amdgcn_readfirstlane(a))
               : "memory");
#else
  double *d = reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + byte_off) + 2 * (threadIdx.x & 63);
  d[0] = g[0];
  d[1] = g[1];
#endif
}
// every direct load this wave issued, except the last `KEEP` vector-memory operations, has landed; then the workgroup meets
template <int KEEP>
__device__ __forceinline__ void gemm_lds_landed() {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (KEEP == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
#else
  __syncthreads();
#endif
}

// -------------------------------------------------------------------------------------------
// k_gemm_tn:  out[y][p + c*ldo] = sum_{r in slab y} V[r + p*ldv] * Ceff[r + c*ldc]
//   p in [0,128), c in [0,ncols); slab y = rows [y*rps, min(rows,(y+1)*rps)), rps % 16 == 0.
//   Ceff = sum_{q < ncsplit} C[q*csplit_stride + ...]  (folds a previous split-K reduction into
//   the operand load; ncsplit = 1 for the big trailing GEMM).
// grid = (ceil(ncols/128), nsplit).  Used for W = V'C (K = panel height), S = V'V, W = op(T)'W.
// VEC = 2: 16-byte global loads (host guarantees rows, ldv, ldc even and 16-byte aligned bases).
// All global loads are unconditional (clamped offset + select): no branch sits between a load
// and its use, so the whole next K-tile is in flight behind the current tile's 64 MFMAs.
// NBV = number of V columns (reflectors) = output rows: 128 for the trailing update, 32 / 64 for
// the narrow block reflectors inside a panel.  Wave layout: NBV=128 -> 2 (cols) x 2 (p) waves of
// 64 x 64; NBV=64 -> 4 x 1 waves of 32 cols x 64 p; NBV=32 -> 4 x 1 waves of 32 cols x 32 p.
// MASK (the batched Gram products of the solve, k_gemm_tn_gram_batch): V == C is a factored panel IN PLACE, whose top
// block still holds R above the diagonal -- element (row r of the panel, column p) counts as 0 where r < p.
template <int VEC, int NCS, int NBV, int MASK>  // MASK: 0 none, 1 panel in place (triangle + no padding columns), 2 no padding columns only
__device__ __forceinline__ void gemm_tn_body(const double *__restrict__ V, int64_t ldv,
                                             const double *__restrict__ C, int64_t ldc,
                                             int ncsplit, int64_t csplit_stride,
                                             int64_t rows, int64_t ncols, int64_t rps,
                                             double *__restrict__ out, int64_t ldo,
                                             int64_t osplit_stride, const unsigned bx, const unsigned by,
                                             const unsigned bz) {
  constexpr int NPI = (NBV >= 64) ? 4 : NBV / 16;  // p tiles per wave
  constexpr int WP = NBV / (16 * NPI);              // waves along p (1 or 2)
  constexpr int WC = 4 / WP;                        // waves along the columns
  constexpr int NCI = 128 / (16 * WC);              // column tiles per wave
  constexpr int NVL = NBV / 32;                     // V staging loads per thread (NBV*8 chunks / 256)
  __shared__ __attribute__((aligned(16))) double Vs[2][NBV * G_LDK];
  __shared__ __attribute__((aligned(16))) double Cs[2][128 * G_LDK];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int wc = (WP == 2) ? (w >> 1) : w, wp = (WP == 2) ? (w & 1) : 0;
  const int64_t c0 = (int64_t)bx * 128;
  const int64_t rbeg = (int64_t)by * rps;
  const int64_t rend = (rbeg + rps < rows) ? rbeg + rps : rows;
  const int nkt = (int)((rend - rbeg + G_KT - 1) / G_KT);
  const int ncv = (int)((ncols - c0 < 128) ? ncols - c0 : 128);  // valid columns in this tile

  dhqr_d4 acc[NCI][NPI];
#pragma unroll
  for (int a = 0; a < NCI; ++a)
#pragma unroll
    for (int b = 0; b < NPI; ++b) acc[a][b] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};

  // staging: tile = 128 columns x 16 rows; chunk q = t + i*256 -> column q/8, row pair q%8.
  // 32-bit element offsets from the (uniform) tile base.
  // blockIdx.z (0 in every launch but the narrow two-panel product, narrow_vtc in dhqr_api.hip): the z-th block of NBV
  // reflectors of V, whose products go NBV rows further down in `out`.
  const double *Vb = V + rbeg + (int64_t)bz * NBV * ldv;
  const double *Cb = C + rbeg + c0 * ldc;
  uint32_t offv[4], offc[4];
  bool okc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = t + i * 256;
    const int col = q >> 3;
    okc[i] = col < ncv;
    offv[i] = (uint32_t)((((MASK != 0) && !okc[i]) ? 0 : col) * ldv);  // MASK: V has no padding columns beyond ncols either
    offc[i] = (uint32_t)((okc[i] ? col : 0) * ldc);
  }
  // Software pipeline: load_tile only ISSUES the global loads of the next K-tile (raw values stay
  // in registers, nothing consumes them), the 64 MFMAs of the current tile run, and only then
  // store_tile masks the out-of-range lanes and writes LDS.  (Masking right after the load made
  // the compiler wait for vmcnt(0) BEFORE the MFMA block -- no overlap at all.)
  double2 sv[4], sc[4];
  auto load_tile = [&](int kt) {
    const int left = (int)(rend - rbeg) - kt * G_KT;  // valid rows in this K-tile (>0)
    const double *Vt = Vb + kt * G_KT;
    const double *Ct = Cb + kt * G_KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rp = (t + i * 256) & 7;
      if constexpr (VEC == 2) {
        const uint32_t ro = (2 * rp < left) ? 2 * rp : 0;  // rows even => pair all-or-nothing
        if (i < NVL) sv[i] = *reinterpret_cast<const double2 *>(Vt + (offv[i] + ro));
        if constexpr (NCS == 1) {
          sc[i] = *reinterpret_cast<const double2 *>(Ct + (offc[i] + ro));
        } else {
          double2 y = make_double2(0.0, 0.0);
          for (int qs = 0; qs < ncsplit; ++qs) {
            const double2 z = *reinterpret_cast<const double2 *>(Ct + (int64_t)qs * csplit_stride + (offc[i] + ro));
            y.x += z.x; y.y += z.y;
          }
          sc[i] = y;
        }
      } else {
        const uint32_t r0o = (2 * rp < left) ? 2 * rp : 0, r1o = (2 * rp + 1 < left) ? 2 * rp + 1 : 0;
        if (i < NVL) {
          sv[i].x = Vt[offv[i] + r0o];
          sv[i].y = Vt[offv[i] + r1o];
        }
        if constexpr (NCS == 1) {
          sc[i].x = Ct[offc[i] + r0o];
          sc[i].y = Ct[offc[i] + r1o];
        } else {
          double2 y = make_double2(0.0, 0.0);
          for (int qs = 0; qs < ncsplit; ++qs) {
            const double *Cq = Ct + (int64_t)qs * csplit_stride;
            y.x += Cq[offc[i] + r0o];
            y.y += Cq[offc[i] + r1o];
          }
          sc[i] = y;
        }
      }
    }
  };
  auto store_tile = [&](int buf, int kt) {
    const int left = (int)(rend - rbeg) - kt * G_KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = t + i * 256;
      const int col = q >> 3, rp = q & 7;
      const bool ok0 = 2 * rp < left, ok1 = 2 * rp + 1 < left;
      double2 x = sv[i], y = sc[i];
      if (!ok0) { x.x = 0.0; y.x = 0.0; }
      if (!ok1) { x.y = 0.0; y.y = 0.0; }
      if (!okc[i]) y = make_double2(0.0, 0.0);
      if constexpr (MASK != 0) {  // V == C, panel in place: no padding columns ...
        if (!okc[i]) x = make_double2(0.0, 0.0);
        if constexpr (MASK == 1) {  // ... and zero above the diagonal of the top block
          const int64_t r0 = rbeg + (int64_t)kt * G_KT + 2 * rp;
          if (r0 < col) { x.x = 0.0; y.x = 0.0; }
          if (r0 + 1 < col) { x.y = 0.0; y.y = 0.0; }
        }
      }
      if (i < NVL) *reinterpret_cast<double2 *>(&Vs[buf][col * G_LDK + 2 * rp]) = x;
      *reinterpret_cast<double2 *>(&Cs[buf][col * G_LDK + 2 * rp]) = y;
    }
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(0, 0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the load issue above, their consumers below the MFMAs
    const double *cs = &Cs[buf][(wc * (NCI * 16) + i16) * G_LDK + k4];
    const double *vs = &Vs[buf][(wp * 64 + i16) * G_LDK + k4];
#pragma unroll
    for (int kk = 0; kk < G_KT / 4; ++kk) {
      double a[NCI], b[NPI];
#pragma unroll
      for (int x = 0; x < NCI; ++x) a[x] = cs[x * 16 * G_LDK + kk * 4];
#pragma unroll
      for (int x = 0; x < NPI; ++x) b[x] = vs[x * 16 * G_LDK + kk * 4];
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) acc[ci][pi] = mfma_f64(a[ci], b[pi], acc[ci][pi]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nkt) store_tile(buf ^ 1, kt + 1);
    __syncthreads();
  }

  double *o = out + (int64_t)by * osplit_stride + c0 * ldo + (int64_t)bz * NBV;
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = wc * (NCI * 16) + ci * 16 + k4 + 4 * g;
      if (cl < ncv) {
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi)
          o[(uint32_t)(wp * 64 + pi * 16 + i16) + (uint32_t)(cl * ldo)] = acc[ci][pi][g];
      }
    }
}

template <int VEC, int NCS, int NBV>
__global__ __launch_bounds__(256, 2) void k_gemm_tn(const double *__restrict__ V, int64_t ldv,
                                                    const double *__restrict__ C, int64_t ldc,
                                                    int ncsplit, int64_t csplit_stride,
                                                    int64_t rows, int64_t ncols, int64_t rps,
                                                    double *__restrict__ out, int64_t ldo,
                                                    int64_t osplit_stride) {
  gemm_tn_body<VEC, NCS, NBV, 0>(V, ldv, C, ldc, ncsplit, csplit_stride, rows, ncols, rps, out, ldo, osplit_stride,
                                     blockIdx.x, blockIdx.y, blockIdx.z);
}

// Batched Gram products of the solve (dhqr_qtb.h): S_k = V_k' V_k for EVERY 128-column panel of a factored matrix in one
// launch, V_k read in place (MASK).  unit_first[k] = index of panel k's first (panel, rps-row slab) unit, unit_first[np] =
// number of units = gridDim.x; unit u of panel k writes its 128 x 128 partial sum to out + u * 128 * 128 (summed in slab
// order by k_qtb_sum_gram).  A panel is rows [128 k, m) x columns [128 k, 128 k + w_k) of A.
// MASKED = false (r6, the row split at P > 1): a rank whose rows all lie BELOW the panels' top blocks -- every panel is the
// rank's whole row range (rows [0, m) of A, columns of panel k), nothing to mask.
template <int VEC, bool MASKED = true>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_gram_batch(const double *__restrict__ A, int64_t lda, int64_t m,
                                                               int64_t n, int64_t rps, const int *__restrict__ unit_first,
                                                               int np, double *__restrict__ out, const int *__restrict__ skip) {
  __shared__ int kk_s;
  if (*skip) return;  // the context kept this factor's T' (dhqr_qtb.h, k_qtb_same_alpha): nothing to compute
  if (threadIdx.x == 0) {  // largest k with unit_first[k] <= blockIdx.x
    int lo = 0, hi = np - 1;
    const int u = (int)blockIdx.x;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (unit_first[mid] <= u) lo = mid; else hi = mid - 1;
    }
    kk_s = lo;
  }
  __syncthreads();
  const int k = kk_s;
  const int64_t c0 = (int64_t)k * DHQR_NBV;
  const int64_t w = (n - c0 < DHQR_NBV) ? n - c0 : DHQR_NBV;
  const double *P = MASKED ? A + c0 + c0 * lda : A + c0 * lda;
  const unsigned y = blockIdx.x - (unsigned)unit_first[k];
  gemm_tn_body<VEC, 1, DHQR_NBV, (MASKED ? 1 : 2)>(P, lda, P, lda, 1, (int64_t)0, MASKED ? m - c0 : m, w, rps,
                                         out + (int64_t)blockIdx.x * (DHQR_NBV * DHQR_NBV) - (int64_t)y * (DHQR_NBV * DHQR_NBV),
                                         (int64_t)DHQR_NBV, (int64_t)(DHQR_NBV * DHQR_NBV), 0u, y, 0u);
}

// -------------------------------------------------------------------------------------------
// k_gemm_tn2:  the W = [V_a V_b]' C pass of a TWO-panel update in one kernel, so C is read once:
//   out[y][p + c*256] = sum_{r in slab y} V[r + p*ldv] * C[r + c*ldc],   p in [0,256)
// (V_b is stored shifted down by 128 rows with zeros above, dhqr_dist.h, so both halves share the row range.)
// Launch: 1-D grid of min(units, #CU - spare) workgroups (dhqr_api.hip: wide_grid).
// 512 threads = 8 waves as 2 (columns) x 4 (p), each wave the same 64 x 64 block of 4 x 4 MFMA tiles as in
// k_gemm_tn; output tile 128 columns x 256 reflectors.  Per K-tile the workgroup stages 48 KB for 1.05 MFLOP
// (21.8 flop/B through the CU memory pipe, against 16 for two 128-reflector passes).  One workgroup per CU
// (110 KB of LDS), i.e. the same 2 waves per SIMD as two k_gemm_tn workgroups.
// SK ("stream-K", wide launches): the fine units (column tile, rps-row slab) are numbered TILE-major and workgroup b takes
// the CONTIGUOUS range [b q, (b + 1) q): at most a few (tile, row range) segments, each ONE K loop whose accumulators run
// over the whole row range -- every workgroup does the same number of K-tiles whatever ntiles is (no rounds, no idle
// eighth of the chip at 224 column tiles), and a tile's partial sums are as many as workgroups share it (2-3) instead of
// one per row slab (up to 12): the split-K partials written and re-read by the reduction shrink accordingly.  Segment of
// workgroup b in tile t goes to partial slot (b - floor(t S / q), t) with S = slabs per tile; k_reduce_pieces sums a
// tile's floor(((t + 1) S - 1) / q) - floor(t S / q) + 1 slots.
// VEC = 2 (r5): operands by direct global -> LDS loads into a ring of THREE 48 KiB stages (glds16 above; 6 loads per wave and
// K-tile, issued two K-tiles ahead: the leader of an XCD's workgroups pays an HBM round trip for every operand slice, which
// one K-tile of distance does not cover), unpadded swizzled images (column c = 8 chunks of 16 B, slot s holds chunk
// s ^ ((c >> 1) & 7)), no staging registers / ds_write / masking.  A slab whose last K-tile is partial stages that one tile
// through registers (zeros below the slab) into the same image.  Same unit decomposition, same sums in the same order:
// bit-identical to the register-staged program, which stays as the VEC = 1 instantiation.
// ROW GROUPS of a stream-K launch (r6; R = 1: the map above, unchanged).  The counters say k_gemm_tn2 fetches 2.6 x its
// algorithmic bytes: per column tile a workgroup streams 2 KiB of [V_a V_b] for every KiB of C, and with the units
// numbered tile-major over ALL rows the 32 workgroups of an XCD sit at 32 different heights of V -- a slice of V (67 MB at
// 32768 rows) is gone from the XCD's 4 MB L2 long before a second workgroup of that XCD asks for it.  With R row groups
// the rows are cut into R ranges and XCD x (workgroup b runs on XCD b % 8: round-robin dispatch) only ever works inside
// range x % R: its slice of V is rows / R x 2 KiB (8 MB at 32768 rows, R = 8; 2-4 MB for most of a factorisation), its
// workgroups sweep that one range tile after tile, and what one of them fetched the others find in the L2.  Each group
// is a stream-K problem of its own: fine units (column tile, slab of the group) numbered tile-major, the group's Gl
// workgroups take contiguous ranges of q_g = ceil(units / Gl).  Price: a column tile's partial sums are R x (1-3) pieces
// instead of 1-3 (slot g * P + piece; k_reduce_pieces walks the groups in order).
struct tn2_sk_group {
  int64_t Gl, nsl, slab0, q;  // workgroups of the group, its slabs, its first slab, units per workgroup
};
__host__ __device__ __forceinline__ tn2_sk_group tn2_sk_group_of(int g, int R, int64_t G, int64_t S, int64_t ntiles, int64_t skq) {
  tn2_sk_group r;
  if (R <= 1) {
    r.Gl = G, r.nsl = S, r.slab0 = 0, r.q = skq;
    return r;
  }
  const int64_t full = G >> 3, rem = G & 7;
  r.Gl = full * (8 / R) + (rem > g ? (rem - g - 1) / R + 1 : 0);
  const int64_t Sg = (S + R - 1) / R;
  r.slab0 = (int64_t)g * Sg;
  r.nsl = (r.slab0 >= S) ? 0 : ((S - r.slab0 < Sg) ? S - r.slab0 : Sg);
  r.q = r.Gl > 0 ? (ntiles * r.nsl + r.Gl - 1) / r.Gl : 1;
  if (r.q < 1) r.q = 1;
  return r;
}
__host__ __device__ __forceinline__ int tn2_sk_pieces(const tn2_sk_group &gr) {  // most pieces a column tile gets from this group
  if (gr.nsl <= 0) return 0;
  return (gr.q >= gr.nsl) ? 2 : (int)((gr.nsl + gr.q - 1) / gr.q) + 1;
}
template <bool SK, int OPT = 0>
__device__ __forceinline__ void gemm_tn2_direct(const double *__restrict__ V, int64_t ldv, const double *__restrict__ C,
                                                int64_t ldc, int64_t rows, int64_t ncols, int64_t rps,
                                                double *__restrict__ out, int64_t osplit_stride, int64_t skq,
                                                int rgroups = 1, int pstride = 0) {
  constexpr int NP = 256, STG = (NP + 128) * G_KT;  // doubles per stage: V image (256 columns) then C image (128)
  __shared__ __attribute__((aligned(1024))) double ring[3 * STG];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int wc = w >> 2, wp = w & 3;
  const int64_t ntiles = (ncols + 127) / 128, nslab_all = (rows + rps - 1) / rps;
  // SK: this workgroup's row group (rgroups == 1: one group, the whole launch), its index in the group
  int64_t nslab = nslab_all, slab0 = 0, wgi = (int64_t)blockIdx.x, pbase = 0;
  if constexpr (SK) {
    if (rgroups > 1) {
      const int x = (int)(blockIdx.x & 7u), g = x % rgroups;
      const tn2_sk_group gr = tn2_sk_group_of(g, rgroups, (int64_t)gridDim.x, nslab_all, ntiles, skq);
      nslab = gr.nsl;
      slab0 = gr.slab0;
      skq = gr.q;
      wgi = (int64_t)(blockIdx.x >> 3) * (8 / rgroups) + x / rgroups;
      pbase = (int64_t)g * pstride;
    }
  }
  const int64_t ufirst = SK ? wgi * skq : (int64_t)blockIdx.x;
  const int64_t ulast = SK ? (ufirst + skq < ntiles * nslab ? ufirst + skq : ntiles * nslab) : ntiles * nslab;
  // fragment addresses (doubles from the stage base): k-step kk, column i16 (+16 x) of the wave's 64
  int af[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) af[kk] = i16 * G_KT + (((2 * kk + (k4 >> 1)) ^ (i16 >> 1)) * 2) + (k4 & 1);
  // this wave's loads per K-tile: V column groups 4 w .. 4 w + 3 (8 columns each), C column groups 2 w, 2 w + 1
  // lane -> (column lcol of the group, chunk): group g = columns 8 g .. 8 g + 7, so (column >> 1) & 7 = (4 (g & 1) + (lcol >> 1)) & 7;
  // the i-th V group of a wave is 4 w + i, the i-th C group 2 w + i: the parity of g is the parity of i
  const int lcol = lane >> 3;
  const int ljp[2] = {(lane & 7) ^ (lcol >> 1), (lane & 7) ^ (4 + (lcol >> 1))};
  uint32_t gv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gv[i] = (uint32_t)((8 * (4 * w + i) + lcol) * ldv) + 2 * ljp[i & 1];
  for (int64_t u = ufirst; u < ulast;) {
    int64_t ux, uy, rbeg, rend;
    if constexpr (SK) {
      ux = u / nslab;
      const int64_t s0 = u - ux * nslab;
      const int64_t s1 = (s0 + (ulast - u) < nslab) ? s0 + (ulast - u) : nslab;
      rbeg = (slab0 + s0) * rps;
      rend = ((slab0 + s1) * rps < rows) ? (slab0 + s1) * rps : rows;
      uy = pbase + wgi - (ux * nslab) / skq;  // this workgroup's piece of tile ux
      u += s1 - s0;
    } else {
      ux = u % ntiles;
      uy = u / ntiles;
      rbeg = uy * rps;
      rend = (rbeg + rps < rows) ? rbeg + rps : rows;
      u += gridDim.x;
    }
    const int64_t c0 = ux * 128;
    const int nkt = (int)((rend - rbeg + G_KT - 1) / G_KT);
    const int ncv = (int)((ncols - c0 < 128) ? ncols - c0 : 128);
    const int tail = (int)(rend - rbeg) - (nkt - 1) * G_KT;  // valid rows of the last K-tile (1..16)
    const double *Vb = V + rbeg;
    const double *Cb = C + rbeg + c0 * ldc;
    uint32_t gc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = 8 * (2 * w + i) + lcol;
      gc[i] = (uint32_t)((col < ncv ? col : 0) * ldc) + 2 * ljp[i & 1];  // columns beyond the matrix: any valid address, never stored
    }
    dhqr_d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
    auto issue_tile = [&](int kt, int stage) {
      const double *Vt = Vb + kt * G_KT;
      const double *Ct = Cb + kt * G_KT;
      double *st = ring + stage * STG;
      if (kt == nkt - 1 && tail < G_KT) {  // uniform; partial tile: through registers, rows >= tail are zeros
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          double2 x = make_double2(0.0, 0.0);
          if (2 * ljp[i & 1] < tail) x = *reinterpret_cast<const double2 *>(Vt + gv[i]);  // rows even: pair all-or-nothing
          *reinterpret_cast<double2 *>(st + (8 * (4 * w + i)) * G_KT + 2 * lane) = x;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double2 x = make_double2(0.0, 0.0);
          if (2 * ljp[i & 1] < tail) x = *reinterpret_cast<const double2 *>(Ct + gc[i]);
          *reinterpret_cast<double2 *>(st + NP * G_KT + (8 * (2 * w + i)) * G_KT + 2 * lane) = x;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(Vt + gv[i], st, (uint32_t)((8 * (4 * w + i)) * G_KT * 8));
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(Ct + gc[i], st, (uint32_t)((NP * G_KT + (8 * (2 * w + i)) * G_KT) * 8));
      }
    };
    auto mma_tile = [&](int stage, int kt_issue, int stage_issue) {
      const double *vs = ring + stage * STG + (wp * 64) * G_KT;
      const double *cs = ring + stage * STG + NP * G_KT + (wc * 64) * G_KT;
#pragma unroll
      for (int kk = 0; kk < G_KT / 4; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          a[x] = cs[af[kk] + x * 16 * G_KT];
          b[x] = vs[af[kk] + x * 16 * G_KT];
        }
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          if ((OPT & 1) && kk == 1 && ci == 2) {  // the loads of tile kt + 2 go out in the middle of the MFMA stream
            __builtin_amdgcn_sched_barrier(0);
            if (kt_issue >= 0) issue_tile(kt_issue, stage_issue);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int pi = 0; pi < 4; ++pi) acc[ci][pi] = mfma_f64(a[ci], b[pi], acc[ci][pi]);
        }
      }
    };
    if (nkt > 0) {
      issue_tile(0, 0);
      if (nkt > 1) {
        issue_tile(1, 1);
        gemm_lds_landed<6>();  // tile 0 is in; tile 1's six loads stay in flight (a register-staged tile 1 is complete: waits longer, still right)
      } else {
        gemm_lds_landed<0>();
      }
    }
    int s0 = 0, s1 = 1, s2 = 2;  // stages of tiles kt, kt + 1, kt + 2
    for (int kt = 0; kt < nkt; ++kt) {
      if (!(OPT & 1) && kt + 2 < nkt) issue_tile(kt + 2, s2);  // stage s2 was read during K-tile kt - 1: every wave is past that barrier
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(s0, kt + 2 < nkt ? kt + 2 : -1, s2);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 2 < nkt) gemm_lds_landed<6>();    // tile kt + 1 has landed everywhere; tile kt + 2 stays in flight
      else gemm_lds_landed<0>();                 // also after the last K-tile: the next unit refills the ring
      const int sx = s0;
      s0 = s1;
      s1 = s2;
      s2 = sx;
    }
    double *o = out + uy * osplit_stride + c0 * NP;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wc * 64 + ci * 16 + k4 + 4 * g;
        if (cl < ncv) {
#pragma unroll
          for (int pi = 0; pi < 4; ++pi) o[(uint32_t)(wp * 64 + pi * 16 + i16) + (uint32_t)(cl * NP)] = acc[ci][pi][g];
        }
      }
  }
}

template <int VEC, bool SK = false>
__global__ __launch_bounds__(512) void k_gemm_tn2(const double *__restrict__ V, int64_t ldv,
                                                  const double *__restrict__ C, int64_t ldc, int64_t rows,
                                                  int64_t ncols, int64_t rps, double *__restrict__ out,
                                                  int64_t osplit_stride, int64_t skq, int rgroups, int pstride) {
  if constexpr (VEC == 2) {
    gemm_tn2_direct<SK>(V, ldv, C, ldc, rows, ncols, rps, out, osplit_stride, skq, rgroups, pstride);
    return;
  }
  constexpr int NP = 256;
  __shared__ __attribute__((aligned(16))) double Vs[2][NP * G_LDK];
  __shared__ __attribute__((aligned(16))) double Cs[2][128 * G_LDK];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int wc = w >> 2, wp = w & 3;
  // PERSISTENT: gridDim.x workgroups (one per CU, the host leaves CUs free for the look-ahead lane / RCCL by launching
  // fewer) loop over the units u = (column tile, row slab), column tile fastest -- the order of the former 2-D grid.
  const int64_t ntiles = (ncols + 127) / 128, nslab_all = (rows + rps - 1) / rps;
  int64_t nslab = nslab_all, slab0 = 0, wgi = (int64_t)blockIdx.x, pbase = 0;  // (row groups: see tn2_sk_group_of)
  if constexpr (SK) {
    if (rgroups > 1) {
      const int x = (int)(blockIdx.x & 7u), g = x % rgroups;
      const tn2_sk_group gr = tn2_sk_group_of(g, rgroups, (int64_t)gridDim.x, nslab_all, ntiles, skq);
      nslab = gr.nsl;
      slab0 = gr.slab0;
      skq = gr.q;
      wgi = (int64_t)(blockIdx.x >> 3) * (8 / rgroups) + x / rgroups;
      pbase = (int64_t)g * pstride;
    }
  }
  const int64_t ufirst = SK ? wgi * skq : (int64_t)blockIdx.x;
  const int64_t ulast = SK ? (ufirst + skq < ntiles * nslab ? ufirst + skq : ntiles * nslab) : ntiles * nslab;
  for (int64_t u = ufirst; u < ulast;) {
  int64_t ux, uy, rbeg, rend;
  if constexpr (SK) {
    ux = u / nslab;
    const int64_t s0 = u - ux * nslab;
    const int64_t s1 = (s0 + (ulast - u) < nslab) ? s0 + (ulast - u) : nslab;
    rbeg = (slab0 + s0) * rps;
    rend = ((slab0 + s1) * rps < rows) ? (slab0 + s1) * rps : rows;
    uy = pbase + wgi - (ux * nslab) / skq;  // this workgroup's piece of tile ux
    u += s1 - s0;
  } else {
    ux = u % ntiles;
    uy = u / ntiles;
    rbeg = uy * rps;
    rend = (rbeg + rps < rows) ? rbeg + rps : rows;
    u += gridDim.x;
  }
  const int64_t c0 = ux * 128;
  const int nkt = (int)((rend - rbeg + G_KT - 1) / G_KT);
  const int ncv = (int)((ncols - c0 < 128) ? ncols - c0 : 128);

  dhqr_d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};

  // staging: V tile = 256 columns x 16 rows = 2048 row pairs (4 per thread), C tile = 128 x 16 = 1024 (2 per thread)
  const double *Vb = V + rbeg;
  const double *Cb = C + rbeg + c0 * ldc;
  uint32_t offv[4], offc[2];
  bool okc[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) offv[i] = (uint32_t)(((t + i * 512) >> 3) * ldv);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = (t + i * 512) >> 3;
    okc[i] = col < ncv;
    offc[i] = (uint32_t)((okc[i] ? col : 0) * ldc);
  }
  double2 sv[4], sc[2];
  auto load_tile = [&](int kt) {  // issue only (see k_gemm_tn)
    const int left = (int)(rend - rbeg) - kt * G_KT;
    const double *Vt = Vb + kt * G_KT;
    const double *Ct = Cb + kt * G_KT;
    const int rp = t & 7;
    if constexpr (VEC == 2) {
      const uint32_t ro = (2 * rp < left) ? 2 * rp : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const double2 *>(Vt + (offv[i] + ro));
#pragma unroll
      for (int i = 0; i < 2; ++i) sc[i] = *reinterpret_cast<const double2 *>(Ct + (offc[i] + ro));
    } else {
      const uint32_t r0o = (2 * rp < left) ? 2 * rp : 0, r1o = (2 * rp + 1 < left) ? 2 * rp + 1 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sv[i].x = Vt[offv[i] + r0o];
        sv[i].y = Vt[offv[i] + r1o];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        sc[i].x = Ct[offc[i] + r0o];
        sc[i].y = Ct[offc[i] + r1o];
      }
    }
  };
  auto store_tile = [&](int buf, int kt) {
    const int left = (int)(rend - rbeg) - kt * G_KT;
    const int rp = t & 7;
    const bool ok0 = 2 * rp < left, ok1 = 2 * rp + 1 < left;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double2 x = sv[i];
      if (!ok0) x.x = 0.0;
      if (!ok1) x.y = 0.0;
      *reinterpret_cast<double2 *>(&Vs[buf][((t + i * 512) >> 3) * G_LDK + 2 * rp]) = x;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double2 y = sc[i];
      if (!ok0) y.x = 0.0;
      if (!ok1) y.y = 0.0;
      if (!okc[i]) y = make_double2(0.0, 0.0);
      *reinterpret_cast<double2 *>(&Cs[buf][((t + i * 512) >> 3) * G_LDK + 2 * rp]) = y;
    }
  };

  auto mma_tile = [&](int buf) {
    const double *cs = &Cs[buf][(wc * 64 + i16) * G_LDK + k4];
    const double *vs = &Vs[buf][(wp * 64 + i16) * G_LDK + k4];
#pragma unroll
    for (int kk = 0; kk < G_KT / 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        a[x] = cs[x * 16 * G_LDK + kk * 4];
        b[x] = vs[x * 16 * G_LDK + kk * 4];
      }
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) acc[ci][pi] = mfma_f64(a[ci], b[pi], acc[ci][pi]);
    }
  };
  if (nkt > 0) {
    load_tile(0);
    store_tile(0, 0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nkt) store_tile(buf ^ 1, kt + 1);
    __syncthreads();
  }

  double *o = out + uy * osplit_stride + c0 * NP;
#pragma unroll
  for (int ci = 0; ci < 4; ++ci)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = wc * 64 + ci * 16 + k4 + 4 * g;
      if (cl < ncv) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) o[(uint32_t)(wp * 64 + pi * 16 + i16) + (uint32_t)(cl * NP)] = acc[ci][pi][g];
      }
    }
  }  // unit loop (every K loop ends with a barrier, so the next unit may overwrite the LDS buffers)
}

// -------------------------------------------------------------------------------------------
// k_gemm_nn_sub:  C[r + c*ldc] -= sum_{p<128} V[r + p*ldv] * W[p + c*ldw]
//   r in [0,rows), c in [0,ncols).  grid = (ceil(rows/128), ceil(ncols/128)).
// The accumulators are initialised with the C tile and the W operand is negated while staging,
// so the MFMA chain itself performs the subtraction.
// INIT0: the accumulators start at zero instead of the C tile (C = -V*W, used for the panel's V = P*M^{-1}).
// stat/epoch: device-side commit predicate of the asynchronous panel pipeline (dhqr_api.hip): the launch is a
// no-op when a panel with index <= epoch failed its verification (stat[0] = index of the first failed panel).
// Phase clock of the TIME instantiation (micro-benchmark only): summed shader cycles of wave 0 per phase.
#ifdef DHQR_BENCH_BUILD
__device__ unsigned long long g_nn_phase[8];
#endif

// TR = rows of the output tile: 128 (the trailing update: each wave 64 x 64) or 64 (four waves side by side, each 64 rows x
// 32 columns).  A workgroup's time is its K loop on one CU (13.7 us for a 128 x 128 x 128 tile); the latency-critical
// products of the panel lane (V = P M^{-1}, the narrow look-ahead update) at a few thousand rows fill a fraction of the
// chip with 128-row tiles, so they run with TR = 64: twice the workgroups, half the time each.
// KW = 512 (k_gemm_nn_quad: FOUR panels = two pairs in one pass over C): reflectors 0..255 come from V, reflectors
// 256..511 from V2 (same leading dimension), whose first `skip2` rows (the second pair starts 256 rows below the
// first) are implicit zeros: row tiles above skip2 run half the K loop and never touch V2 (the host passes
// V2 = first stored row - skip2).  Per flop the C traffic and the tile prologue / epilogue are half those of K = 256.
template <int VEC, int KW, bool INIT0, bool TIME, int TR>
__device__ __forceinline__ void gemm_nn_sub_body(const double *__restrict__ V, int64_t ldv, const double *__restrict__ V2,
                                                 int64_t skip2, const double *__restrict__ W, int64_t ldw,
                                                 double *__restrict__ C, int64_t ldc, int64_t rows, int64_t ncols, int swz,
                                                 const int *__restrict__ stat, int epoch,
                                                 const double *__restrict__ fix_alpha = nullptr) {
  static_assert(TR == 128 || TR == 64, "tile rows");
  constexpr int NCI = TR / 32;                     // 16-column MFMA tiles per wave: 4 (64 columns) or 2 (32 columns)
  constexpr int WCOLS = NCI * 16;                  // columns per wave
  constexpr int LDRV = (TR == 128) ? G_LDR : 66;   // LDS stride of a V tile column (k rows 4 banks apart for ds_read_b128)
  constexpr int NVL = TR / 32;                     // V staging chunks per thread: 16 p x TR/2 row pairs / 256 threads
  constexpr int RPSH = (TR == 128) ? 6 : 5;        // log2(row pairs per tile column)
  // one LDS block, two views: [Vs | Ws] padded (register-staged general path) and, for the interior tiles of the trailing
  // updates, four linear 16 KiB operand images filled by direct loads (below)
  __shared__ __attribute__((aligned(1024))) double lds_raw[2 * G_KT * LDRV + 2 * 128 * G_LDK];
  double(*const Vs)[G_KT * LDRV] = reinterpret_cast<double(*)[G_KT * LDRV]>(lds_raw);
  double(*const Ws)[128 * G_LDK] = reinterpret_cast<double(*)[128 * G_LDK]>(lds_raw + 2 * G_KT * LDRV);
  if (stat != nullptr && stat[0] <= epoch) return;  // uniform: every workgroup of the launch takes the same branch
  long long tph[5] = {0, 0, 0, 0, 0};
  if constexpr (TIME) tph[0] = clock64();
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  const int wr = (TR == 128) ? (w & 1) : 0, wc = (TR == 128) ? (w >> 1) : w;
  int64_t tr = blockIdx.x, tc = blockIdx.y;
  if (TR == 128 && swz) {
    // XCD-aware order (1-D launch): workgroup L runs on XCD L % 8 (observed dispatch rule, speed
    // only).  Each XCD walks its own 8 x 8 blocks of tiles so the V row-tiles and W column-tiles it
    // re-reads (8 + 8 tiles x 128 KiB = 2 MiB) stay in its private 4 MiB L2.
    const int64_t gx = (rows + 127) / 128, gy = (ncols + 127) / 128;
    const int64_t bx = (gx + 7) / 8;
    const int64_t L = blockIdx.x;
    const int64_t xcd = L & 7, sq = L >> 3;
    const int64_t blk = (sq >> 6) * 8 + xcd, idx = sq & 63;
    tr = (blk % bx) * 8 + (idx & 7);
    tc = (blk / bx) * 8 + (idx >> 3);
    if (tr >= gx || tc >= gy) return;
  }
  const int64_t r0 = tr * TR;
  const int64_t c0 = tc * 128;
  const int nrv = (int)((rows - r0 < TR) ? rows - r0 : TR);      // valid rows in this tile
  const int ncv = (int)((ncols - c0 < 128) ? ncols - c0 : 128);  // valid columns

  const double *Vb = V + r0;
  const double *Vb2 = (KW == 512) ? V2 + r0 : V;  // only dereferenced by tiles at or below skip2
  const double *Wb = W + c0 * ldw;
  double *Cb = C + r0 + c0 * ldc;
  const bool below2 = (KW != 512) || r0 >= skip2;  // uniform: the tile sees the second pair's reflectors

  // staging offsets (32-bit, from uniform bases); invalid rows/columns are clamped to element 0
  uint32_t offv[4], offw[4];
  bool okv0[4], okv1[4], okw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = t + i * 256;
    const int p = (q >> RPSH) & (G_KT - 1), rp = q & ((1 << RPSH) - 1);  // V tile: 16 p-columns x TR rows (i < NVL)
    okv0[i] = 2 * rp < nrv;
    okv1[i] = 2 * rp + 1 < nrv;
    offv[i] = (uint32_t)(p * ldv) + (okv0[i] ? 2 * rp : 0);
    const int col = q >> 3, pp = q & 7;  // W tile: 128 columns x 16 p (128 B per column)
    okw[i] = col < ncv;
    offw[i] = (uint32_t)((okw[i] ? col : 0) * ldw) + 2 * pp;
  }
  double2 sv[4], sw[4];
  auto load_tile = [&](int kt) {  // issue only; masking / negation happen in store_tile
    const double *Vt = (KW == 512 && kt >= KW / (2 * G_KT)) ? Vb2 + (int64_t)(kt - KW / (2 * G_KT)) * G_KT * ldv
                                                            : Vb + (int64_t)kt * G_KT * ldv;
    const double *Wt = Wb + kt * G_KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (VEC == 2) {
        if (i < NVL) sv[i] = *reinterpret_cast<const double2 *>(Vt + offv[i]);  // rows even: pair all-or-nothing
        sw[i] = *reinterpret_cast<const double2 *>(Wt + offw[i]);
      } else {
        if (i < NVL) {
          sv[i].x = Vt[offv[i]];
          sv[i].y = Vt[offv[i] + (okv1[i] ? 1 : 0)];
        }
        sw[i].x = Wt[offw[i]];
        sw[i].y = Wt[offw[i] + 1];
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = t + i * 256;
      double2 x = sv[i], y = sw[i];
      if (!okv0[i]) x.x = 0.0;
      if (!okv1[i]) x.y = 0.0;
      if (!okw[i]) y = make_double2(0.0, 0.0);
      if (i < NVL) *reinterpret_cast<double2 *>(&Vs[buf][(q >> RPSH) * LDRV + 2 * (q & ((1 << RPSH) - 1))]) = x;
      *reinterpret_cast<double2 *>(&Ws[buf][(q >> 3) * G_LDK + 2 * (q & 7)]) = make_double2(-y.x, -y.y);
    }
  };

  // interior tiles of the trailing updates take the direct-to-LDS program below; everything else is register staged
  constexpr bool STREAM = !INIT0 && VEC == 2 && (KW == 512 || KW == 256 || KW == 128) && TR == 128;
  const bool full = (VEC == 2) && nrv == TR && ncv == 128;
  const bool direct = STREAM && full && below2;  // uniform
  if (!direct) load_tile(0);

  // Accumulator map.  The 16 x 16 MFMA tile ri of a wave does NOT hold 16 consecutive rows: lane i16 of tile ri owns
  // row 4*i16 + ri of the wave's 64 rows (the V fragments are read with the same map), so the four tiles together
  // give every lane FOUR CONSECUTIVE ROWS per column:  lane (i16,k4), register g of tile (ci,ri) holds
  //     C[r0 + wr*64 + 4*i16 + ri][c0 + wc*64 + ci*16 + k4 + 4g].
  // Interior tiles move C with 16-byte accesses (32 B per lane and column, 512 contiguous bytes per 16 lanes).
  constexpr int NKT = KW / G_KT;
  dhqr_d4 acc[NCI][4];
  auto mma_tile = [&](int buf) {
    const double *ws = &Ws[buf][(wc * WCOLS + i16) * G_LDK + k4];
    const double *vs = &Vs[buf][k4 * LDRV + wr * 64 + 4 * i16];  // rows 4*i16 .. 4*i16+3: the four b fragments
#pragma unroll
    for (int kk = 0; kk < G_KT / 4; ++kk) {
      double a[NCI], b[4];
#pragma unroll
      for (int x = 0; x < NCI; ++x) a[x] = ws[x * 16 * G_LDK + kk * 4];
      // two ds_read_b128; stride 130 (66) doubles puts the four k rows of a read 4 banks apart: conflict free
      const double2 b01 = *reinterpret_cast<const double2 *>(vs + kk * 4 * LDRV);
      const double2 b23 = *reinterpret_cast<const double2 *>(vs + kk * 4 * LDRV + 2);
      b[0] = b01.x;
      b[1] = b01.y;
      b[2] = b23.x;
      b[3] = b23.y;
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = mfma_f64(a[ci], b[ri], acc[ci][ri]);
    }
  };
  // the lane's 4 NCI (column, 4-row) units of an interior tile: unit n = 4*ci + g is column wc*WCOLS + k4 + 4n, so the
  // units are one pointer walking with the uniform stride 4*ldc
  double *const cunit0 = Cb + ((uint32_t)((wc * WCOLS + k4) * ldc) + (uint32_t)(wr * 64 + 4 * i16));
  const int64_t cstep = 4 * ldc;
  auto store_full = [&]() {  // interior tiles: 16 bytes per store, no masks
    double *cp = cunit0;
#pragma unroll
    for (int n = 0; n < 4 * NCI; ++n) {
      *reinterpret_cast<double2 *>(cp) = make_double2(acc[n >> 2][0][n & 3], acc[n >> 2][1][n & 3]);
      *reinterpret_cast<double2 *>(cp + 2) = make_double2(acc[n >> 2][2][n & 3], acc[n >> 2][3][n & 3]);
      cp += cstep;
    }
  };
  auto time_end = [&]() {
    if constexpr (TIME) {
      __builtin_amdgcn_sched_barrier(0);
      tph[3] = clock64();
      __builtin_amdgcn_s_waitcnt(0x0F70);  // the stores have left the wave
      tph[4] = clock64();
#ifdef DHQR_BENCH_BUILD
      if (threadIdx.x == 0) {
        for (int q = 0; q < 4; ++q) atomicAdd(&g_nn_phase[q], (unsigned long long)(tph[q + 1] - tph[q]));
        atomicAdd(&g_nn_phase[4], 1ull);
      }
#endif
    }
  };

  // ---- interior tiles of the trailing updates: operands by direct loads, C streams in DURING the K loop ---------------
  // acc = V W - C is a sum: the accumulators start at zero and every K-tile subtracts 16 / NKT of the lane's sixteen
  // (column, 4-row) units of C, requested at the top of the K-tile and subtracted below its 64 MFMAs; the stores write
  // -acc (the same numbers, bit for bit, as C - V W accumulated with a negated W).  The tile has no C prologue.  Why: with
  // the C tile fetched up front the whole chip falls into a convoy -- every workgroup waits for its 128 KB while the HBM
  // serves all 512 of them at once (phase clock: 28k of a tile's 167k cycles with idle matrix pipes), and waiting on a
  // saturated memory re-forms the convoy after any perturbation (random start phases changed nothing).  Streaming
  // spreads the same reads evenly over the K loops.
  // Operands (r5): global -> LDS directly (glds16), two 32 KiB stages, the next K-tile's 8 loads per wave issued at the
  // top of the current one; one wait + barrier per K-tile.  LDS images (linear per wave instruction; the swizzle is in
  // the per-lane global address, conflict-free fragment reads without padding):
  //   V tile: column p (16) = 64 chunks of 16 B (rows 2 ch, 2 ch + 1); position pos holds chunk pos ^ ((pos >> 4) & 1)
  //   W tile: column c (128) = 8 chunks of 16 B (k = 2 j, 2 j + 1);     slot s holds chunk s ^ ((c >> 1) & 7)
  // Per workgroup-timeline (tools/gemm_lab.hip, gemm_lab_timeline.py, 16384^2): tile = 288k cycles of which the K loop
  // 272k (floor with two workgroups per CU: 262k), prologue 9k, stores 7k, 7k until the successor starts.
  if constexpr (STREAM) if (direct) {  // uniform branch
    typedef double dhqr_d2 __attribute__((ext_vector_type(2)));
    double *const Vg = lds_raw, *const Wg = lds_raw + 2 * G_KT * 128;  // [2][16 * 128] each
    uint32_t gv[4], gw[4];  // this wave's 4 V columns (p = 4 w + i) and 4 W column groups (columns 8 (4 w + i) .. + 7)
    {
      const int ch = lane ^ ((lane >> 4) & 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gv[i] = (uint32_t)((4 * w + i) * ldv) + 2 * ch;
        const int col = 8 * (4 * w + i) + (lane >> 3), j = (lane & 7) ^ ((col >> 1) & 7);
        gw[i] = (uint32_t)(col * ldw) + 2 * j;
      }
    }
    auto issue_tile = [&](int kt) {
      const double *Vt = (KW == 512 && kt >= KW / (2 * G_KT)) ? Vb2 + (int64_t)(kt - KW / (2 * G_KT)) * G_KT * ldv
                                                              : Vb + (int64_t)kt * G_KT * ldv;
      const double *Wt = Wb + kt * G_KT;
      const uint32_t bo = (uint32_t)(((kt & 1) * G_KT * 128 + (4 * w) * 128) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        glds16(Vt + gv[i], Vg, bo + (uint32_t)i * 1024u);
        glds16(Wt + gw[i], Wg, bo + (uint32_t)i * 1024u);
      }
    };
    int aw[4];  // W fragment of k-step kk: column wc*64 + i16 (+16 x), chunk (2 kk + (k4 >> 1)) ^ (i16 >> 1), half k4 & 1
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) aw[kk] = (wc * 64 + i16) * G_KT + (((2 * kk + (k4 >> 1)) ^ (i16 >> 1)) * 2) + (k4 & 1);
    const int ch0 = wr * 32 + 2 * i16, fl = (i16 >> 3) & 1;  // rows 4 i16 .. + 3 of the wave: chunks ch0, ch0 + 1
    const int av0 = k4 * 128 + ((ch0 ^ fl) * 2), av1 = k4 * 128 + (((ch0 ^ fl) ^ 1) * 2);
    // kt_next >= 0: the loads of that K-tile are issued from the MIDDLE of this tile's MFMA stream (behind 24 of the 64):
    // a direct load costs ~60 cycles of issue (s_mov m0, address, the load); at the top of the K-tile, where the wave has
    // no MFMA in flight, 8 of them stood between the barrier and the first MFMA (67.98 -> 70.32 TFLOP/s, same bits)
    auto mma_direct = [&](int buf, int kt_next) {
      const double *ws = Wg + buf * (G_KT * 128);
      const double *vs = Vg + buf * (G_KT * 128);
#pragma unroll
      for (int kk = 0; kk < G_KT / 4; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) a[x] = ws[aw[kk] + x * 16 * G_KT];
        const double2 b01 = *reinterpret_cast<const double2 *>(vs + av0 + kk * 4 * 128);
        const double2 b23 = *reinterpret_cast<const double2 *>(vs + av1 + kk * 4 * 128);
        b[0] = b01.x;
        b[1] = b01.y;
        b[2] = b23.x;
        b[3] = b23.y;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          if (kk == 1 && ci == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (kt_next >= 0) issue_tile(kt_next);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = mfma_f64(a[ci], b[ri], acc[ci][ri]);
        }
      }
    };
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
    issue_tile(0);
    gemm_lds_landed<0>();
    if constexpr (TIME) tph[1] = clock64();
    constexpr int KTPU = NKT > 16 ? NKT / 16 : 1;     // K-tiles per unit (KW = 512: a unit every second K-tile)
    constexpr int UPT = NKT > 16 ? 1 : 16 / NKT;      // units per K-tile that carries units
    const double *cin = cunit0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const bool carry = (kt % KTPU) == 0;  // compile-time after unrolling
      double2 cu[UPT][2];
      if (carry) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {  // non-temporal: C passes through once and should not push the operands out of L2
          const dhqr_d2 x0 = __builtin_nontemporal_load(reinterpret_cast<const dhqr_d2 *>(cin));
          const dhqr_d2 x1 = __builtin_nontemporal_load(reinterpret_cast<const dhqr_d2 *>(cin + 2));
          cu[u][0] = make_double2(x0[0], x0[1]);
          cu[u][1] = make_double2(x1[0], x1[1]);
          cin += cstep;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_direct(kt & 1, kt + 1 < NKT ? kt + 1 : -1);
      __builtin_amdgcn_sched_barrier(0);
      if (carry) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
          const int ci = ((kt / KTPU) * UPT + u) >> 2, g = ((kt / KTPU) * UPT + u) & 3;
          acc[ci][0][g] -= cu[u][0].x;
          acc[ci][1][g] -= cu[u][0].y;
          acc[ci][2][g] -= cu[u][1].x;
          acc[ci][3][g] -= cu[u][1].y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the C units are consumed (their vmcnt wait sits) BEFORE the next loads go out
      if (kt + 1 < NKT) gemm_lds_landed<0>();
    }
    if constexpr (TIME) {
      __builtin_amdgcn_sched_barrier(0);
      tph[2] = clock64();
    }
    {
      double *cp = cunit0;
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        dhqr_d2 y0, y1;  // (non-temporal like the loads: HBM-side reads of a K = 256 launch 15.0 -> 11.4 GB, K = 512 20.0 -> 18.9 GB
        y0[0] = -acc[n >> 2][0][n & 3];  //  at 32768 x 28672, same bits, K = 256 1 % faster: profiles/r05_ab_gemm_structure.txt)
        y0[1] = -acc[n >> 2][1][n & 3];
        y1[0] = -acc[n >> 2][2][n & 3];
        y1[1] = -acc[n >> 2][3][n & 3];
        __builtin_nontemporal_store(y0, reinterpret_cast<dhqr_d2 *>(cp));
        __builtin_nontemporal_store(y1, reinterpret_cast<dhqr_d2 *>(cp + 2));
        cp += cstep;
      }
    }
    time_end();
    return;
  }

  // ---- general path (edge tiles, unaligned operands, INIT0, narrow reflector blocks): C first -------------------------
  if (INIT0) {
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
  } else if (full) {
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double *cp = cunit0 + (4 * ci + g) * cstep;
        const double2 x0 = *reinterpret_cast<const double2 *>(cp), x1 = *reinterpret_cast<const double2 *>(cp + 2);
        acc[ci][0][g] = x0.x;
        acc[ci][1][g] = x0.y;
        acc[ci][2][g] = x1.x;
        acc[ci][3][g] = x1.y;
      }
  } else {  // clamped addresses, all loads issued before the first mask is applied
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wc * WCOLS + ci * 16 + k4 + 4 * g;
        const uint32_t co = (uint32_t)((cl < ncv ? cl : 0) * ldc);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int rl = wr * 64 + 4 * i16 + ri;
          acc[ci][ri][g] = Cb[co + (uint32_t)(rl < nrv ? rl : 0)];
        }
      }
    if constexpr (VEC == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bool cok = wc * WCOLS + ci * 16 + k4 + 4 * g < ncv;
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
          if (!(cok && wr * 64 + 4 * i16 + ri < nrv)) acc[ci][ri][g] = 0.0;
      }
  }

  store_tile(0);
  __syncthreads();
  if constexpr (TIME) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the C tile has arrived
    tph[1] = clock64();
  }
  const int nkt_run = below2 ? NKT : NKT / 2;  // tiles above skip2: the second pair's reflectors are zero there
#pragma unroll 1
  for (int kt = 0; kt < nkt_run - 1; ++kt) {
    const int buf = kt & 1;
    load_tile(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(buf);
    __builtin_amdgcn_sched_barrier(0);
    store_tile(buf ^ 1);
    __syncthreads();
  }
  mma_tile((nkt_run - 1) & 1);
  if constexpr (TIME) {
    __builtin_amdgcn_sched_barrier(0);
    tph[2] = clock64();
  }
  if constexpr (INIT0) {
    // The panel's V = tril((P - alpha E) M^{-1}) (dhqr_recon.h) in the product's own epilogue: the launch computed P M^{-1}
    // (W = -M^{-1}); on the top 128 rows subtract alpha_i M^{-1}[i][j] on and below the diagonal and clear what lies above
    // it -- one launch (k_recon_fix) less on the panel chain.
    if (fix_alpha != nullptr && r0 < 128) {
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = (int)c0 + wc * WCOLS + ci * 16 + k4 + 4 * g;
#pragma unroll
          for (int ri = 0; ri < 4; ++ri) {
            const int row = (int)r0 + wr * 64 + 4 * i16 + ri;
            if (row < 128 && col < 128)
              acc[ci][ri][g] = (row >= col) ? acc[ci][ri][g] + fix_alpha[row] * W[row + col * ldw] : 0.0;
          }
        }
    }
  }
  if (full) {
    store_full();
  } else {
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wc * WCOLS + ci * 16 + k4 + 4 * g;
        if (cl < ncv) {
          const uint32_t co = (uint32_t)(cl * ldc);
#pragma unroll
          for (int ri = 0; ri < 4; ++ri) {
            const int rl = wr * 64 + 4 * i16 + ri;
            if (rl < nrv) Cb[co + (uint32_t)rl] = acc[ci][ri][g];
          }
        }
      }
  }
  time_end();
}

template <int VEC, int KW, bool INIT0 = false, bool TIME = false, int TR = 128>
__global__ __launch_bounds__(256, 2) void k_gemm_nn_sub(const double *__restrict__ V, int64_t ldv,
                                                        const double *__restrict__ W, int64_t ldw,
                                                        double *__restrict__ C, int64_t ldc,
                                                        int64_t rows, int64_t ncols, int swz,
                                                        const int *__restrict__ stat, int epoch) {
  static_assert(KW != 512, "K = 512 takes two reflector operands: k_gemm_nn_quad");
  gemm_nn_sub_body<VEC, KW, INIT0, TIME, TR>(V, ldv, nullptr, 0, W, ldw, C, ldc, rows, ncols, swz, stat, epoch);
}

// out = -V W with the reflector fix of the panel chain in the epilogue (gemm_nn_sub_body, INIT0): V = tril((P - alpha E) M^{-1})
template <int VEC, int TR>
__global__ __launch_bounds__(256, 2) void k_gemm_nn_vfix(const double *__restrict__ V, int64_t ldv, const double *__restrict__ W,
                                                         int64_t ldw, double *__restrict__ C, int64_t ldc, int64_t rows,
                                                         int64_t ncols, const double *__restrict__ fix_alpha) {
  gemm_nn_sub_body<VEC, 128, true, false, TR>(V, ldv, nullptr, 0, W, ldw, C, ldc, rows, ncols, 0, nullptr, 0, fix_alpha);
}

// The four-panel update C -= [V | V2] W (W: 512 x ncols, rows 0..255 for V, 256..511 for V2), see gemm_nn_sub_body.
template <int VEC, int TR = 128>
__global__ __launch_bounds__(256, 2) void k_gemm_nn_quad(const double *__restrict__ V, const double *__restrict__ V2,
                                                         int64_t ldv, int64_t skip2, const double *__restrict__ W,
                                                         int64_t ldw, double *__restrict__ C, int64_t ldc, int64_t rows,
                                                         int64_t ncols, int swz, const int *__restrict__ stat, int epoch) {
  gemm_nn_sub_body<VEC, 512, false, false, TR>(V, ldv, V2, skip2, W, ldw, C, ldc, rows, ncols, swz, stat, epoch);
}

// out[e] = sum_{s<nsplit} in[s*stride + e], e < count  (split-K reduction, deterministic order).
// A block of 256 threads owns 64 consecutive elements; its four 64-thread groups each sum a quarter
// of the splits (interleaved) and the quarters are combined through LDS in a fixed order, so a
// 128 x 128 Gram matrix with 256 partials is reduced by 256 blocks instead of 64.
// The reduction of a stream-K k_gemm_tn2 launch: element e of Y (256 x ncols, ld 256) belongs to column tile
// t = e / (256 * 128) and is the sum of that tile's pieces (slots 0 .. npieces(t) - 1, see k_gemm_tn2).
// rgroups > 1: the pieces of row group g sit in slots g * pstride ..; G = workgroups of the k_gemm_tn2 launch.
__global__ __launch_bounds__(256) void k_reduce_pieces(const double *__restrict__ in, int64_t nslab, int64_t skq,
                                                       int64_t stride, int64_t count, double *__restrict__ out,
                                                       int rgroups, int pstride, int64_t G, int64_t ntiles) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const int64_t tile = e / (256 * 128);
  double s = 0.0;
  for (int g = 0; g < (rgroups > 1 ? rgroups : 1); ++g) {
    const tn2_sk_group gr = tn2_sk_group_of(g, rgroups, G, nslab, ntiles, skq);
    if (gr.nsl <= 0) continue;
    const int np = (int)((((tile + 1) * gr.nsl - 1) / gr.q) - ((tile * gr.nsl) / gr.q) + 1);
    const double *ing = in + (int64_t)g * pstride * stride;
    double sg = ing[e];
    for (int q = 1; q < np; ++q) sg += ing[(int64_t)q * stride + e];
    s = (g == 0) ? sg : s + sg;
  }
  out[e] = s;
}

__global__ __launch_bounds__(256) void k_reduce_splits(const double *__restrict__ in, int nsplit,
                                                       int64_t stride, int64_t count,
                                                       double *__restrict__ out) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < count) {
    // eight loads in flight per lane (the loop was latency-bound: one L2 / fabric round trip per term of a dependent sum:
    // 234 partials of a 128 x 128 Gram matrix in 17 us = 1.8 TB/s); the order of the additions is fixed
    int q = grp;
    for (; q + 28 < nsplit; q += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = in[(int64_t)(q + 4 * u) * stride + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; q < nsplit; q += 4) s += in[(int64_t)q * stride + e];
  }
  part[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && e < count) out[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// The T products of a NARROW update (one or two column tiles: the look-ahead lane) in one launch, Y = the reduced
// split-K sums (ldy x ncols; ldy = 128, or 256 = [y_a; y_b] for a pair of panels):
//   w_a = Top_a' y_a;   pair:  y_b -= Sba w_a,  w_b = Top_b' y_b;      W2 <- [w_a; w_b]
// with TaT / TbT = the TRANSPOSES of Top_a / Top_b (column-major, so a wave reads 64 consecutive doubles per term).
// Through the GEMM kernels these were one to three 128 x 128 x ncols products of ONE workgroup per 128 columns -- 23 us
// each (a workgroup's K loop on one CU), up to 70 us of a pair's critical chain; here 4 columns per workgroup (64
// workgroups for 256 columns), a few microseconds.  Deterministic: fixed summation order.
template <bool PAIR>
__global__ __launch_bounds__(256) void k_tw_fused(const double *__restrict__ Y, int64_t ncols, const double *__restrict__ TaT,
                                                  const double *__restrict__ TbT, const double *__restrict__ Sba,
                                                  double *__restrict__ W2, int64_t ldw) {
  constexpr int LDY = PAIR ? 256 : 128, NC = 4;  // ldw: leading dimension of W2 (LDY, or 512 inside a four-panel update)
  __shared__ double y[NC][LDY], wa[NC][128];
  const int t = threadIdx.x, p = t & 127, ch = t >> 7;  // this thread: output row p of columns 2 ch, 2 ch + 1
  const int64_t c0 = (int64_t)blockIdx.x * NC;
  for (int e = t; e < NC * LDY; e += 256) {  // the 4-column slab of Y
    const int c = e / LDY, r = e - c * LDY;
    y[c][r] = (c0 + c < ncols) ? Y[r + (c0 + c) * LDY] : 0.0;
  }
  __syncthreads();
  auto tprod = [&](const double *__restrict__ TT, const double (*v)[LDY], int off, double &o0, double &o1) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int r = 0; r < 128; ++r) {
      const double tv = TT[p + r * 128];
      s0 = fma(tv, v[2 * ch][off + r], s0);
      s1 = fma(tv, v[2 * ch + 1][off + r], s1);
    }
    o0 = s0;
    o1 = s1;
  };
  double a0, a1;
  tprod(TaT, y, 0, a0, a1);  // w_a = Top_a' y_a
  if (c0 + 2 * ch < ncols) W2[p + (c0 + 2 * ch) * ldw] = a0;
  if (c0 + 2 * ch + 1 < ncols) W2[p + (c0 + 2 * ch + 1) * ldw] = a1;
  if constexpr (PAIR) {
    wa[2 * ch][p] = a0;
    wa[2 * ch + 1][p] = a1;
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;  // y_b -= Sba w_a  (Sba = V_b'V_a, column-major)
#pragma unroll 8
    for (int q = 0; q < 128; ++q) {
      const double sv = Sba[p + q * 128];
      s0 = fma(sv, wa[2 * ch][q], s0);
      s1 = fma(sv, wa[2 * ch + 1][q], s1);
    }
    __syncthreads();  // nobody reads y_a any more; y_b rows are private to their (p, ch) owners until the barrier below
    y[2 * ch][128 + p] -= s0;
    y[2 * ch + 1][128 + p] -= s1;
    __syncthreads();
    double b0, b1;
    tprod(TbT, y, 128, b0, b1);  // w_b = Top_b' y_b
    if (c0 + 2 * ch < ncols) W2[128 + p + (c0 + 2 * ch) * ldw] = b0;
    if (c0 + 2 * ch + 1 < ncols) W2[128 + p + (c0 + 2 * ch + 1) * ldw] = b1;
  }
}

// Compact-WY algebra used by the T kernel (k_build_t, dhqr_recon.h): for reflectors
// H_j = I - v_j v_j' (tau_j == 1 because ||v_j||^2 = 2), H_1...H_nb = I - V T V' with
//   T^{-1} = I + striu(V'V),  equivalently  T[0:j, j] = -T[0:j,0:j] (V'V)[0:j, j], T[j][j] = 1.
// This holds for ANY vectors v_j, so the reference's zero-pivot reflectors (||v||^2 != 2, src:8)
// are reproduced exactly.  The trailing update is A <- A - V (T' (V' A)).

