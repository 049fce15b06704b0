// dhqr_api.hip -- host side of libdhqr.so: context, workspaces, panel/trailing-update drivers and
// the extern "C" entry points declared in include/dhqr.h.  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dhqr.h"
#include "dhqr_common.h"
#include "dhqr_complex.h"
#include "dhqr_gemm.h"
#include "dhqr_panel.h"
#include "dhqr_rank1.h"
#include "dhqr_recon.h"
#include "dhqr_solve.h"

static thread_local char g_err[512] = "";
static int32_t set_err(int32_t code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHECK(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return set_err(DHQR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),        \
                     __FILE__, __LINE__);                                                     \
  } while (0)
#define CHECK(expr)                  \
  do {                               \
    int32_t rc_ = (expr);            \
    if (rc_ != DHQR_OK) return rc_;  \
  } while (0)
#define LAUNCHCHECK() HIPCHECK(hipGetLastError())

enum { CAT_PANEL = 0, CAT_TBUILD, CAT_VTA, CAT_TW, CAT_AVW, CAT_RANK1, CAT_SOLVE, CAT_N };

struct Buf {
  double *p = nullptr;
  size_t cap = 0;  // doubles
};

struct dhqr_ctx {
  int device = 0;
  hipStream_t own = nullptr, stream = nullptr;
  bool profiling = false;
  hipStream_t hi = nullptr;      // high-priority stream: panel factorisation under look-ahead
  int swizzle = 1;               // XCD-aware tile order in k_gemm_nn_sub (+1.5 % at 32768^2; DHQR_SWIZZLE=0 disables)
  struct WS { Buf w1, w1r, w1r2, w2; } ws[2];  // [0] wide trailing update, [1] panel / narrow updates
  int cur_ws = 0;
  bool lookahead = true;
  hipEvent_t ev_panel[4] = {}, ev_wide[4] = {};
  Buf vbuf, vt, vt2, vts, spart, sfull, scratch, pbuf;
  Buf pairv[2], pairt;           // two-panel (K = 256) update: [V_a V_b] buffers, T_a/T_b/S_ba side storage
  int pair = 1;                  // 1: wide updates apply two panels per pass (DHQR_PAIR=0 disables)
  int64_t pair_min_n = 20480;    // below this the longer look-ahead lane of the pair driver costs more than it saves
  int panel_impl = 3;  // 3: R-first (CholeskyQR2 + reconstruction, dhqr_recon.h) with fallback to 2;
                       // 2: row-split sub-panel kernels (dhqr_panel.h); 1: one workgroup per column
  int *hflag = nullptr;  // pinned host copy of the fast path's verification flags
  Buf rbuf;              // R1, -R1^{-1}, R, Rref, -M^{-1}, alpha_tmp, flags
  int64_t n_fast = 0, n_fallback = 0;
  int cholqr_passes = 1;  // Gram/Cholesky passes of the fast path (2 = CholeskyQR2)
  double recon_tol = 2e-12;  // accepted deviation of ||v_j||^2 from 2 before falling back
  int ib = DHQR_IB;
  int smallk = 3;  // generation of the single-workgroup panel kernels (DHQR_SMALLK): 3 default, 4 = one barrier per
                   // step, 5 = 4 + blocked triangular inverses (five barriers instead of 128)
  // profiling
  struct Ev { hipEvent_t a, b; int cat; };
  std::vector<Ev> evs;
  size_t ev_used = 0;
  dhqr_stats st;
};

static int32_t ensure(dhqr_ctx *c, Buf &b, size_t need) {
  if (need <= b.cap) return DHQR_OK;
  if (b.p) {
    HIPCHECK(hipDeviceSynchronize());  // another stream of this ctx may still use the old buffer
    HIPCHECK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  need = (need + 1023) & ~(size_t)1023;
  hipError_t e = hipMalloc((void **)&b.p, need * sizeof(double));
  if (e != hipSuccess)
    return set_err(DHQR_ENOMEM, "hipMalloc(%zu bytes) failed: %s", need * sizeof(double),
                   hipGetErrorString(e));
  b.cap = need;
  return DHQR_OK;
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---- profiling: one hipEvent pair per timed launch group, resolved in dhqr_get_stats ---------
static int32_t prof_begin(dhqr_ctx *c, int cat) {
  if (!c->profiling) return DHQR_OK;
  if (c->ev_used == c->evs.size()) {
    dhqr_ctx::Ev e;
    HIPCHECK(hipEventCreate(&e.a));
    HIPCHECK(hipEventCreate(&e.b));
    e.cat = cat;
    c->evs.push_back(e);
  }
  c->evs[c->ev_used].cat = cat;
  HIPCHECK(hipEventRecord(c->evs[c->ev_used].a, c->stream));
  return DHQR_OK;
}
static int32_t prof_end(dhqr_ctx *c) {
  if (!c->profiling) return DHQR_OK;
  HIPCHECK(hipEventRecord(c->evs[c->ev_used].b, c->stream));
  c->ev_used++;
  return DHQR_OK;
}
static int32_t prof_resolve(dhqr_ctx *c) {
  HIPCHECK(hipStreamSynchronize(c->stream));
  double *ms[CAT_N] = {&c->st.ms_panel, &c->st.ms_tbuild, &c->st.ms_gemm_vta, &c->st.ms_gemm_tw,
                       &c->st.ms_gemm_avw, &c->st.ms_rank1, &c->st.ms_solve};
  int64_t *cnt[CAT_N] = {&c->st.n_panel, &c->st.n_tbuild, &c->st.n_gemm_vta, &c->st.n_gemm_tw,
                         &c->st.n_gemm_avw, &c->st.n_rank1, &c->st.n_solve};
  for (size_t i = 0; i < c->ev_used; ++i) {
    float t = 0.f;
    HIPCHECK(hipEventElapsedTime(&t, c->evs[i].a, c->evs[i].b));
    *ms[c->evs[i].cat] += (double)t;
    *cnt[c->evs[i].cat] += 1;
  }
  c->ev_used = 0;
  return DHQR_OK;
}

// ---- unblocked factorisation of the columns of a rows x ncols block (src:122-148,198-213) ----
// P's row 0 is the diagonal row of column 0.  One launch per column (k_rank1_*), the workgroup
// owning column j+1 builds the next reflector in the same launch.
template <int VEC>
static void launch_rank1(dhqr_ctx *c, double *P, int64_t ldp, int64_t rows, int64_t j, int64_t nupd,
                         const double *vcur, double *vnext, double *alpha) {
  const int64_t r0 = (VEC == 2) ? (j & ~(int64_t)1) : j;
  const int64_t cov = rows - r0;
  dim3 grid((unsigned)nupd);
#define DHQR_R1(T_, E_)                                                                          \
  hipLaunchKernelGGL((k_rank1_fused<T_, E_, VEC>), grid, dim3(T_), 0, c->stream, P, ldp, rows, j, \
                     vcur, vnext, alpha)
  if (cov <= 256 * 2) DHQR_R1(256, 2);
  else if (cov <= 256 * 4) DHQR_R1(256, 4);
  else if (cov <= 256 * 8) DHQR_R1(256, 8);
  else if (cov <= 512 * 8) DHQR_R1(512, 8);
  else if (cov <= 1024 * 8) DHQR_R1(1024, 8);
  else
    hipLaunchKernelGGL((k_rank1_generic<1024, VEC>), grid, dim3(1024), 0, c->stream, P, ldp, rows,
                       j, vcur, vnext, alpha);
#undef DHQR_R1
}

static int32_t factor_unblocked_cols(dhqr_ctx *c, double *P, int64_t rows, int64_t ncols,
                                     int64_t ldp, double *alpha, int cat) {
  const size_t vlen = (size_t)((rows + 17) & ~(int64_t)15);
  CHECK(ensure(c, c->vbuf, 2 * vlen));
  double *vb[2] = {c->vbuf.p, c->vbuf.p + vlen};
  const bool vec = (ldp % 2 == 0) && (rows % 2 == 0) && aligned16(P);
  CHECK(prof_begin(c, cat));
  hipLaunchKernelGGL((k_reflector<1024>), dim3(1), dim3(1024), 0, c->stream, P, rows, (int64_t)0,
                     vb[0], alpha);
  CHECK(prof_end(c));
  for (int64_t j = 0; j + 1 < ncols; ++j) {
    const int64_t nupd = ncols - (j + 1);
    CHECK(prof_begin(c, cat));
    if (vec) launch_rank1<2>(c, P, ldp, rows, j, nupd, vb[j & 1], vb[(j + 1) & 1], alpha);
    else launch_rank1<1>(c, P, ldp, rows, j, nupd, vb[j & 1], vb[(j + 1) & 1], alpha);
    CHECK(prof_end(c));
    if (c->profiling) {
      const double by = 16.0 * (double)(rows - j) * (double)nupd;
      if (cat == CAT_RANK1) c->st.bytes_rank1 += by;
      else c->st.bytes_panel += by;
    }
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

// ---- the three single-workgroup dense 128 x 128 kernels of the panel chain (dhqr_recon.h), by generation
static inline void launch_chol_inv(dhqr_ctx *c, const double *G, const double *Rprev, double *Rout, double *negX,
                                   int *flag) {
  if (c->smallk >= 4)
    hipLaunchKernelGGL(k_chol_inv4, dim3(1), dim3(1024), 0, c->stream, G, Rprev, Rout, negX, flag);
  else
    hipLaunchKernelGGL(k_chol_inv, dim3(1), dim3(1024), 0, c->stream, G, Rprev, Rout, negX, flag);
}
static inline void launch_recon_top(dhqr_ctx *c, const double *P, int64_t ldp, const double *R, double *alpha,
                                    double *Rref, double *negMinv) {
  if (c->smallk == 5)
    hipLaunchKernelGGL(k_recon_top5, dim3(1), dim3(1024), 0, c->stream, P, ldp, R, alpha, Rref, negMinv);
  else if (c->smallk == 4)
    hipLaunchKernelGGL(k_recon_top4, dim3(1), dim3(1024), 0, c->stream, P, ldp, R, alpha, Rref, negMinv);
  else
    hipLaunchKernelGGL(k_recon_top, dim3(1), dim3(1024), 0, c->stream, P, ldp, R, alpha, Rref, negMinv);
}
static inline void launch_build_t(dhqr_ctx *c, const double *S, int ncols, double *T, double *Tt) {
  if (c->smallk == 5) hipLaunchKernelGGL(k_build_t5, dim3(1), dim3(1024), 0, c->stream, S, ncols, T, Tt);
  else if (c->smallk == 4) hipLaunchKernelGGL(k_build_t4, dim3(1), dim3(1024), 0, c->stream, S, ncols, T, Tt);
  else hipLaunchKernelGGL(k_build_t3, dim3(1), dim3(1024), 0, c->stream, S, ncols, T, Tt);
}

// ---- packed panel buffer helpers ------------------------------------------------------------
static inline int64_t panel_ldv(int64_t rows) { return (rows + 15) & ~(int64_t)15; }
static inline double *vt_T(double *vt, int64_t rows) { return vt + panel_ldv(rows) * DHQR_NBV; }
static inline double *vt_Tt(double *vt, int64_t rows) { return vt_T(vt, rows) + DHQR_NBV * DHQR_NBV; }
static inline double *vt_alpha(double *vt, int64_t rows) { return vt_Tt(vt, rows) + DHQR_NBV * DHQR_NBV; }
static inline int64_t panel_elems(int64_t rows) {
  return panel_ldv(rows) * DHQR_NBV + 2 * DHQR_NBV * DHQR_NBV + DHQR_NBV;
}

// Split-K factor for k_gemm_tn: `ntiles` column tiles x ns row slabs should fill the 512 resident
// workgroup slots (256 CUs x 2) in whole waves -- 765 workgroups on 512 slots run at 75 %.
static void pick_split(int64_t rows, int64_t ntiles, int64_t target_wgs, int64_t max_split,
                       int64_t *nsplit, int64_t *rps) {
  const int64_t slots = 512;
  int64_t cap = std::min<int64_t>(max_split, std::max<int64_t>(1, rows / 128));
  int64_t best = 1;
  double best_score = -1.0;
  for (int64_t ns = 1; ns <= cap; ++ns) {
    const int64_t wgs = ntiles * ns;
    const int64_t waves = (wgs + slots - 1) / slots;
    double eff = (double)wgs / (double)(waves * slots);
    if (wgs < target_wgs && wgs < slots) eff *= 0.999;  // fine, just not full
    // prefer fewer splits at equal efficiency (less partial traffic); stop growing past 4 waves
    const double score = eff - 1e-4 * (double)ns - (waves > 4 ? 0.05 : 0.0);
    if (score > best_score) { best_score = score; best = ns; }
    if (wgs >= 4 * slots) break;
  }
  int64_t r = (rows + best - 1) / best;
  r = (r + G_KT - 1) / G_KT * G_KT;
  if (r < G_KT) r = G_KT;
  *rps = r;
  *nsplit = std::max<int64_t>(1, (rows + r - 1) / r);
}

// T / T' of a packed panel buffer whose V part is already in place (ncols real columns).
static int32_t panel_build_t(dhqr_ctx *c, int64_t rows, int64_t ncols, double *vt, int kw = DHQR_NBV);

// Pack V (R part zeroed) and build T / T' for a factored panel P (rows x ncols, ncols <= 128).
static int32_t panel_pack_and_t(dhqr_ctx *c, const double *P, int64_t rows, int64_t ncols,
                                int64_t ldp, const double *alpha, double *vt) {
  const int64_t ldv = panel_ldv(rows);
  double *V = vt;
  CHECK(prof_begin(c, CAT_TBUILD));
  {
    dim3 grid((unsigned)std::min<int64_t>((ldv + 255) / 256, 64), DHQR_NBV);
    hipLaunchKernelGGL(k_pack_v, grid, dim3(256), 0, c->stream, P, ldp, rows, ncols, V, ldv);
  }
  CHECK(panel_build_t(c, rows, ncols, vt));
  if (alpha) {  // nullptr: the caller already placed alpha in the buffer tail (or does not need it)
    HIPCHECK(hipMemsetAsync(vt_alpha(vt, rows), 0, DHQR_NBV * sizeof(double), c->stream));
    HIPCHECK(hipMemcpyAsync(vt_alpha(vt, rows), alpha, (size_t)ncols * sizeof(double),
                            hipMemcpyDeviceToDevice, c->stream));
  }
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

static int32_t panel_build_t(dhqr_ctx *c, int64_t rows, int64_t ncols, double *vt, int kw) {
  const int64_t ldv = panel_ldv(rows);
  double *V = vt;
  int64_t nsplit, rps;
  pick_split(rows, 1, 128, 128, &nsplit, &rps);
  CHECK(ensure(c, c->spart, (size_t)nsplit * DHQR_NBV * DHQR_NBV));
  CHECK(ensure(c, c->sfull, (size_t)DHQR_NBV * DHQR_NBV));
#define DHQR_SGEMM(KW_)                                                                           \
  hipLaunchKernelGGL((k_gemm_tn<2, 1, KW_>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, V, ldv, \
                     V, ldv, 1, (int64_t)0, rows, (int64_t)KW_, rps, c->spart.p, (int64_t)DHQR_NBV,      \
                     (int64_t)DHQR_NBV * DHQR_NBV)
  if (kw == 32) DHQR_SGEMM(32);
  else if (kw == 64) DHQR_SGEMM(64);
  else DHQR_SGEMM(128);
#undef DHQR_SGEMM
  hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)(DHQR_NBV * kw / 64)), dim3(256), 0, c->stream,
                     (const double *)c->spart.p, (int)nsplit, (int64_t)DHQR_NBV * DHQR_NBV,
                     (int64_t)DHQR_NBV * kw, c->sfull.p);
  launch_build_t(c, c->sfull.p, (int)ncols, vt_T(vt, rows), vt_Tt(vt, rows));
  LAUNCHCHECK();
  return DHQR_OK;
}

// C (rows x ncols) <- (I - V op(T) V') C with op(T) = T' (trans=1) or T (trans=0).
static int32_t panel_apply(dhqr_ctx *c, const double *vt, int64_t rows, double *C, int64_t ncols,
                           int64_t ldc, int trans, int kw = DHQR_NBV) {
  if (ncols <= 0 || rows <= 0) return DHQR_OK;
  const int64_t ldv = panel_ldv(rows);
  const double *V = vt;
  const double *Top = trans ? vt_T(const_cast<double *>(vt), rows) : vt_Tt(const_cast<double *>(vt), rows);
  const int64_t ntiles = (ncols + 127) / 128;
  int64_t nsplit, rps;
  // narrow updates (one or two column tiles: the look-ahead lane / a rank's single block) are split
  // over up to 256 row slabs so the latency-critical chain uses the whole chip
  pick_split(rows, ntiles, 512, ntiles <= 2 ? 256 : 64, &nsplit, &rps);
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  CHECK(ensure(c, ws.w1, (size_t)nsplit * DHQR_NBV * (size_t)ncols));
  CHECK(ensure(c, ws.w2, (size_t)DHQR_NBV * (size_t)ncols));
  if (nsplit > 1) CHECK(ensure(c, ws.w1r, (size_t)DHQR_NBV * (size_t)ncols));
  const bool vec = (ldc % 2 == 0) && (rows % 2 == 0) && aligned16(C);
  const int64_t wstride = (int64_t)DHQR_NBV * ncols;

  // one instantiation per reflector-block width (32 / 64 inside a panel, 128 for the trailing update)
#define DHQR_APPLY(KW_)                                                                              \
  do {                                                                                               \
    CHECK(prof_begin(c, CAT_VTA));                                                                   \
    if (vec)                                                                                         \
      hipLaunchKernelGGL((k_gemm_tn<2, 1, KW_>), dim3((unsigned)ntiles, (unsigned)nsplit), dim3(256), 0, \
                         c->stream, V, ldv, (const double *)C, ldc, 1, (int64_t)0, rows, ncols, rps,    \
                         ws.w1.p, (int64_t)DHQR_NBV, wstride);                                           \
    else                                                                                             \
      hipLaunchKernelGGL((k_gemm_tn<1, 1, KW_>), dim3((unsigned)ntiles, (unsigned)nsplit), dim3(256), 0, \
                         c->stream, V, ldv, (const double *)C, ldc, 1, (int64_t)0, rows, ncols, rps,    \
                         ws.w1.p, (int64_t)DHQR_NBV, wstride);                                           \
    CHECK(prof_end(c));                                                                              \
    CHECK(prof_begin(c, CAT_TW));                                                                    \
    const double *w1sum = ws.w1.p;                                                                   \
    if (nsplit > 1) { /* bandwidth-friendly, deterministic split-K reduction */                      \
      hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)((wstride + 63) / 64)), dim3(256), 0,       \
                         c->stream, (const double *)ws.w1.p, (int)nsplit, wstride, wstride, ws.w1r.p); \
      w1sum = ws.w1r.p;                                                                              \
    }                                                                                                \
    hipLaunchKernelGGL((k_gemm_tn<2, 1, KW_>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Top, \
                       (int64_t)DHQR_NBV, w1sum, (int64_t)DHQR_NBV, 1, (int64_t)0, (int64_t)KW_, ncols,  \
                       (int64_t)KW_, ws.w2.p, (int64_t)DHQR_NBV, (int64_t)0);                            \
    CHECK(prof_end(c));                                                                              \
    CHECK(prof_begin(c, CAT_AVW));                                                                   \
    const int64_t gx_ = (rows + 127) / 128;                                                          \
    const int swz_ = (c->swizzle && gx_ >= 16 && ntiles >= 16) ? 1 : 0;                              \
    dim3 grid((unsigned)gx_, (unsigned)ntiles);                                                      \
    if (swz_) grid = dim3((unsigned)((((gx_ + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);      \
    if (vec)                                                                                         \
      hipLaunchKernelGGL((k_gemm_nn_sub<2, KW_>), grid, dim3(256), 0, c->stream, V, ldv,             \
                         (const double *)ws.w2.p, (int64_t)DHQR_NBV, C, ldc, rows, ncols, swz_);     \
    else                                                                                             \
      hipLaunchKernelGGL((k_gemm_nn_sub<1, KW_>), grid, dim3(256), 0, c->stream, V, ldv,             \
                         (const double *)ws.w2.p, (int64_t)DHQR_NBV, C, ldc, rows, ncols, swz_);     \
    CHECK(prof_end(c));                                                                              \
  } while (0)
  if (kw == 32) DHQR_APPLY(32);
  else if (kw == 64) DHQR_APPLY(64);
  else DHQR_APPLY(128);
#undef DHQR_APPLY
  if (c->profiling) {
    c->st.flops_gemm_vta += 2.0 * DHQR_NBV * (double)rows * (double)ncols;
    c->st.flops_gemm_avw += 2.0 * DHQR_NBV * (double)rows * (double)ncols;
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

// ---- panel factorisation, row-split sub-panel version (dhqr_panel.h) --------------------------
// Factors the rows x w panel P in place, writes alpha[0:w], and leaves the packed V, T, T', alpha in
// `vt` (the operand of the trailing update / the broadcast buffer).
static int32_t factor_panel_v2(dhqr_ctx *c, double *P, int64_t rows, int64_t w, int64_t ldp,
                               double *alpha, double *vt) {
  const int64_t ldvw = panel_ldv(rows);
  const int ib = c->ib;
  const int64_t nchmax = (rows + PS_RC - 1) / PS_RC;
  const int64_t rpad = (rows + 31) & ~(int64_t)15;
  CHECK(ensure(c, c->pbuf, (size_t)(2 * rpad + 2 * DHQR_NBV + 2 * (int64_t)ib * nchmax + 64)));
  CHECK(ensure(c, c->vts, (size_t)panel_elems(rows)));
  double *piv[2] = {c->pbuf.p, c->pbuf.p + rpad};
  double *prow[2] = {c->pbuf.p + 2 * rpad, c->pbuf.p + 2 * rpad + DHQR_NBV};
  double *part[2] = {c->pbuf.p + 2 * rpad + 2 * DHQR_NBV, c->pbuf.p + 2 * rpad + 2 * DHQR_NBV + (int64_t)ib * nchmax};
  const bool vec = (ldp % 2 == 0) && (rows % 2 == 0) && aligned16(P);
  CHECK(prof_begin(c, CAT_PANEL));
  const bool was = c->profiling;
  c->profiling = false;  // the nested T builds / GEMMs are accounted to the panel
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemsetAsync(vt, 0, (size_t)ldvw * DHQR_NBV * sizeof(double), c->stream));
    for (int64_t j0 = 0; j0 < w; j0 += ib) {
      const int ncs = (int)std::min<int64_t>(ib, w - j0);
      const int64_t rows_s = rows - j0;
      double *Ps = P + j0 + j0 * ldp;
      const int nch = (int)((rows_s + PS_RC - 1) / PS_RC);
      const int64_t ldvs = panel_ldv(rows_s);
      HIPCHECK(hipMemsetAsync(c->vts.p, 0, (size_t)ldvs * (ncs <= 32 ? 32 : (ncs <= 64 ? 64 : 128)) * sizeof(double), c->stream));
      if (vec)
        hipLaunchKernelGGL((k_panel_init<2>), dim3(nch, ncs), dim3(256), 0, c->stream, (const double *)Ps, ldp,
                           rows_s, piv[0], prow[0], part[0], nch);
      else
        hipLaunchKernelGGL((k_panel_init<1>), dim3(nch, ncs), dim3(256), 0, c->stream, (const double *)Ps, ldp,
                           rows_s, piv[0], prow[0], part[0], nch);
      for (int q = 0; q < ncs; ++q) {
        dim3 grid(nch - q / PS_RC, 1 + (ncs - q - 1 + PS_CPW - 1) / PS_CPW);
        double *vwq = vt + j0 + (j0 + q) * ldvw;
        if (vec)
          hipLaunchKernelGGL((k_panel_step<2, PS_CPW>), grid, dim3(256), 0, c->stream, Ps, ldp, rows_s, q, ncs,
                             (const double *)piv[q & 1], piv[(q + 1) & 1], (const double *)prow[q & 1],
                             prow[(q + 1) & 1], (const double *)part[q & 1], part[(q + 1) & 1], nch, c->vts.p,
                             ldvs, vwq, ldvw, alpha + j0 + q);
        else
          hipLaunchKernelGGL((k_panel_step<1, PS_CPW>), grid, dim3(256), 0, c->stream, Ps, ldp, rows_s, q, ncs,
                             (const double *)piv[q & 1], piv[(q + 1) & 1], (const double *)prow[q & 1],
                             prow[(q + 1) & 1], (const double *)part[q & 1], part[(q + 1) & 1], nch, c->vts.p,
                             ldvs, vwq, ldvw, alpha + j0 + q);
      }
      if (j0 + ncs < w) {  // block reflector of this sub-panel onto the rest of the panel (MFMA)
        const int kw = ncs <= 32 ? 32 : (ncs <= 64 ? 64 : 128);
        CHECK(panel_build_t(c, rows_s, ncs, c->vts.p, kw));
        CHECK(panel_apply(c, c->vts.p, rows_s, P + j0 + (j0 + ncs) * ldp, w - j0 - ncs, ldp, 1, kw));
      }
    }
    {
      dim3 grid((unsigned)std::min<int64_t>((rows + 255) / 256, 64), (unsigned)w);
      hipLaunchKernelGGL(k_unpack_v, grid, dim3(256), 0, c->stream, P, ldp, rows, w, (const double *)vt, ldvw);
    }
    CHECK(panel_build_t(c, rows, w, vt));
    HIPCHECK(hipMemsetAsync(vt_alpha(vt, rows), 0, DHQR_NBV * sizeof(double), c->stream));
    HIPCHECK(hipMemcpyAsync(vt_alpha(vt, rows), alpha, (size_t)w * sizeof(double), hipMemcpyDeviceToDevice,
                            c->stream));
    LAUNCHCHECK();
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  if (c->profiling) {
    for (int64_t j = 0; j + 1 < w; ++j) c->st.bytes_panel += 16.0 * (double)(rows - j) * (double)(w - j - 1);
  }
  CHECK(prof_end(c));
  return DHQR_OK;
}

// ---- panel factorisation, R-first fast path (dhqr_recon.h) -------------------------------------
// G = X'X (128 x 128) for a rows x 128 operand: split-K TN GEMM + deterministic reduction.
static int32_t gram128(dhqr_ctx *c, const double *X, int64_t ldx, int64_t rows, double *out) {
  int64_t nsplit, rps;
  pick_split(rows, 1, 512, 256, &nsplit, &rps);
  CHECK(ensure(c, c->spart, (size_t)nsplit * DHQR_NBV * DHQR_NBV));
  const bool vec = (ldx % 2 == 0) && (rows % 2 == 0) && aligned16(X);
  if (vec)
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, X, ldx, X, ldx,
                       1, (int64_t)0, rows, (int64_t)DHQR_NBV, rps, c->spart.p, (int64_t)DHQR_NBV,
                       (int64_t)DHQR_NBV * DHQR_NBV);
  else
    hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, X, ldx, X, ldx,
                       1, (int64_t)0, rows, (int64_t)DHQR_NBV, rps, c->spart.p, (int64_t)DHQR_NBV,
                       (int64_t)DHQR_NBV * DHQR_NBV);
  hipLaunchKernelGGL(k_reduce_splits, dim3(DHQR_NBV * DHQR_NBV / 64), dim3(256), 0, c->stream,
                     (const double *)c->spart.p, (int)nsplit, (int64_t)DHQR_NBV * DHQR_NBV,
                     (int64_t)DHQR_NBV * DHQR_NBV, out);
  return DHQR_OK;
}
// out (rows x 128, ld ldo) = X (rows x 128, ld ldx) * Y with negY = -Y given (128 x 128, ld 128)
static int32_t mul128(dhqr_ctx *c, const double *X, int64_t ldx, int64_t rows, const double *negY, double *out,
                      int64_t ldo) {
  HIPCHECK(hipMemsetAsync(out, 0, (size_t)ldo * DHQR_NBV * sizeof(double), c->stream));
  const bool vec = (ldx % 2 == 0) && (rows % 2 == 0) && aligned16(X);
  dim3 grid((unsigned)((rows + 127) / 128), 1);
  if (vec)
    hipLaunchKernelGGL((k_gemm_nn_sub<2, 128>), grid, dim3(256), 0, c->stream, X, ldx, negY, (int64_t)DHQR_NBV, out,
                       ldo, rows, (int64_t)DHQR_NBV, 0);
  else
    hipLaunchKernelGGL((k_gemm_nn_sub<1, 128>), grid, dim3(256), 0, c->stream, X, ldx, negY, (int64_t)DHQR_NBV, out,
                       ldo, rows, (int64_t)DHQR_NBV, 0);
  return DHQR_OK;
}

static int32_t factor_panel_v2(dhqr_ctx *c, double *P, int64_t rows, int64_t w, int64_t ldp, double *alpha,
                               double *vt);

// Full-width panel (w == 128).  Nothing is written to P until the reflectors are verified; on a
// failed check (ill-conditioned panel) the robust column-by-column path runs on the untouched P.
// The verification flag is read on the host: one stream synchronisation per panel (the caller has
// already queued the concurrent trailing update on the other stream).
static int32_t factor_panel_v3(dhqr_ctx *c, double *P, int64_t rows, int64_t ldp, double *alpha, double *vt) {
  const int64_t ldv = panel_ldv(rows);
  const size_t NN = (size_t)DHQR_NBV * DHQR_NBV;
  CHECK(ensure(c, c->rbuf, 6 * NN + 1024));
  CHECK(ensure(c, c->vts, (size_t)panel_elems(rows)));
  CHECK(ensure(c, c->sfull, NN));
  double *R1 = c->rbuf.p, *negR1inv = R1 + NN, *Rf = R1 + 2 * NN, *Rref = R1 + 3 * NN, *negMinv = R1 + 4 * NN;
  double *G = R1 + 5 * NN, *altmp = R1 + 6 * NN;
  int *dflag = (int *)(altmp + 256);
  double *Q1 = c->vts.p;  // rows x 128 scratch, ld = ldv
  CHECK(prof_begin(c, CAT_PANEL));
  const bool was = c->profiling;
  c->profiling = false;
  bool ok = false;
  int passes = c->cholqr_passes;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemsetAsync(dflag, 0, 4 * sizeof(int), c->stream));
    CHECK(gram128(c, P, ldp, rows, G));                                            // G  = P'P
    if (passes == 2) {
      launch_chol_inv(c, G, nullptr, R1, negR1inv, dflag);                          // R1, -R1^{-1}
      CHECK(mul128(c, P, ldp, rows, negR1inv, Q1, ldv));                           // Q1 = P R1^{-1}
      CHECK(gram128(c, Q1, ldv, rows, G));                                         // G2 = Q1'Q1
      launch_chol_inv(c, G, R1, Rf, nullptr, dflag);                                // R  = chol(G2) R1
    } else {
      launch_chol_inv(c, G, nullptr, Rf, nullptr, dflag);                           // R = chol(P'P)
    }
    launch_recon_top(c, P, ldp, Rf, altmp, Rref, negMinv);                          // alpha, R_ref, -M^{-1}
    CHECK(mul128(c, P, ldp, rows, negMinv, vt, ldv));                              // Vw = P M^{-1}
    hipLaunchKernelGGL(k_recon_fix, dim3(NN / 256), dim3(256), 0, c->stream, vt, ldv, (const double *)altmp,
                       (const double *)negMinv);                                   // Vw = tril((P - aE) M^{-1})
    CHECK(gram128(c, vt, ldv, rows, c->sfull.p));                                  // S = V'V
    hipLaunchKernelGGL(k_recon_check, dim3(1), dim3(128), 0, c->stream, (const double *)c->sfull.p, c->recon_tol, dflag);
    HIPCHECK(hipMemcpyAsync(c->hflag, dflag, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    LAUNCHCHECK();
    HIPCHECK(hipStreamSynchronize(c->stream));
    ok = (c->hflag[0] == 0 && c->hflag[1] == 0);
    if (ok) {  // commit: reflectors, R, alpha, T
      dim3 grid((unsigned)std::min<int64_t>((rows + 255) / 256, 64), DHQR_NBV);
      hipLaunchKernelGGL(k_unpack_v, grid, dim3(256), 0, c->stream, P, ldp, rows, (int64_t)DHQR_NBV,
                         (const double *)vt, ldv);
      hipLaunchKernelGGL(k_recon_write_r, dim3(NN / 256), dim3(256), 0, c->stream, P, ldp, (const double *)Rref);
      HIPCHECK(hipMemcpyAsync(alpha, altmp, DHQR_NBV * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      HIPCHECK(hipMemcpyAsync(vt_alpha(vt, rows), altmp, DHQR_NBV * sizeof(double), hipMemcpyDeviceToDevice,
                              c->stream));
      launch_build_t(c, c->sfull.p, (int)DHQR_NBV, vt_T(vt, rows), vt_Tt(vt, rows));
      LAUNCHCHECK();
    }
    return DHQR_OK;
  };
  int32_t rc = body();
  if (rc == DHQR_OK && !ok && passes == 1) {  // moderately ill-conditioned panel: CholeskyQR2 before giving up
    passes = 2;
    rc = body();
  }
  c->profiling = was;
  CHECK(rc);
  if (ok) {
    c->n_fast++;
    if (c->profiling)
      for (int64_t j = 0; j + 1 < DHQR_NBV; ++j)
        c->st.bytes_panel += 16.0 * (double)(rows - j) * (double)(DHQR_NBV - j - 1);
    CHECK(prof_end(c));
    return DHQR_OK;
  }
  c->n_fallback++;
  CHECK(prof_end(c));
  return factor_panel_v2(c, P, rows, DHQR_NBV, ldp, alpha, vt);
}

// Factor one panel and leave (V, T, T', alpha) packed in vt.
static int32_t factor_panel(dhqr_ctx *c, double *P, int64_t rows, int64_t w, int64_t ldp, double *alpha,
                            double *vt) {
  if (c->panel_impl == 3 && w == DHQR_NBV && rows >= 2 * DHQR_NBV) return factor_panel_v3(c, P, rows, ldp, alpha, vt);
  if (c->panel_impl >= 2) return factor_panel_v2(c, P, rows, w, ldp, alpha, vt);
  CHECK(factor_unblocked_cols(c, P, rows, w, ldp, alpha, CAT_PANEL));
  return panel_pack_and_t(c, P, rows, w, ldp, alpha, vt);
}

// ---- blocked driver (BASELINE config 3) ------------------------------------------------------
// Right-looking with look-ahead depth 1 on two streams of the same GPU:
//   stream B (high priority): narrow update of block k+1 by panel k, then factorisation of
//                             panel k+1  (latency-bound, needs few CUs)
//   stream A (caller's)     : wide update of blocks >= k+2 by panel k  (MFMA-bound)
// so the panel factorisation the reference serialises in front of every trailing update
// (src:127-143) runs underneath the previous trailing update.
static int32_t factor_blocked_pair(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha);
static int32_t factor_blocked(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  const int64_t K = (n + DHQR_NBV - 1) / DHQR_NBV;
  CHECK(ensure(c, c->vt, (size_t)panel_elems(m)));
  if (c->lookahead && c->pair && K >= 4 && n % DHQR_NBV == 0 && n >= c->pair_min_n)
    return factor_blocked_pair(c, dA, m, n, lda, dalpha);
  if (!c->lookahead || K < 3) {
    for (int64_t c0 = 0; c0 < n; c0 += DHQR_NBV) {
      const int64_t w = std::min<int64_t>(DHQR_NBV, n - c0), rows = m - c0;
      double *P = dA + c0 + c0 * lda;
      CHECK(factor_panel(c, P, rows, w, lda, dalpha + c0, c->vt.p));
      if (c0 + w < n) CHECK(panel_apply(c, c->vt.p, rows, dA + c0 + (c0 + w) * lda, n - c0 - w, lda, 1));
    }
    return DHQR_OK;
  }
  // size every workspace up front: no (re)allocation while two streams are in flight
  CHECK(ensure(c, c->vt2, (size_t)panel_elems(m)));
  CHECK(ensure(c, c->vts, (size_t)panel_elems(m)));
  {
    const size_t ntmax = (size_t)((n + 127) / 128);
    const size_t w1cap = (size_t)DHQR_NBV * DHQR_NBV * (2048 + ntmax + 64);
    for (int s = 0; s < 2; ++s) {
      CHECK(ensure(c, c->ws[s].w1, s == 0 ? w1cap : (size_t)DHQR_NBV * DHQR_NBV * 1100));
      CHECK(ensure(c, c->ws[s].w1r, (size_t)DHQR_NBV * (size_t)n));
      CHECK(ensure(c, c->ws[s].w2, (size_t)DHQR_NBV * (size_t)n));
    }
    CHECK(ensure(c, c->spart, (size_t)128 * DHQR_NBV * DHQR_NBV));
    CHECK(ensure(c, c->sfull, (size_t)DHQR_NBV * DHQR_NBV));
  }
  double *vt[2] = {c->vt.p, c->vt2.p};
  hipStream_t sU = c->stream, sB = c->hi;          // sU: the caller's stream
  hipStream_t sA = sU;                             // stream of the wide updates
  auto on = [&](hipStream_t s, int wsi) { c->stream = s; c->cur_ws = wsi; };
  auto body = [&]() -> int32_t {
    // order stream B after whatever the caller queued on A (e.g. the fill)
    HIPCHECK(hipEventRecord(c->ev_wide[3], sU));
    HIPCHECK(hipStreamWaitEvent(sB, c->ev_wide[3], 0));
    if (sA != sU) HIPCHECK(hipStreamWaitEvent(sA, c->ev_wide[3], 0));
    on(sB, 1);
    CHECK(factor_panel(c, dA, m, std::min<int64_t>(DHQR_NBV, n), lda, dalpha, vt[0]));
    HIPCHECK(hipEventRecord(c->ev_panel[0], sB));
    for (int64_t k = 0; k < K; ++k) {
      const int64_t c0 = k * DHQR_NBV, w = std::min<int64_t>(DHQR_NBV, n - c0), rows = m - c0;
      if (k + 1 < K) {
        const int64_t c1 = c0 + w, w1 = std::min<int64_t>(DHQR_NBV, n - c1);
        const int64_t c2 = c1 + w1;
        // wide update first: the panel path below synchronises its stream on the host once
        on(sA, 0);
        HIPCHECK(hipStreamWaitEvent(sA, c->ev_panel[k & 3], 0));
        if (c2 < n) CHECK(panel_apply(c, vt[k & 1], rows, dA + c0 + c2 * lda, n - c2, lda, 1));
        HIPCHECK(hipEventRecord(c->ev_wide[k & 3], sA));
        on(sB, 1);
        if (k >= 1) HIPCHECK(hipStreamWaitEvent(sB, c->ev_wide[(k - 1) & 3], 0));  // block k+1 is current, vt[(k+1)&1] free
        {  // narrow (one block column) update of the look-ahead lane: accounted to the panel group
          CHECK(prof_begin(c, CAT_PANEL));
          const bool was = c->profiling;
          c->profiling = false;
          const int32_t rcn = panel_apply(c, vt[k & 1], rows, dA + c0 + c1 * lda, w1, lda, 1);
          c->profiling = was;
          CHECK(rcn);
          CHECK(prof_end(c));
        }
        CHECK(factor_panel(c, dA + c1 + c1 * lda, m - c1, w1, lda, dalpha + c1, vt[(k + 1) & 1]));
        HIPCHECK(hipEventRecord(c->ev_panel[(k + 1) & 3], sB));
      }
    }
    // the caller's stream owns the result: wait for the last panel (and the last wide update)
    HIPCHECK(hipStreamWaitEvent(sU, c->ev_panel[(K - 1) & 3], 0));
    if (sA != sU) {
      HIPCHECK(hipEventRecord(c->ev_wide[2], sA));
      HIPCHECK(hipStreamWaitEvent(sU, c->ev_wide[2], 0));
    }
    return DHQR_OK;
  };
  const int32_t rc = body();
  on(sU, 0);
  return rc;
}

// ---- two-panel trailing update (K = 256) ------------------------------------------------------
// C <- (I - V_b T_b' V_b')(I - V_a T_a' V_a') C in ONE pass over C for the subtraction:
//   W_a = T_a' (V_a' C),  W_b = T_b' (V_b' C - (V_b' V_a) W_a),  C -= [V_a V_b] [W_a; W_b].
// Vp = [V_a | V_b] (ldv x 256; V_b shifted down by 128 rows, zeros above), rows = rows of panel a.
// Halves the C read+write traffic of the NN GEMM per flop (0.125 -> 0.094 B/flop through the CU
// memory pipe), which is what bounds k_gemm_nn_sub.
__global__ __launch_bounds__(256) void k_copy_vb(const double *__restrict__ Vb, int64_t ldvb, int64_t rows_b,
                                                 double *__restrict__ Vp, int64_t ldv, int64_t rows_a) {
  const int64_t p = blockIdx.y;  // column of V_b
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ldv; r += stride) {
    double x = 0.0;
    if (r >= DHQR_NBV && r < rows_a && r - DHQR_NBV < rows_b) x = Vb[(r - DHQR_NBV) + p * ldvb];
    Vp[r + (DHQR_NBV + p) * ldv] = x;
  }
}

static int32_t pair_apply(dhqr_ctx *c, const double *Vp, int64_t ldv, int64_t rows, const double *Ta,
                          const double *Tb, const double *Sba, double *C, int64_t ncols, int64_t ldc) {
  if (ncols <= 0) return DHQR_OK;
  const int64_t rows_b = rows - DHQR_NBV;
  const int64_t ntiles = (ncols + 127) / 128;
  int64_t nsplit, rps;
  pick_split(rows, ntiles, 512, ntiles <= 2 ? 256 : 64, &nsplit, &rps);
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  CHECK(ensure(c, ws.w1, (size_t)nsplit * DHQR_NBV * (size_t)ncols));
  CHECK(ensure(c, ws.w1r, (size_t)DHQR_NBV * (size_t)ncols));
  CHECK(ensure(c, ws.w1r2, (size_t)DHQR_NBV * (size_t)ncols));
  CHECK(ensure(c, ws.w2, (size_t)2 * DHQR_NBV * (size_t)ncols));
  const bool vec = (ldc % 2 == 0) && (rows % 2 == 0) && aligned16(C);
  const int64_t wstride = (int64_t)DHQR_NBV * ncols;
  const double *Vb = Vp + DHQR_NBV + (int64_t)DHQR_NBV * ldv;  // first non-zero row of V_b
  const dim3 gtn((unsigned)ntiles, (unsigned)nsplit), gred((unsigned)((wstride + 63) / 64));

  CHECK(prof_begin(c, CAT_VTA));
  if (vec) {
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), gtn, dim3(256), 0, c->stream, Vp, ldv, (const double *)C, ldc, 1,
                       (int64_t)0, rows, ncols, rps, ws.w1.p, (int64_t)DHQR_NBV, wstride);
    hipLaunchKernelGGL(k_reduce_splits, gred, dim3(256), 0, c->stream, (const double *)ws.w1.p, (int)nsplit, wstride,
                       wstride, ws.w1r.p);
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), gtn, dim3(256), 0, c->stream, Vb, ldv, (const double *)(C + DHQR_NBV),
                       ldc, 1, (int64_t)0, rows_b, ncols, rps, ws.w1.p, (int64_t)DHQR_NBV, wstride);
  } else {
    hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), gtn, dim3(256), 0, c->stream, Vp, ldv, (const double *)C, ldc, 1,
                       (int64_t)0, rows, ncols, rps, ws.w1.p, (int64_t)DHQR_NBV, wstride);
    hipLaunchKernelGGL(k_reduce_splits, gred, dim3(256), 0, c->stream, (const double *)ws.w1.p, (int)nsplit, wstride,
                       wstride, ws.w1r.p);
    hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), gtn, dim3(256), 0, c->stream, Vb, ldv, (const double *)(C + DHQR_NBV),
                       ldc, 1, (int64_t)0, rows_b, ncols, rps, ws.w1.p, (int64_t)DHQR_NBV, wstride);
  }
  hipLaunchKernelGGL(k_reduce_splits, gred, dim3(256), 0, c->stream, (const double *)ws.w1.p, (int)nsplit, wstride,
                     wstride, ws.w1r2.p);
  CHECK(prof_end(c));

  CHECK(prof_begin(c, CAT_TW));
  const int64_t ld2 = 2 * DHQR_NBV;
  // W_a = T_a' Y_a  -> rows 0..127 of W2 (ld 256)
  hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Ta, (int64_t)DHQR_NBV,
                     (const double *)ws.w1r.p, (int64_t)DHQR_NBV, 1, (int64_t)0, (int64_t)DHQR_NBV, ncols,
                     (int64_t)DHQR_NBV, ws.w2.p, ld2, (int64_t)0);
  // Y_b -= (V_b' V_a) W_a
  hipLaunchKernelGGL((k_gemm_nn_sub<2, 128>), dim3(1, (unsigned)ntiles), dim3(256), 0, c->stream, Sba, (int64_t)DHQR_NBV,
                     (const double *)ws.w2.p, ld2, ws.w1r2.p, (int64_t)DHQR_NBV, (int64_t)DHQR_NBV, ncols, 0);
  // W_b = T_b' Y_b  -> rows 128..255 of W2
  hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Tb, (int64_t)DHQR_NBV,
                     (const double *)ws.w1r2.p, (int64_t)DHQR_NBV, 1, (int64_t)0, (int64_t)DHQR_NBV, ncols,
                     (int64_t)DHQR_NBV, ws.w2.p + DHQR_NBV, ld2, (int64_t)0);
  CHECK(prof_end(c));

  CHECK(prof_begin(c, CAT_AVW));
  const int64_t gx = (rows + 127) / 128;
  const int swz = (c->swizzle && gx >= 16 && ntiles >= 16) ? 1 : 0;
  dim3 grid((unsigned)gx, (unsigned)ntiles);
  if (swz) grid = dim3((unsigned)((((gx + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
  if (vec)
    hipLaunchKernelGGL((k_gemm_nn_sub<2, 256>), grid, dim3(256), 0, c->stream, Vp, ldv, (const double *)ws.w2.p, ld2, C,
                       ldc, rows, ncols, swz);
  else
    hipLaunchKernelGGL((k_gemm_nn_sub<1, 256>), grid, dim3(256), 0, c->stream, Vp, ldv, (const double *)ws.w2.p, ld2, C,
                       ldc, rows, ncols, swz);
  CHECK(prof_end(c));
  if (c->profiling) {
    c->st.flops_gemm_vta += 2.0 * DHQR_NBV * ((double)rows + (double)rows_b) * (double)ncols;
    c->st.flops_gemm_avw += 2.0 * DHQR_NBV * ((double)rows + (double)rows_b) * (double)ncols;
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

// Blocked driver, two panels per wide update.  Panels are paired (0,1), (2,3), ...; while the wide
// stream applies pair q to blocks >= 2q+4, the look-ahead lane brings blocks 2q+2 and 2q+3 up to
// date, factors them and assembles pair q+1.
static int32_t factor_blocked_pair(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  const int64_t K = (n + DHQR_NBV - 1) / DHQR_NBV, NB = DHQR_NBV;
  const int64_t ldvmax = panel_ldv(m);
  const size_t NN = (size_t)NB * NB;
  // size everything up front
  for (int s = 0; s < 2; ++s) CHECK(ensure(c, c->pairv[s], (size_t)ldvmax * 2 * NB + 2 * NN + NB + 1024));
  CHECK(ensure(c, c->pairt, 2 * 3 * NN));  // per pair buffer: T_a, T_b, S_ba
  CHECK(ensure(c, c->vt2, (size_t)panel_elems(m)));
  CHECK(ensure(c, c->vts, (size_t)panel_elems(m)));
  {
    const size_t ntmax = (size_t)((n + 127) / 128);
    const size_t w1cap = (size_t)NB * NB * (2048 + ntmax + 64);
    for (int s = 0; s < 2; ++s) {
      CHECK(ensure(c, c->ws[s].w1, s == 0 ? w1cap : (size_t)NB * NB * 1100));
      CHECK(ensure(c, c->ws[s].w1r, (size_t)NB * (size_t)n));
      CHECK(ensure(c, c->ws[s].w1r2, (size_t)NB * (size_t)n));
      CHECK(ensure(c, c->ws[s].w2, (size_t)2 * NB * (size_t)n));
    }
    CHECK(ensure(c, c->spart, (size_t)256 * NN));
    CHECK(ensure(c, c->sfull, NN));
    CHECK(ensure(c, c->rbuf, 6 * NN + 1024));
    CHECK(ensure(c, c->scratch, 4096));
  }
  hipStream_t sU = c->stream, sB = c->hi, sA = sU;
  auto on = [&](hipStream_t s, int wsi) { c->stream = s; c->cur_ws = wsi; };
  auto Ta = [&](int q) { return c->pairt.p + (size_t)(q & 1) * 3 * NN; };
  auto Tb = [&](int q) { return Ta(q) + NN; };
  auto Sba = [&](int q) { return Ta(q) + 2 * NN; };
  auto colptr = [&](int64_t rowblk, int64_t colblk) { return dA + rowblk * NB + colblk * NB * lda; };
  auto wcols = [&](int64_t k) { return std::min<int64_t>(NB, n - k * NB); };

  // lane helpers (stream sB, workspace set 1), all accounted to the panel group
  auto lane_single_apply = [&](const double *vt, int64_t k, int64_t blk) -> int32_t {
    return panel_apply(c, vt, m - k * NB, colptr(k, blk), wcols(blk), lda, 1);
  };
  // assemble pair q from panel a (already in pairv[q&1] as a standard VT) and panel b (in vt2)
  auto build_pair = [&](int q) -> int32_t {
    const int64_t a = 2 * (int64_t)q, rows_a = m - a * NB, rows_b = rows_a - NB;
    const int64_t ldva = panel_ldv(rows_a), ldvb = panel_ldv(rows_b);
    double *pv = c->pairv[q & 1].p;
    HIPCHECK(hipMemcpyAsync(Ta(q), vt_T(pv, rows_a), NN * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(Tb(q), vt_T(c->vt2.p, rows_b), NN * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    dim3 grid((unsigned)std::min<int64_t>((ldva + 255) / 256, 64), (unsigned)NB);
    hipLaunchKernelGGL(k_copy_vb, grid, dim3(256), 0, c->stream, (const double *)c->vt2.p, ldvb, rows_b, pv, ldva, rows_a);
    // S_ba = V_b' V_a over the rows of panel b
    int64_t nsplit, rps;
    pick_split(rows_b, 1, 512, 256, &nsplit, &rps);
    const double *Vb = pv + NB + NB * ldva;
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, Vb, ldva,
                       (const double *)(pv + NB), ldva, 1, (int64_t)0, rows_b, (int64_t)NB, rps, c->spart.p, (int64_t)NB,
                       (int64_t)NN);
    hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)(NN / 64)), dim3(256), 0, c->stream, (const double *)c->spart.p,
                       (int)nsplit, (int64_t)NN, (int64_t)NN, Sba(q));
    LAUNCHCHECK();
    return DHQR_OK;
  };

  auto body = [&]() -> int32_t {
    HIPCHECK(hipEventRecord(c->ev_wide[3], sU));
    HIPCHECK(hipStreamWaitEvent(sB, c->ev_wide[3], 0));
    on(sB, 1);
    const bool was0 = c->profiling;
    // ---- pair 0
    CHECK(factor_panel(c, dA, m, wcols(0), lda, dalpha, c->pairv[0].p));
    if (K >= 2) {
      CHECK(prof_begin(c, CAT_PANEL));
      c->profiling = false;
      int32_t r1 = lane_single_apply(c->pairv[0].p, 0, 1);
      c->profiling = was0;
      CHECK(r1);
      CHECK(prof_end(c));
      CHECK(factor_panel(c, colptr(1, 1), m - NB, wcols(1), lda, dalpha + NB, c->vt2.p));
      if (K >= 3) {
        CHECK(prof_begin(c, CAT_PANEL));
        c->profiling = false;
        r1 = build_pair(0);
        c->profiling = was0;
        CHECK(r1);
        CHECK(prof_end(c));
      }
    }
    HIPCHECK(hipEventRecord(c->ev_panel[0], sB));
    for (int q = 0;; ++q) {
      const int64_t a = 2 * (int64_t)q, b = a + 1;
      if (b + 1 >= K) break;  // no block after panel b: nothing left to update
      const int64_t A2 = b + 1, B2 = b + 2, W0 = b + 3;  // next pair's panels, first block of the wide update
      // wide update first (the lane below synchronises its stream on the host)
      on(sA, 0);
      HIPCHECK(hipStreamWaitEvent(sA, c->ev_panel[q & 3], 0));
      if (W0 < K)
        CHECK(pair_apply(c, c->pairv[q & 1].p, panel_ldv(m - a * NB), m - a * NB, Ta(q), Tb(q), Sba(q), colptr(a, W0),
                         n - W0 * NB, lda));
      HIPCHECK(hipEventRecord(c->ev_wide[q & 3], sA));
      // look-ahead lane: blocks A2, B2
      on(sB, 1);
      if (q >= 1) HIPCHECK(hipStreamWaitEvent(sB, c->ev_wide[(q - 1) & 3], 0));
      const bool was = c->profiling;
      CHECK(prof_begin(c, CAT_PANEL));
      c->profiling = false;
      // blocks A2 and B2 are adjacent: one 256-column pair update brings both up to date
      int32_t rc2 = pair_apply(c, c->pairv[q & 1].p, panel_ldv(m - a * NB), m - a * NB, Ta(q), Tb(q), Sba(q),
                               colptr(a, A2), wcols(A2) + (B2 < K ? wcols(B2) : 0), lda);
      c->profiling = was;
      CHECK(rc2);
      CHECK(prof_end(c));
      CHECK(factor_panel(c, colptr(A2, A2), m - A2 * NB, wcols(A2), lda, dalpha + A2 * NB, c->pairv[(q + 1) & 1].p));
      if (B2 < K) {
        CHECK(prof_begin(c, CAT_PANEL));
        c->profiling = false;
        rc2 = lane_single_apply(c->pairv[(q + 1) & 1].p, A2, B2);
        c->profiling = was;
        CHECK(rc2);
        CHECK(prof_end(c));
        CHECK(factor_panel(c, colptr(B2, B2), m - B2 * NB, wcols(B2), lda, dalpha + B2 * NB, c->vt2.p));
        if (B2 + 1 < K) {
          CHECK(prof_begin(c, CAT_PANEL));
          c->profiling = false;
          rc2 = build_pair(q + 1);
          c->profiling = was;
          CHECK(rc2);
          CHECK(prof_end(c));
        }
      }
      HIPCHECK(hipEventRecord(c->ev_panel[(q + 1) & 3], sB));
    }
    // join the lane back into the caller's stream
    HIPCHECK(hipEventRecord(c->ev_panel[3], sB));
    HIPCHECK(hipStreamWaitEvent(sU, c->ev_panel[3], 0));
    return DHQR_OK;
  };
  const int32_t rc = body();
  on(sU, 0);
  return rc;
}

static int32_t check_ctx(dhqr_ctx *c) {
  if (!c) return set_err(DHQR_EINVAL, "null context");
  HIPCHECK(hipSetDevice(c->device));
  return DHQR_OK;
}
// The reference's loops simply do not execute for a matrix without columns (src:127 `for j in Hl.colrange`,
// src:217 `for j in 1:n`): qr! returns an empty alpha, `\` an empty x.  Same here: n == 0 is a no-op.
static inline bool no_columns(int64_t m, int64_t n) { return n == 0 && m >= 0; }

static int32_t check_mat(const void *A, int64_t m, int64_t n, int64_t lda, bool need_tall) {
  if (!A) return set_err(DHQR_EINVAL, "null matrix pointer");
  if (m <= 0 || n <= 0) return set_err(DHQR_EINVAL, "m and n must be positive (m=%lld n=%lld)", (long long)m, (long long)n);
  if (need_tall && m < n) return set_err(DHQR_EINVAL, "m >= n required (m=%lld n=%lld)", (long long)m, (long long)n);
  if (lda < m) return set_err(DHQR_EINVAL, "leading dimension %lld < m=%lld", (long long)lda, (long long)m);
  return DHQR_OK;
}

// Q' B (trans=1) / Q B (trans=0) panel by panel; when `triangular` only columns >= the panel's
// first column are touched (B = [R;0] while forming Q*R).
static int32_t apply_q_impl(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda,
                            const double *dalpha, double *dB, int64_t nrhs, int64_t ldb, int trans,
                            bool triangular) {
  const int64_t npan = (n + DHQR_NBV - 1) / DHQR_NBV;
  CHECK(ensure(c, c->vt, (size_t)panel_elems(m)));
  // alpha is only carried along in the packed buffer; a dummy pointer is fine when absent
  for (int64_t q = 0; q < npan; ++q) {
    const int64_t k = trans ? q : npan - 1 - q;
    const int64_t c0 = k * DHQR_NBV, w = std::min<int64_t>(DHQR_NBV, n - c0), rows = m - c0;
    const double *P = dA + c0 + c0 * lda;
    CHECK(panel_pack_and_t(c, P, rows, w, lda, dalpha ? dalpha + c0 : nullptr, c->vt.p));
    if (triangular) {
      if (nrhs - c0 > 0) CHECK(panel_apply(c, c->vt.p, rows, dB + c0 + c0 * ldb, nrhs - c0, ldb, trans));
    } else {
      CHECK(panel_apply(c, c->vt.p, rows, dB + c0, nrhs, ldb, trans));
    }
  }
  return DHQR_OK;
}

// =================================================================================== C ABI
extern "C" {

int32_t dhqr_version(void) { return DHQR_VERSION; }
const char *dhqr_last_error(void) { return g_err; }

int32_t dhqr_device_count(int32_t *count) {
  if (!count) return set_err(DHQR_EINVAL, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return set_err(DHQR_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return DHQR_OK;
}

int32_t dhqr_create(dhqr_ctx **out, int32_t device) {
  if (!out) return set_err(DHQR_EINVAL, "null ctx out-pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_err(DHQR_ENODEVICE, "no HIP device visible (%s); libdhqr has no CPU fallback",
                   e != hipSuccess ? hipGetErrorString(e) : "count = 0");
  if (device < 0 || device >= n) return set_err(DHQR_EINVAL, "device %d out of range [0,%d)", device, n);
  HIPCHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_err(DHQR_ENODEVICE, "device %d is %s; this library is built for gfx950 only", device,
                   prop.gcnArchName);
  dhqr_ctx *c = new dhqr_ctx();
  c->device = device;
  memset(&c->st, 0, sizeof(c->st));
  auto init = [&]() -> int32_t {  // any failure below releases what was created so far (dhqr_destroy)
    HIPCHECK(hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking));
    c->stream = c->own;
    {
      int lo = 0, hi = 0;
      HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // hi = numerically lowest = highest priority
      HIPCHECK(hipStreamCreateWithPriority(&c->hi, hipStreamNonBlocking, hi));
      for (int i = 0; i < 4; ++i) {
        HIPCHECK(hipEventCreateWithFlags(&c->ev_panel[i], hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&c->ev_wide[i], hipEventDisableTiming));
      }
    }
    if (const char *e = getenv("DHQR_LOOKAHEAD")) c->lookahead = atoi(e) != 0;
    if (const char *e = getenv("DHQR_SWIZZLE")) c->swizzle = atoi(e) != 0;
    if (const char *e = getenv("DHQR_PAIR")) c->pair = atoi(e) != 0;
    if (const char *e = getenv("DHQR_PAIR_MIN_N")) c->pair_min_n = atoll(e);
    HIPCHECK(hipHostMalloc((void **)&c->hflag, 4 * sizeof(int), hipHostMallocDefault));
    return DHQR_OK;
  };
  const int32_t rc_init = init();
  if (rc_init != DHQR_OK) {
    (void)dhqr_destroy(c);
    return rc_init;
  }
  if (const char *e = getenv("DHQR_CHOLQR_PASSES")) c->cholqr_passes = atoi(e) == 2 ? 2 : 1;
  if (const char *e = getenv("DHQR_RECON_TOL")) c->recon_tol = atof(e);
  if (const char *e = getenv("DHQR_SMALLK")) {
    const int v = atoi(e);
    c->smallk = (v == 4 || v == 5) ? v : 3;
  }
  if (const char *e = getenv("DHQR_PANEL")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 3) c->panel_impl = v;
  }
  if (const char *e = getenv("DHQR_IB")) {
    const int v = atoi(e);
    if (v == 16 || v == 32 || v == 64 || v == 128) c->ib = v;
  }
  *out = c;
  return DHQR_OK;
}

int32_t dhqr_destroy(dhqr_ctx *c) {
  if (!c) return DHQR_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  Buf *bufs[] = {&c->vbuf, &c->vt, &c->vt2, &c->vts, &c->ws[0].w1, &c->ws[0].w1r, &c->ws[0].w1r2, &c->ws[0].w2, &c->ws[1].w1,
                 &c->ws[1].w1r, &c->ws[1].w1r2, &c->ws[1].w2, &c->pairv[0], &c->pairv[1], &c->pairt, &c->spart, &c->sfull, &c->scratch, &c->pbuf, &c->rbuf};
  for (Buf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto &e : c->evs) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (int i = 0; i < 4; ++i) {
    if (c->ev_panel[i]) (void)hipEventDestroy(c->ev_panel[i]);
    if (c->ev_wide[i]) (void)hipEventDestroy(c->ev_wide[i]);
  }
  if (c->hflag) (void)hipHostFree(c->hflag);
  if (c->hi) (void)hipStreamDestroy(c->hi);
  if (c->own) (void)hipStreamDestroy(c->own);
  delete c;
  return DHQR_OK;
}

int32_t dhqr_set_stream(dhqr_ctx *c, void *s) {
  CHECK(check_ctx(c));
  c->stream = (hipStream_t)s;  // NULL is the device's default (null) stream, as torch uses it
  return DHQR_OK;
}
int32_t dhqr_use_own_stream(dhqr_ctx *c) {
  CHECK(check_ctx(c));
  c->stream = c->own;
  return DHQR_OK;
}
int32_t dhqr_synchronize(dhqr_ctx *c) {
  CHECK(check_ctx(c));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}
int32_t dhqr_set_profiling(dhqr_ctx *c, int32_t on) {
  CHECK(check_ctx(c));
  if (c->profiling && !on) CHECK(prof_resolve(c));
  c->profiling = on != 0;
  return DHQR_OK;
}
int32_t dhqr_reset_stats(dhqr_ctx *c) {
  CHECK(check_ctx(c));
  HIPCHECK(hipStreamSynchronize(c->stream));
  c->ev_used = 0;
  memset(&c->st, 0, sizeof(c->st));
  c->n_fast = c->n_fallback = 0;
  return DHQR_OK;
}
int32_t dhqr_get_stats(dhqr_ctx *c, dhqr_stats *out) {
  CHECK(check_ctx(c));
  if (!out) return set_err(DHQR_EINVAL, "null stats pointer");
  CHECK(prof_resolve(c));
  *out = c->st;
  return DHQR_OK;
}

int32_t dhqr_get_panel_counters(dhqr_ctx *c, int64_t *n_fast, int64_t *n_fallback) {
  CHECK(check_ctx(c));
  if (n_fast) *n_fast = c->n_fast;
  if (n_fallback) *n_fallback = c->n_fallback;
  return DHQR_OK;
}

int32_t dhqr_fill_uniform_f64(dhqr_ctx *c, double *dA, int64_t rows, int64_t cols, int64_t lda,
                              uint64_t seed, int64_t global_m, int64_t row0, int64_t colblock,
                              int32_t nranks, int32_t rank) {
  CHECK(check_ctx(c));
  CHECK(check_mat(dA, rows, cols, lda, false));
  if (colblock <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || global_m < rows)
    return set_err(DHQR_EINVAL, "bad layout arguments to dhqr_fill_uniform_f64");
  const int64_t total = rows * cols;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(k_fill_uniform, dim3(grid), dim3(256), 0, c->stream, dA, rows, cols, lda, seed,
                     global_m, row0, colblock, (int)nranks, (int)rank);
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_factor_f64(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha,
                        int32_t nb) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  if (!dalpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  if (nb != 0 && nb != DHQR_NB)
    return set_err(DHQR_EINVAL, "nb must be 0 (unblocked) or %d (blocked); got %d", DHQR_NB, nb);
  if (nb == 0) return factor_unblocked_cols(c, dA, m, n, lda, dalpha, CAT_RANK1);
  return factor_blocked(c, dA, m, n, lda, dalpha);
}

int32_t dhqr_qr_f64(dhqr_ctx *c, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha,
                    int32_t nb) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  double *dA = nullptr, *dal = nullptr;
  const int64_t ldd = (m + 1) & ~(int64_t)1;
  if (hipMalloc((void **)&dA, (size_t)ldd * n * sizeof(double)) != hipSuccess)
    return set_err(DHQR_ENOMEM, "hipMalloc of the %lld x %lld matrix failed", (long long)m, (long long)n);
  if (hipMalloc((void **)&dal, (size_t)n * sizeof(double)) != hipSuccess) {
    (void)hipFree(dA);
    return set_err(DHQR_ENOMEM, "hipMalloc of alpha failed");
  }
  int32_t rc = DHQR_OK;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemcpy2DAsync(dA, ldd * sizeof(double), hA, lda * sizeof(double), m * sizeof(double),
                              n, hipMemcpyHostToDevice, c->stream));
    // the reference factors whatever m x n block it is given; m odd is handled by the scalar path
    CHECK(dhqr_factor_f64(c, dA, m, n, ldd, dal, nb));
    HIPCHECK(hipMemcpy2DAsync(hA, lda * sizeof(double), dA, ldd * sizeof(double), m * sizeof(double),
                              n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(halpha, dal, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dal);
  return rc;
}

int32_t dhqr_solve_f64(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda,
                       const double *dalpha, double *db) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  if (!dalpha || !db) return set_err(DHQR_EINVAL, "null alpha or b pointer");
  CHECK(prof_begin(c, CAT_SOLVE));
  const bool was = c->profiling;
  c->profiling = false;  // the solve is timed as one group
  int32_t rc = apply_q_impl(c, dA, m, n, lda, dalpha, db, 1, m, 1, false);  // src:215-242
  if (rc == DHQR_OK) {
    for (int64_t hi = n; hi > 0; hi -= BS_NB) {  // src:244-282
      const int64_t lo = std::max<int64_t>(0, hi - BS_NB);
      hipLaunchKernelGGL(k_backsub_diag, dim3(1), dim3(64), 0, c->stream, dA, lda, dalpha, db, lo, hi);
      if (lo > 0)
        hipLaunchKernelGGL(k_backsub_update, dim3((unsigned)((lo + 255) / 256)), dim3(256), 0,
                           c->stream, dA, lda, db, lo, hi);
    }
  }
  c->profiling = was;
  CHECK(rc);
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_backsub_block_f64(dhqr_ctx *c, const double *dAcols, int64_t lda, const double *dalpha,
                               double *db, int64_t lo, int64_t hi, int32_t do_diag, int32_t do_update) {
  CHECK(check_ctx(c));
  if (!dAcols || !dalpha || !db) return set_err(DHQR_EINVAL, "null pointer argument");
  if (lo < 0 || hi <= lo) return set_err(DHQR_EINVAL, "bad block [%lld,%lld)", (long long)lo, (long long)hi);
  for (int64_t h = hi; h > lo; h -= BS_NB) {  // blocks wider than 64 are walked in 64-row steps
    const int64_t l = std::max<int64_t>(lo, h - BS_NB);
    if (do_diag) {
      hipLaunchKernelGGL(k_backsub_diag, dim3(1), dim3(64), 0, c->stream, dAcols, lda, dalpha, db, l, h);
      if (l > lo)  // rows of this block above the 64-row step just solved
        hipLaunchKernelGGL(k_backsub_update_range, dim3((unsigned)((l - lo + 255) / 256)), dim3(256), 0,
                           c->stream, dAcols, lda, db, lo, l, l, h);
    }
    if (do_update && lo > 0)  // rows above the block
      hipLaunchKernelGGL(k_backsub_update_range, dim3((unsigned)((lo + 255) / 256)), dim3(256), 0,
                         c->stream, dAcols, lda, db, (int64_t)0, lo, l, h);
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_ldiv_f64(dhqr_ctx *c, const double *hA, int64_t m, int64_t n, int64_t lda,
                      const double *halpha, const double *hb, double *hx) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  double *dA = nullptr, *dal = nullptr, *db = nullptr;
  const int64_t ldd = (m + 1) & ~(int64_t)1;
  if (hipMalloc((void **)&dA, (size_t)ldd * n * sizeof(double)) != hipSuccess)
    return set_err(DHQR_ENOMEM, "hipMalloc failed");
  if (hipMalloc((void **)&dal, (size_t)n * sizeof(double)) != hipSuccess) { (void)hipFree(dA); return set_err(DHQR_ENOMEM, "hipMalloc failed"); }
  if (hipMalloc((void **)&db, (size_t)(m + 2) * sizeof(double)) != hipSuccess) { (void)hipFree(dA); (void)hipFree(dal); return set_err(DHQR_ENOMEM, "hipMalloc failed"); }
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemcpy2DAsync(dA, ldd * sizeof(double), hA, lda * sizeof(double), m * sizeof(double),
                              n, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(dal, halpha, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(db, hb, m * sizeof(double), hipMemcpyHostToDevice, c->stream));  // src:318 copy of b
    CHECK(dhqr_solve_f64(c, dA, m, n, ldd, dal, db));
    HIPCHECK(hipMemcpyAsync(hx, db, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));  // src:320
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dal);
  (void)hipFree(db);
  return rc;
}

int32_t dhqr_partialdot_f64(dhqr_ctx *c, const double *da, const double *db, int64_t lo, int64_t hi,
                            double *hout) {
  CHECK(check_ctx(c));
  if (!da || !db || !hout) return set_err(DHQR_EINVAL, "null pointer argument");
  if (lo < 0 || hi < lo) return set_err(DHQR_EINVAL, "bad range [%lld,%lld)", (long long)lo, (long long)hi);
  CHECK(ensure(c, c->scratch, 4096));
  const int64_t len = hi - lo;
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((len + 255) / 256, 1024));
  hipLaunchKernelGGL(k_partialdot_partial, dim3(nblk), dim3(256), 0, c->stream, da, db, lo, hi, c->scratch.p);
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, c->stream, (const double *)c->scratch.p, nblk, c->scratch.p + 2048);
  LAUNCHCHECK();
  HIPCHECK(hipMemcpyAsync(hout, c->scratch.p + 2048, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}

int32_t dhqr_partialdot_host_f64(dhqr_ctx *c, const double *ha, const double *hb, int64_t lo, int64_t hi,
                                 double *hout) {
  CHECK(check_ctx(c));
  if (!ha || !hb || !hout) return set_err(DHQR_EINVAL, "null pointer argument");
  if (lo < 0 || hi < lo) return set_err(DHQR_EINVAL, "bad range [%lld,%lld)", (long long)lo, (long long)hi);
  if (hi == lo) { *hout = 0.0; return DHQR_OK; }
  double *d = nullptr;
  const size_t len = (size_t)(hi - lo);
  if (hipMalloc((void **)&d, 2 * len * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemcpyAsync(d, ha + lo, len * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(d + len, hb + lo, len * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return dhqr_partialdot_f64(c, d, d + len, 0, (int64_t)len, hout);
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  return rc;
}

// ---------------------------------------------------------------------------- ComplexF64
// (re, im) interleaved; device code sees double2 elements.
static int32_t check_zptr(const void *p, const char *what) {
  if (!p) return set_err(DHQR_EINVAL, "null %s pointer", what);
  if (!aligned16(p)) return set_err(DHQR_EINVAL, "%s pointer must be 16-byte aligned (ComplexF64 elements)", what);
  return DHQR_OK;
}

int32_t dhqr_factor_c64(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_zptr(dA, "matrix"));
  CHECK(check_zptr(dalpha, "alpha"));
  const size_t vlen = (size_t)((m + 15) & ~(int64_t)15);  // complex elements per staging vector
  CHECK(ensure(c, c->vbuf, 4 * vlen));
  double2 *A = reinterpret_cast<double2 *>(dA), *al = reinterpret_cast<double2 *>(dalpha);
  double2 *vb[2] = {reinterpret_cast<double2 *>(c->vbuf.p), reinterpret_cast<double2 *>(c->vbuf.p) + vlen};
  CHECK(prof_begin(c, CAT_RANK1));
  hipLaunchKernelGGL((k_zreflector<1024>), dim3(1), dim3(1024), 0, c->stream, A, m, (int64_t)0, vb[0], al);
  CHECK(prof_end(c));
  for (int64_t j = 0; j + 1 < n; ++j) {
    const unsigned nupd = (unsigned)(n - (j + 1));
    const int64_t cov = m - j;
    CHECK(prof_begin(c, CAT_RANK1));
    if (cov <= 1024)
      hipLaunchKernelGGL((k_zrank1<256>), dim3(nupd), dim3(256), 0, c->stream, A, lda, m, j,
                         (const double2 *)vb[j & 1], vb[(j + 1) & 1], al);
    else if (cov <= 4096)
      hipLaunchKernelGGL((k_zrank1<512>), dim3(nupd), dim3(512), 0, c->stream, A, lda, m, j,
                         (const double2 *)vb[j & 1], vb[(j + 1) & 1], al);
    else
      hipLaunchKernelGGL((k_zrank1<1024>), dim3(nupd), dim3(1024), 0, c->stream, A, lda, m, j,
                         (const double2 *)vb[j & 1], vb[(j + 1) & 1], al);
    CHECK(prof_end(c));
    if (c->profiling) c->st.bytes_rank1 += 32.0 * (double)cov * (double)nupd;
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_qr_c64(dhqr_ctx *c, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  double *dA = nullptr, *dal = nullptr;
  const size_t esz = 2 * sizeof(double);
  if (hipMalloc((void **)&dA, (size_t)m * n * esz) != hipSuccess)
    return set_err(DHQR_ENOMEM, "hipMalloc of the %lld x %lld complex matrix failed", (long long)m, (long long)n);
  if (hipMalloc((void **)&dal, (size_t)n * esz) != hipSuccess) {
    (void)hipFree(dA);
    return set_err(DHQR_ENOMEM, "hipMalloc of alpha failed");
  }
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemcpy2DAsync(dA, m * esz, hA, lda * esz, m * esz, n, hipMemcpyHostToDevice, c->stream));
    CHECK(dhqr_factor_c64(c, dA, m, n, m, dal));
    HIPCHECK(hipMemcpy2DAsync(hA, lda * esz, dA, m * esz, m * esz, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(halpha, dal, n * esz, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dal);
  return rc;
}

int32_t dhqr_solve_c64(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda,
                       const double *dalpha, double *db) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_zptr(dA, "matrix"));
  CHECK(check_zptr(dalpha, "alpha"));
  CHECK(check_zptr(db, "b"));
  const double2 *A = reinterpret_cast<const double2 *>(dA), *al = reinterpret_cast<const double2 *>(dalpha);
  double2 *b = reinterpret_cast<double2 *>(db);
  CHECK(prof_begin(c, CAT_SOLVE));
  for (int64_t j = 0; j < n; ++j) {  // src:215-224: reflectors in column order
    const int64_t cov = m - j;
    if (cov <= 2048)
      hipLaunchKernelGGL((k_zqtb_col<256>), dim3(1), dim3(256), 0, c->stream, A + j * lda, b, m, j);
    else
      hipLaunchKernelGGL((k_zqtb_col<1024>), dim3(1), dim3(1024), 0, c->stream, A + j * lda, b, m, j);
  }
  for (int64_t hi = n; hi > 0; hi -= ZBS_NB) {  // src:244-254
    const int64_t lo = std::max<int64_t>(0, hi - ZBS_NB);
    hipLaunchKernelGGL(k_zbacksub_diag, dim3(1), dim3(64), 0, c->stream, A, lda, al, b, lo, hi);
    if (lo > 0)
      hipLaunchKernelGGL(k_zbacksub_update, dim3((unsigned)((lo + 255) / 256)), dim3(256), 0, c->stream, A,
                         lda, b, lo, hi);
  }
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_ldiv_c64(dhqr_ctx *c, const double *hA, int64_t m, int64_t n, int64_t lda,
                      const double *halpha, const double *hb, double *hx) {
  CHECK(check_ctx(c));
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  double *dA = nullptr, *dal = nullptr, *db = nullptr;
  const size_t esz = 2 * sizeof(double);
  if (hipMalloc((void **)&dA, (size_t)m * n * esz) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
  if (hipMalloc((void **)&dal, (size_t)n * esz) != hipSuccess) { (void)hipFree(dA); return set_err(DHQR_ENOMEM, "hipMalloc failed"); }
  if (hipMalloc((void **)&db, (size_t)m * esz) != hipSuccess) { (void)hipFree(dA); (void)hipFree(dal); return set_err(DHQR_ENOMEM, "hipMalloc failed"); }
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemcpy2DAsync(dA, m * esz, hA, lda * esz, m * esz, n, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(dal, halpha, n * esz, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(db, hb, m * esz, hipMemcpyHostToDevice, c->stream));  // src:318 copy of b
    CHECK(dhqr_solve_c64(c, dA, m, n, m, dal, db));
    HIPCHECK(hipMemcpyAsync(hx, db, n * esz, hipMemcpyDeviceToHost, c->stream));  // src:320
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dal);
  (void)hipFree(db);
  return rc;
}

int32_t dhqr_partialdot_c64(dhqr_ctx *c, const double *da, const double *db, int64_t lo, int64_t hi,
                            double *hout) {
  CHECK(check_ctx(c));
  CHECK(check_zptr(da, "a"));
  CHECK(check_zptr(db, "b"));
  if (!hout) return set_err(DHQR_EINVAL, "null output pointer");
  if (lo < 0 || hi < lo) return set_err(DHQR_EINVAL, "bad range [%lld,%lld)", (long long)lo, (long long)hi);
  CHECK(ensure(c, c->scratch, 4096));
  const int64_t len = hi - lo;
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((len + 255) / 256, 1024));
  hipLaunchKernelGGL(k_zpartialdot_partial, dim3(nblk), dim3(256), 0, c->stream,
                     reinterpret_cast<const double2 *>(da), reinterpret_cast<const double2 *>(db), lo, hi,
                     c->scratch.p);
  hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(256), 0, c->stream, (const double *)c->scratch.p, nblk,
                     c->scratch.p + 2048);
  LAUNCHCHECK();
  HIPCHECK(hipMemcpyAsync(hout, c->scratch.p + 2048, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}

int32_t dhqr_partialdot_host_c64(dhqr_ctx *c, const double *ha, const double *hb, int64_t lo, int64_t hi,
                                 double *hout) {
  CHECK(check_ctx(c));
  if (!ha || !hb || !hout) return set_err(DHQR_EINVAL, "null pointer argument");
  if (lo < 0 || hi < lo) return set_err(DHQR_EINVAL, "bad range [%lld,%lld)", (long long)lo, (long long)hi);
  if (hi == lo) { hout[0] = 0.0; hout[1] = 0.0; return DHQR_OK; }
  double *d = nullptr;
  const size_t len = (size_t)(hi - lo), esz = 2 * sizeof(double);
  if (hipMalloc((void **)&d, 2 * len * esz) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemcpyAsync(d, ha + 2 * lo, len * esz, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(d + 2 * len, hb + 2 * lo, len * esz, hipMemcpyHostToDevice, c->stream));
    return dhqr_partialdot_c64(c, d, d + 2 * len, 0, (int64_t)len, hout);
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  return rc;
}

int32_t dhqr_apply_q_f64(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda, double *dB,
                         int64_t nrhs, int64_t ldb, int32_t trans) {
  CHECK(check_ctx(c));
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_mat(dB, m, nrhs, ldb, false));
  return apply_q_impl(c, dA, m, n, lda, nullptr, dB, nrhs, ldb, trans ? 1 : 0, false);
}

int32_t dhqr_residual_f64(dhqr_ctx *c, const double *dAfact, int64_t m, int64_t n, int64_t lda,
                          const double *dalpha, const double *dAorig, int64_t ldo, double *dwork,
                          double *hrel) {
  CHECK(check_ctx(c));
  CHECK(check_mat(dAfact, m, n, lda, true));
  CHECK(check_mat(dAorig, m, n, ldo, true));
  if (!dalpha || !dwork || !hrel) return set_err(DHQR_EINVAL, "null pointer argument");
  {
    dim3 grid((unsigned)std::min<int64_t>((m + 255) / 256, 128), (unsigned)std::min<int64_t>(n, 32768));
    hipLaunchKernelGGL(k_form_r0, grid, dim3(256), 0, c->stream, dAfact, lda, dalpha, m, n, dwork, m,
                       (int64_t)DHQR_NBV, 1, 0);
  }
  const bool was = c->profiling;
  c->profiling = false;
  int32_t rc = apply_q_impl(c, dAfact, m, n, lda, dalpha, dwork, n, m, 0, true);
  c->profiling = was;
  CHECK(rc);
  CHECK(ensure(c, c->scratch, 4096));
  const int nblk = 1024;
  hipLaunchKernelGGL(k_diff_norms, dim3(nblk), dim3(256), 0, c->stream, dAorig, ldo, (const double *)dwork, m, m, n, c->scratch.p);
  hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(256), 0, c->stream, (const double *)c->scratch.p, nblk, c->scratch.p + 2048);
  LAUNCHCHECK();
  double h[2] = {0, 0};
  HIPCHECK(hipMemcpyAsync(h, c->scratch.p + 2048, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  *hrel = std::sqrt(h[0] / h[1]);
  return DHQR_OK;
}

int64_t dhqr_panel_ldv(int64_t rows) { return panel_ldv(rows); }
int64_t dhqr_panel_buffer_elems(int64_t rows) { return panel_elems(rows); }

int32_t dhqr_panel_factor_f64(dhqr_ctx *c, double *dP, int64_t rows, int64_t ncols, int64_t ldp,
                              double *dVT) {
  CHECK(check_ctx(c));
  CHECK(check_mat(dP, rows, ncols, ldp, true));
  if (ncols > DHQR_NB) return set_err(DHQR_EINVAL, "panel wider than %d", DHQR_NB);
  if (!dVT || !aligned16(dVT)) return set_err(DHQR_EINVAL, "dVT must be a 16-byte aligned device buffer");
  // alpha is produced in a small scratch vector and packed into the tail of dVT by factor_panel
  CHECK(ensure(c, c->scratch, 4096));
  double *al = c->scratch.p + 3072;
  return factor_panel(c, dP, rows, ncols, ldp, al, dVT);
}

int32_t dhqr_panel_pack_f64(dhqr_ctx *c, const double *dP, int64_t rows, int64_t ncols, int64_t ldp,
                            double *dVT) {
  CHECK(check_ctx(c));
  CHECK(check_mat(dP, rows, ncols, ldp, true));
  if (ncols > DHQR_NB) return set_err(DHQR_EINVAL, "panel wider than %d", DHQR_NB);
  if (!dVT || !aligned16(dVT)) return set_err(DHQR_EINVAL, "dVT must be a 16-byte aligned device buffer");
  HIPCHECK(hipMemsetAsync(vt_alpha(dVT, rows), 0, DHQR_NBV * sizeof(double), c->stream));
  return panel_pack_and_t(c, dP, rows, ncols, ldp, nullptr, dVT);
}

int32_t dhqr_form_r0_f64(dhqr_ctx *c, const double *dA, int64_t m, int64_t cols, int64_t lda,
                         const double *dalpha, double *dW, int64_t ldw, int64_t colblock,
                         int32_t nranks, int32_t rank) {
  CHECK(check_ctx(c));
  if (cols == 0) return DHQR_OK;
  CHECK(check_mat(dA, m, cols, lda, false));
  CHECK(check_mat(dW, m, cols, ldw, false));
  if (!dalpha || colblock <= 0 || nranks <= 0 || rank < 0 || rank >= nranks)
    return set_err(DHQR_EINVAL, "bad arguments to dhqr_form_r0_f64");
  dim3 grid((unsigned)std::min<int64_t>((m + 255) / 256, 128), (unsigned)std::min<int64_t>(cols, 32768));
  hipLaunchKernelGGL(k_form_r0, grid, dim3(256), 0, c->stream, dA, lda, dalpha, m, cols, dW, ldw, colblock,
                     (int)nranks, (int)rank);
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_diff_norms_f64(dhqr_ctx *c, const double *dX, int64_t ldx, const double *dY, int64_t ldy,
                            int64_t m, int64_t n, double *hout2) {
  CHECK(check_ctx(c));
  if (!hout2) return set_err(DHQR_EINVAL, "null output");
  hout2[0] = hout2[1] = 0.0;
  if (m == 0 || n == 0) return DHQR_OK;
  CHECK(check_mat(dX, m, n, ldx, false));
  CHECK(check_mat(dY, m, n, ldy, false));
  CHECK(ensure(c, c->scratch, 4096));
  const int nblk = 1024;
  hipLaunchKernelGGL(k_diff_norms, dim3(nblk), dim3(256), 0, c->stream, dX, ldx, dY, ldy, m, n, c->scratch.p);
  hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(256), 0, c->stream, (const double *)c->scratch.p, nblk, c->scratch.p + 2048);
  LAUNCHCHECK();
  HIPCHECK(hipMemcpyAsync(hout2, c->scratch.p + 2048, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}

// ---- row-split building blocks (BASELINE configs[4]: tall-skinny, rows distributed over ranks) ----
// Each call works on the caller's LOCAL row slab; the sums over ranks (Gram matrices, V'C partial
// dots) are all-reduced by the host layer (rowsplit.py) between calls.
int32_t dhqr_rs_gram_f64(dhqr_ctx *c, const double *dX, int64_t ldx, int64_t rows, double *dG) {
  CHECK(check_ctx(c));
  if (!dX || !dG) return set_err(DHQR_EINVAL, "null pointer argument");
  if (rows <= 0) {  // a rank may own no active rows of this panel
    HIPCHECK(hipMemsetAsync(dG, 0, (size_t)DHQR_NBV * DHQR_NBV * sizeof(double), c->stream));
    return DHQR_OK;
  }
  CHECK(gram128(c, dX, ldx, rows, dG));
  LAUNCHCHECK();
  return DHQR_OK;
}
// R = chol(G) (upper, dense 128 x 128).  flags are left on the device: dflag[0] != 0 on breakdown.
int32_t dhqr_rs_chol_f64(dhqr_ctx *c, const double *dG, double *dR, int32_t *dflag) {
  CHECK(check_ctx(c));
  if (!dG || !dR || !dflag) return set_err(DHQR_EINVAL, "null pointer argument");
  launch_chol_inv(c, dG, nullptr, dR, nullptr, (int *)dflag);
  LAUNCHCHECK();
  return DHQR_OK;
}
// Top-block replay on the rank that owns the panel's diagonal rows: dPtop = &P[diag row, first col].
int32_t dhqr_rs_recon_top_f64(dhqr_ctx *c, const double *dPtop, int64_t ldp, const double *dR, double *dalpha128,
                              double *dRref, double *dnegMinv) {
  CHECK(check_ctx(c));
  if (!dPtop || !dR || !dalpha128 || !dRref || !dnegMinv) return set_err(DHQR_EINVAL, "null pointer argument");
  launch_recon_top(c, dPtop, ldp, dR, dalpha128, dRref, dnegMinv);
  LAUNCHCHECK();
  return DHQR_OK;
}
// dOut (rows x 128, ld ldo) = dX (rows x 128) * Y, given negY = -Y (128 x 128).
int32_t dhqr_rs_mul_f64(dhqr_ctx *c, const double *dX, int64_t ldx, int64_t rows, const double *dnegY, double *dOut,
                        int64_t ldo) {
  CHECK(check_ctx(c));
  if (rows <= 0) return DHQR_OK;
  if (!dX || !dnegY || !dOut || ldo < rows) return set_err(DHQR_EINVAL, "bad arguments to dhqr_rs_mul_f64");
  CHECK(mul128(c, dX, ldx, rows, dnegY, dOut, ldo));
  LAUNCHCHECK();
  return DHQR_OK;
}
// finish V = tril((P - alpha E) M^{-1}) on the 128 diagonal rows (diagonal owner only)
int32_t dhqr_rs_fix_top_f64(dhqr_ctx *c, double *dVw, int64_t ldv, const double *dalpha128, const double *dnegMinv) {
  CHECK(check_ctx(c));
  hipLaunchKernelGGL(k_recon_fix, dim3(DHQR_NBV * DHQR_NBV / 256), dim3(256), 0, c->stream, dVw, ldv, dalpha128,
                     dnegMinv);
  LAUNCHCHECK();
  return DHQR_OK;
}
int32_t dhqr_rs_write_r_f64(dhqr_ctx *c, double *dPtop, int64_t ldp, const double *dRref) {
  CHECK(check_ctx(c));
  hipLaunchKernelGGL(k_recon_write_r, dim3(DHQR_NBV * DHQR_NBV / 256), dim3(256), 0, c->stream, dPtop, ldp, dRref);
  LAUNCHCHECK();
  return DHQR_OK;
}
// commit one panel's reflectors into the local slab: the diagonal owner writes the lower trapezoid
// (rows >= column) and the reference-format R above it, every other rank copies all of its rows.
int32_t dhqr_rs_commit_f64(dhqr_ctx *c, double *dP, int64_t ldp, int64_t rows, const double *dVw, int64_t ldv,
                           int32_t diag_owner, const double *dRref) {
  CHECK(check_ctx(c));
  if (rows <= 0) return DHQR_OK;
  if (diag_owner) {
    dim3 grid((unsigned)std::min<int64_t>((rows + 255) / 256, 64), DHQR_NBV);
    hipLaunchKernelGGL(k_unpack_v, grid, dim3(256), 0, c->stream, dP, ldp, rows, (int64_t)DHQR_NBV, dVw, ldv);
    hipLaunchKernelGGL(k_recon_write_r, dim3(DHQR_NBV * DHQR_NBV / 256), dim3(256), 0, c->stream, dP, ldp, dRref);
  } else {
    HIPCHECK(hipMemcpy2DAsync(dP, ldp * sizeof(double), dVw, ldv * sizeof(double), rows * sizeof(double), DHQR_NBV,
                              hipMemcpyDeviceToDevice, c->stream));
  }
  LAUNCHCHECK();
  return DHQR_OK;
}
// V operand of an ALREADY factored panel for re-applying Q: the diagonal owner gets its rows with
// the R part zeroed, other ranks a plain copy of their rows.  dVw: ldv x 128.
int32_t dhqr_rs_pack_f64(dhqr_ctx *c, const double *dP, int64_t ldp, int64_t rows, double *dVw, int64_t ldv,
                         int32_t diag_owner) {
  CHECK(check_ctx(c));
  if (rows <= 0) return DHQR_OK;
  if (diag_owner) {
    dim3 grid((unsigned)std::min<int64_t>((ldv + 255) / 256, 64), DHQR_NBV);
    hipLaunchKernelGGL(k_pack_v, grid, dim3(256), 0, c->stream, dP, ldp, rows, (int64_t)DHQR_NBV, dVw, ldv);
  } else {
    HIPCHECK(hipMemcpy2DAsync(dVw, ldv * sizeof(double), dP, ldp * sizeof(double), rows * sizeof(double), DHQR_NBV,
                              hipMemcpyDeviceToDevice, c->stream));
  }
  LAUNCHCHECK();
  return DHQR_OK;
}
int32_t dhqr_rs_build_t_f64(dhqr_ctx *c, const double *dS, int32_t ncols, double *dT, double *dTt) {
  CHECK(check_ctx(c));
  launch_build_t(c, dS, (int)ncols, dT, dTt);
  LAUNCHCHECK();
  return DHQR_OK;
}
// dW1 (128 x ncols, ld 128) = dV' dC over the local rows (split-K partials reduced on the device)
int32_t dhqr_rs_vtc_f64(dhqr_ctx *c, const double *dV, int64_t ldv, const double *dC, int64_t ldc, int64_t rows,
                        int64_t ncols, double *dW1) {
  CHECK(check_ctx(c));
  if (ncols <= 0) return DHQR_OK;
  const int64_t wstride = (int64_t)DHQR_NBV * ncols;
  if (rows <= 0) {
    HIPCHECK(hipMemsetAsync(dW1, 0, (size_t)wstride * sizeof(double), c->stream));
    return DHQR_OK;
  }
  const int64_t ntiles = (ncols + 127) / 128;
  int64_t nsplit, rps;
  pick_split(rows, ntiles, 512, ntiles <= 2 ? 256 : 64, &nsplit, &rps);
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  CHECK(ensure(c, ws.w1, (size_t)nsplit * DHQR_NBV * (size_t)ncols));
  const bool vec = (ldc % 2 == 0) && (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(dC) && aligned16(dV);
  const dim3 gtn((unsigned)ntiles, (unsigned)nsplit);
  if (vec)
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), gtn, dim3(256), 0, c->stream, dV, ldv, dC, ldc, 1, (int64_t)0, rows, ncols,
                       rps, ws.w1.p, (int64_t)DHQR_NBV, wstride);
  else
    hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), gtn, dim3(256), 0, c->stream, dV, ldv, dC, ldc, 1, (int64_t)0, rows, ncols,
                       rps, ws.w1.p, (int64_t)DHQR_NBV, wstride);
  hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)((wstride + 63) / 64)), dim3(256), 0, c->stream,
                     (const double *)ws.w1.p, (int)nsplit, wstride, wstride, dW1);
  LAUNCHCHECK();
  return DHQR_OK;
}
// dW2 (128 x ncols) = op(T)' dW1 with dTop = T (update) or T' (apply Q)
int32_t dhqr_rs_tw_f64(dhqr_ctx *c, const double *dTop, const double *dW1, int64_t ncols, double *dW2) {
  CHECK(check_ctx(c));
  if (ncols <= 0) return DHQR_OK;
  const int64_t ntiles = (ncols + 127) / 128;
  hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, dTop, (int64_t)DHQR_NBV,
                     dW1, (int64_t)DHQR_NBV, 1, (int64_t)0, (int64_t)DHQR_NBV, ncols, (int64_t)DHQR_NBV, dW2,
                     (int64_t)DHQR_NBV, (int64_t)0);
  LAUNCHCHECK();
  return DHQR_OK;
}
// dC (rows x ncols) -= dV (rows x 128) * dW2 (128 x ncols)
int32_t dhqr_rs_vw_f64(dhqr_ctx *c, const double *dV, int64_t ldv, const double *dW2, double *dC, int64_t ldc,
                       int64_t rows, int64_t ncols) {
  CHECK(check_ctx(c));
  if (rows <= 0 || ncols <= 0) return DHQR_OK;
  const bool vec = (ldc % 2 == 0) && (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(dC) && aligned16(dV);
  dim3 grid((unsigned)((rows + 127) / 128), (unsigned)((ncols + 127) / 128));
  if (vec)
    hipLaunchKernelGGL((k_gemm_nn_sub<2, 128>), grid, dim3(256), 0, c->stream, dV, ldv, dW2, (int64_t)DHQR_NBV, dC, ldc,
                       rows, ncols, 0);
  else
    hipLaunchKernelGGL((k_gemm_nn_sub<1, 128>), grid, dim3(256), 0, c->stream, dV, ldv, dW2, (int64_t)DHQR_NBV, dC, ldc,
                       rows, ncols, 0);
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_panel_apply_f64(dhqr_ctx *c, const double *dVT, int64_t rows, double *dC, int64_t ncols,
                             int64_t ldc, int32_t trans) {
  CHECK(check_ctx(c));
  if (!dVT || !aligned16(dVT)) return set_err(DHQR_EINVAL, "dVT must be a 16-byte aligned device buffer");
  if (ncols == 0) return DHQR_OK;
  CHECK(check_mat(dC, rows, ncols, ldc, false));
  return panel_apply(c, dVT, rows, dC, ncols, ldc, trans ? 1 : 0);
}

int32_t dhqr_bench_mfma_f64(dhqr_ctx *c, double *tflops) {
  CHECK(check_ctx(c));
  if (!tflops) return set_err(DHQR_EINVAL, "null output");
  const int nblk = 256 * 8, iters = 4000;
  CHECK(ensure(c, c->scratch, (size_t)nblk * 256 + 4096));
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_mfma_bench, dim3(nblk), dim3(256), 0, c->stream, c->scratch.p, 100);
  HIPCHECK(hipEventRecord(a, c->stream));
  hipLaunchKernelGGL(k_mfma_bench, dim3(nblk), dim3(256), 0, c->stream, c->scratch.p, iters);
  HIPCHECK(hipEventRecord(b, c->stream));
  HIPCHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  const double flops = (double)nblk * 4.0 * (double)iters * 16.0 * 2048.0;
  *tflops = flops / ((double)ms * 1e-3) / 1e12;
  return DHQR_OK;
}

int32_t dhqr_bench_issue_f64(dhqr_ctx *c, int32_t kind, int32_t nblocks, double *cycles_per_instr,
                             double *tflops) {
  CHECK(check_ctx(c));
  if (!cycles_per_instr || !tflops || nblocks <= 0 || nblocks > 4096 || (kind != 0 && kind != 1))
    return set_err(DHQR_EINVAL, "bad arguments");
  const int iters = 2000;
  CHECK(ensure(c, c->scratch, (size_t)nblocks * 256 + 4096 + (size_t)nblocks * 4 + 16));
  double *sink = c->scratch.p;
  long long *cyc = (long long *)(c->scratch.p + (size_t)nblocks * 256 + 4096);
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {  // first pass warms clocks / code
    HIPCHECK(hipEventRecord(a, c->stream));
    if (kind == 0) hipLaunchKernelGGL((k_issue_probe<0>), dim3(nblocks), dim3(256), 0, c->stream, sink, cyc, iters);
    else hipLaunchKernelGGL((k_issue_probe<1>), dim3(nblocks), dim3(256), 0, c->stream, sink, cyc, iters);
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  std::vector<long long> h((size_t)nblocks * 4);
  HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (long long v : h) sum += (double)v;
  *cycles_per_instr = sum / (double)h.size() / ((double)iters * 16.0);
  const double flop_per_instr = kind == 0 ? 2048.0 : 128.0;
  *tflops = (double)nblocks * 4.0 * iters * 16.0 * flop_per_instr / ((double)ms * 1e-3) / 1e12;
  return DHQR_OK;
}

int32_t dhqr_bench_issue2_f64(dhqr_ctx *c, int32_t mode, int32_t threads, int32_t nblocks, double *out4) {
  CHECK(check_ctx(c));
  if (!out4 || nblocks <= 0 || nblocks > 4096 || mode < 0 || mode > 2 || threads % 256 || threads > 1024)
    return set_err(DHQR_EINVAL, "bad arguments");
  const int iters = 1000, wpb = threads / 64;
  CHECK(ensure(c, c->scratch, (size_t)nblocks * threads + 4096 + (size_t)nblocks * wpb + 16));
  double *sink = c->scratch.p;
  long long *cyc = (long long *)(c->scratch.p + (size_t)nblocks * threads + 4096);
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCHECK(hipEventRecord(a, c->stream));
    hipLaunchKernelGGL(k_issue_probe2, dim3(nblocks), dim3(threads), 0, c->stream, sink, cyc, iters, (int)mode);
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  std::vector<long long> h((size_t)nblocks * wpb);
  HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double sm = 0, sv = 0;
  int64_t nm = 0, nv = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const int wave = (int)(i % wpb);
    const bool mf = (mode == 0) || (mode == 2 && wave < 4);
    if (mf) { sm += (double)h[i]; nm++; } else { sv += (double)h[i]; nv++; }
  }
  out4[0] = nm ? sm / nm / (iters * 8.0) : 0.0;          // cycles per MFMA per wave
  out4[1] = nv ? sv / nv / (iters * 8.0 * 16.0) : 0.0;   // cycles per v_fma_f64 per wave
  out4[2] = (double)nm * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
  out4[3] = (double)nv * iters * 8.0 * 16.0 * 128.0 / (ms * 1e-3) / 1e12;
  return DHQR_OK;
}

int32_t dhqr_bench_stream_f64(dhqr_ctx *c, int64_t bytes, double *gbps) {
  CHECK(check_ctx(c));
  if (!gbps || bytes < 4096) return set_err(DHQR_EINVAL, "bad arguments");
  const int64_t n2 = bytes / 16;
  double *x = nullptr, *y = nullptr;
  if (hipMalloc((void **)&x, (size_t)n2 * 16) != hipSuccess || hipMalloc((void **)&y, (size_t)n2 * 16) != hipSuccess) {
    if (x) (void)hipFree(x);
    return set_err(DHQR_ENOMEM, "hipMalloc failed in dhqr_bench_stream_f64");
  }
  (void)hipMemsetAsync(x, 0, (size_t)n2 * 16, c->stream);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const unsigned grid = 256 * 16;
  hipLaunchKernelGGL(k_stream_bench, dim3(grid), dim3(256), 0, c->stream, (const double2 *)x, (double2 *)y, n2);
  (void)hipEventRecord(a, c->stream);
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_stream_bench, dim3(grid), dim3(256), 0, c->stream, (const double2 *)x, (double2 *)y, n2);
  (void)hipEventRecord(b, c->stream);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(x);
  (void)hipFree(y);
  *gbps = 2.0 * (double)n2 * 16.0 * reps / ((double)ms * 1e-3) / 1e9;
  return DHQR_OK;
}

// test hook (not in dhqr.h's stable surface, declared in the test binding only):
// raw MFMA D registers for the documented operand maps
int32_t dhqr_debug_mfma_probe(dhqr_ctx *c, const double *da, const double *db, double *dout) {
  CHECK(check_ctx(c));
  hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, c->stream, da, db, dout);
  LAUNCHCHECK();
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}

}  // extern "C"
