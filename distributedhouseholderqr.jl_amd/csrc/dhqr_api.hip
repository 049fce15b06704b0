// dhqr_api.hip -- host side of libdhqr.so: context, workspaces, panel/trailing-update drivers and
// the extern "C" entry points declared in include/dhqr.h.  gfx950 only; no CPU fallback.
// (The nb = 0 path lives in dhqr_unblocked.hip; what the two units share is in dhqr_internal.h.)
#include <chrono>
#include <mutex>

#include "dhqr_internal.h"
#include "dhqr_complex.h"
#include "dhqr_gemm.h"
#include "dhqr_pack.h"
#include "dhqr_panel.h"
#include "dhqr_recon.h"
#include "dhqr_solve.h"
#include "dhqr_qtb.h"
#include "dhqr_small.h"
#include "dhqr_tsqr.h"

static thread_local char g_err[512] = "";
int32_t set_err(int32_t code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int32_t ensure(dhqr_ctx *c, Buf &b, size_t need) {
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

// DHQR_TUNE="key=value,key=value,...": the size thresholds and grid sizes that only tests and tuning runs change, in ONE
// switch (INTEGRATION.md section 5 lists the keys).  Returns true and *out when `key` is present.
bool tune_get(const char *key, long long *out) {
  const char *e = getenv("DHQR_TUNE");
  if (!e) return false;
  const size_t kl = strlen(key);
  for (const char *p = e; *p;) {
    const char *q = strchr(p, ',');
    const size_t len = q ? (size_t)(q - p) : strlen(p);
    if (len > kl + 1 && strncmp(p, key, kl) == 0 && p[kl] == '=') {
      *out = atoll(p + kl + 1);
      return true;
    }
    if (!q) break;
    p = q + 1;
  }
  return false;
}

// contexts alive per device in this process: kernels whose workgroups wait for each other in both directions (k_qtb_persist)
// are only launched while a context has its device to itself -- two such launches could each hold slots the other needs
static std::atomic<int> g_live_ctx[64];


// ---- profiling: one hipEvent pair per timed launch group, resolved in dhqr_get_stats ---------
int32_t prof_begin(dhqr_ctx *c, int cat) {
  if (!c->profiling) return DHQR_OK;
  if (c->ev_used == c->evs.size()) {
    dhqr_ctx::Ev e;
    HIPCHECK(hipEventCreate(&e.a));
    HIPCHECK(hipEventCreate(&e.b));
    e.cat = cat;
    c->evs.push_back(e);
  }
  c->evs[c->ev_used].cat = cat;
  c->evs[c->ev_used].start_from = -1;
  HIPCHECK(hipEventRecord(c->evs[c->ev_used].a, c->stream));
  return DHQR_OK;
}
int32_t prof_end(dhqr_ctx *c) {
  if (!c->profiling) return DHQR_OK;
  HIPCHECK(hipEventRecord(c->evs[c->ev_used].b, c->stream));
  c->ev_used++;
  return DHQR_OK;
}
// End the running section and begin the next one (category `cat`) at the SAME point of the same stream with ONE event
// record instead of two: every record is a bubble on the stream, and the wide stream carries three back-to-back sections
// per update (V'C | T products | subtraction).
int32_t prof_switch(dhqr_ctx *c, int cat) {
  if (!c->profiling) return DHQR_OK;
  CHECK(prof_end(c));
  if (c->ev_used == c->evs.size()) {
    dhqr_ctx::Ev e;
    HIPCHECK(hipEventCreate(&e.a));
    HIPCHECK(hipEventCreate(&e.b));
    e.cat = cat;
    c->evs.push_back(e);
  }
  c->evs[c->ev_used].cat = cat;
  c->evs[c->ev_used].start_from = (int)c->ev_used - 1;
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
    const int sf = c->evs[i].start_from;
    HIPCHECK(hipEventElapsedTime(&t, sf >= 0 ? c->evs[(size_t)sf].b : c->evs[i].a, c->evs[i].b));
    *ms[c->evs[i].cat] += (double)t;
    *cnt[c->evs[i].cat] += 1;
  }
  c->ev_used = 0;
  return DHQR_OK;
}

// ---- the three single-workgroup dense 128 x 128 kernels of the panel chain (dhqr_recon.h)
static inline void launch_chol_inv(dhqr_ctx *c, const double *G, const double *Rprev, double *Rout, double *negX,
                                   int *flag) {
  hipLaunchKernelGGL(k_chol_inv, dim3(1), dim3(1024), 0, c->stream, G, Rprev, Rout, negX, flag);
}
static inline void launch_recon_top(dhqr_ctx *c, const double *P, int64_t ldp, const double *R, double *alpha,
                                    double *Rref, double *negMinv) {
  hipLaunchKernelGGL(k_recon_top, dim3(1), dim3(1024), 0, c->stream, P, ldp, R, alpha, Rref, negMinv);
}
static inline void launch_build_t(dhqr_ctx *c, const double *S, int ncols, double *T, double *Tt) {
  hipLaunchKernelGGL(k_build_t, dim3(1), dim3(1024), 0, c->stream, S, ncols, T, Tt, 0.0, (int *)nullptr, 0, (double *)nullptr,
                     c->tt_keep);
}

// ---- where a factored panel's GEMM operands live --------------------------------------------------
// V: rows x 128 with leading dimension ldv (R part above the diagonal zeroed, columns >= ncols zero);
// T / Tt: the upper-triangular compact-WY factor and its transpose (128 x 128 each); alpha: 128 doubles
// followed by DHQR_STATW status words (word 0: failure index travelling with a broadcast panel).
// Legacy packed buffer ("VT", dhqr_panel_* entry points):  [ V : ldv x 128 | T | Tt | alpha | status ].
#define DHQR_STATW 16
struct PanelBuf {
  double *V;
  int64_t ldv;
  double *T, *Tt, *alpha;
};
static inline int64_t panel_ldv(int64_t rows) { return (rows + 15) & ~(int64_t)15; }
static inline int64_t panel_tail_elems() { return 2 * (int64_t)DHQR_NBV * DHQR_NBV + DHQR_NBV + DHQR_STATW; }
static inline int64_t panel_elems(int64_t rows) { return panel_ldv(rows) * DHQR_NBV + panel_tail_elems(); }
static inline PanelBuf tail_view(double *V, int64_t ldv, double *tail) {
  PanelBuf pb;
  pb.V = V;
  pb.ldv = ldv;
  pb.T = tail;
  pb.Tt = tail + DHQR_NBV * DHQR_NBV;
  pb.alpha = tail + 2 * DHQR_NBV * DHQR_NBV;
  return pb;
}
static inline PanelBuf vt_view(double *vt, int64_t rows) {
  return tail_view(vt, panel_ldv(rows), vt + panel_ldv(rows) * DHQR_NBV);
}
static inline PanelBuf vt_view(const double *vt, int64_t rows) { return vt_view(const_cast<double *>(vt), rows); }

// (stat, epoch) of a matrix-writing launch: active while an asynchronous factorisation is being enqueued
static inline const int *pred_stat(dhqr_ctx *c) { return c->epoch >= 0 ? c->dstat : nullptr; }

// C -= V W on the MFMA kernel (dhqr_gemm.h); INIT0: C = -V W.
// Narrow products (one or two column tiles) that would not fill the chip with 128-row tiles run with 64-row tiles
// (k_gemm_nn_sub<..., TR = 64>): twice the workgroups, half the K-loop time each -- the lane's critical chain.
template <int KW, bool INIT0 = false>
static void launch_nn_sub(dhqr_ctx *c, bool vec, dim3 grid, const double *V, int64_t ldv, const double *W, int64_t ldw,
                          double *C, int64_t ldc, int64_t rows, int64_t ncols, int swz, bool predicated) {
  const int *st = predicated ? pred_stat(c) : nullptr;
  const int64_t ntiles = (ncols + 127) / 128;
  if (vec && !swz && ntiles <= 2 && ((rows + 127) / 128) * ntiles < 512) {
    const dim3 g64((unsigned)((rows + 63) / 64), (unsigned)ntiles);
    hipLaunchKernelGGL((k_gemm_nn_sub<2, KW, INIT0, false, 64>), g64, dim3(256), 0, c->stream, V, ldv, W, ldw, C, ldc, rows, ncols,
                       0, st, c->epoch);
    return;
  }
  if (vec)
    hipLaunchKernelGGL((k_gemm_nn_sub<2, KW, INIT0>), grid, dim3(256), 0, c->stream, V, ldv, W, ldw, C, ldc, rows, ncols,
                       swz, st, c->epoch);
  else
    hipLaunchKernelGGL((k_gemm_nn_sub<1, KW, INIT0>), grid, dim3(256), 0, c->stream, V, ldv, W, ldw, C, ldc, rows, ncols,
                       swz, st, c->epoch);
}

// Workgroups of a persistent wide launch: one per CU, minus the CUs kept free for the lane / RCCL (ctx->spare_cus).
// r5: on a SMALL trailing matrix (ncols <= tn_spare_cols) the V'C pass also leaves tn_spare CUs to the look-ahead lane:
// k_gemm_tn2's workgroups hold whole CUs for the whole launch, the lane makes no progress under them, and once a wide
// step is no longer than the lane's chain (two panels x 290 us) the step is wide + chain instead of max(wide, chain).
// Measured (profiles/r05_ab_tn_spare.txt): 32 CUs: 8192^2 31.8 -> 30.2 ms, 12288^2 68.5 -> 64.5, 16384^2 128.8 -> 124.3.
static inline int64_t wide_slots(const dhqr_ctx *c, int64_t ncols = -1) {
  int spare = c->spare_cus;
  if (ncols >= 0 && ncols <= c->tn_spare_cols && c->lookahead && !c->spare_cus_set)
    spare = std::max(spare, ncols <= c->tn_spare_cols / 2 ? 2 * c->tn_spare : c->tn_spare);  // (64 below 8192 columns: 8192^2 30.4 -> 29.8 ms)
  return std::max<int64_t>(8, (int64_t)c->ncu - spare);
}

// Split-K factor for k_gemm_tn: `ntiles` column tiles x ns row slabs should fill the 512 resident
// workgroup slots (256 CUs x 2) in whole waves -- 765 workgroups on 512 slots run at 75 %.
// min_rows: smallest row slab of a split.  The latency-critical narrow products of the panel lane (one or two column
// tiles: Gram matrices, the look-ahead update) go down to 64 rows per workgroup: a workgroup's time is its K loop
// (1.7 us of MFMA per 16-row K-tile on one CU), so at 8192 rows 128 workgroups x 4 K-tiles beat 64 x 8.
static void pick_split(int64_t rows, int64_t ntiles, int64_t target_wgs, int64_t max_split,
                       int64_t *nsplit, int64_t *rps, int64_t slots = 512, int64_t min_rows = 128) {
  int64_t cap = std::min<int64_t>(max_split, std::max<int64_t>(1, rows / min_rows));
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

// T / T' of a panel whose V part is already in place (ncols real columns).
static int32_t panel_build_t(dhqr_ctx *c, int64_t rows, int64_t ncols, const PanelBuf &pb, int kw = DHQR_NBV) {
  int64_t nsplit, rps;
  pick_split(rows, 1, 128, 128, &nsplit, &rps);
  CHECK(ensure(c, c->spart, (size_t)nsplit * DHQR_NBV * DHQR_NBV));
  CHECK(ensure(c, c->sfull, (size_t)DHQR_NBV * DHQR_NBV));
  const double *V = pb.V;
  const int64_t ldv = pb.ldv;
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
  launch_build_t(c, c->sfull.p, (int)ncols, pb.T, pb.Tt);
  LAUNCHCHECK();
  return DHQR_OK;
}

// Pack V (R part zeroed) and build T / T' for a factored panel P (rows x ncols, ncols <= 128).
static int32_t panel_pack_and_t(dhqr_ctx *c, const double *P, int64_t rows, int64_t ncols,
                                int64_t ldp, const double *alpha, const PanelBuf &pb) {
  const int64_t npad = panel_ldv(rows);
  CHECK(prof_begin(c, CAT_TBUILD));
  {
    dim3 grid((unsigned)std::min<int64_t>((npad + 255) / 256, 64), DHQR_NBV);
    hipLaunchKernelGGL(k_pack_v, grid, dim3(256), 0, c->stream, P, ldp, rows, ncols, pb.V, pb.ldv, npad);
  }
  CHECK(panel_build_t(c, rows, ncols, pb));
  if (alpha)  // nullptr: the caller already placed alpha in the buffer tail (or does not need it)
    hipLaunchKernelGGL(k_commit_alpha, dim3(1), dim3(DHQR_NBV), 0, c->stream, alpha, (int)ncols, (double *)nullptr,
                       pb.alpha, (const int *)nullptr, 0);
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

// C (rows x ncols) <- (I - V op(T) V') C with op(T) = T' (trans=1) or T (trans=0).
// phase 0: everything; 1: only Y = V' C and its split-K reduction (needs V, not T: the lane's side stream runs it while the
// panel is still being verified); 2: the rest (T product + subtraction) on the Y a phase-1 call left in the SAME workspace.
static int32_t panel_apply(dhqr_ctx *c, const PanelBuf &pb, int64_t rows, double *C, int64_t ncols,
                           int64_t ldc, int trans, int kw = DHQR_NBV, int phase = 0) {
  if (ncols <= 0 || rows <= 0) return DHQR_OK;
  const int64_t ldv = pb.ldv;
  const double *V = pb.V;
  const double *Top = trans ? pb.T : pb.Tt, *TopT = trans ? pb.Tt : pb.T;
  const int64_t ntiles = (ncols + 127) / 128;
  int64_t nsplit, rps;
  // narrow updates (one or two column tiles: the look-ahead lane / a rank's single block) are split
  // over up to 256 row slabs so the latency-critical chain uses the whole chip
  pick_split(rows, ntiles, 512, ntiles <= 2 ? 256 : 64, &nsplit, &rps, 512, ntiles <= 2 ? 64 : 128);
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  CHECK(ensure(c, ws.w1, (size_t)nsplit * DHQR_NBV * (size_t)ncols));
  CHECK(ensure(c, ws.w2, (size_t)DHQR_NBV * (size_t)ncols));
  if (nsplit > 1) CHECK(ensure(c, ws.w1r, (size_t)DHQR_NBV * (size_t)ncols));
  const bool vec = (ldc % 2 == 0) && (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(C) && aligned16(V);
  const int64_t wstride = (int64_t)DHQR_NBV * ncols;

  // one instantiation per reflector-block width (32 / 64 inside a panel, 128 for the trailing update)
#define DHQR_APPLY(KW_)                                                                              \
  do {                                                                                               \
    CHECK(prof_begin(c, CAT_VTA));                                                                   \
    if (phase != 2) {                                                                                \
    if (vec)                                                                                         \
      hipLaunchKernelGGL((k_gemm_tn<2, 1, KW_>), dim3((unsigned)ntiles, (unsigned)nsplit), dim3(256), 0, \
                         c->stream, V, ldv, (const double *)C, ldc, 1, (int64_t)0, rows, ncols, rps,    \
                         ws.w1.p, (int64_t)DHQR_NBV, wstride);                                           \
    else                                                                                             \
      hipLaunchKernelGGL((k_gemm_tn<1, 1, KW_>), dim3((unsigned)ntiles, (unsigned)nsplit), dim3(256), 0, \
                         c->stream, V, ldv, (const double *)C, ldc, 1, (int64_t)0, rows, ncols, rps,    \
                         ws.w1.p, (int64_t)DHQR_NBV, wstride);                                           \
    }                                                                                                \
    CHECK(prof_switch(c, CAT_TW));                                                                              \
    /* (one event ends the previous section and starts this one) */                                                                    \
    const double *w1sum = ws.w1.p;                                                                   \
    if (nsplit > 1) { /* bandwidth-friendly, deterministic split-K reduction */                      \
      if (phase != 2)                                                                                \
      hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)((wstride + 63) / 64)), dim3(256), 0,       \
                         c->stream, (const double *)ws.w1.p, (int)nsplit, wstride, wstride, ws.w1r.p); \
      w1sum = ws.w1r.p;                                                                              \
    }                                                                                                \
    if (phase == 1) { CHECK(prof_end(c)); break; }                                                   \
    if (KW_ == DHQR_NBV && ntiles <= 2) /* the lane's narrow updates: k_tw_fused (dhqr_gemm.h) */    \
      hipLaunchKernelGGL((k_tw_fused<false>), dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, c->stream, w1sum, ncols, \
                         TopT, (const double *)nullptr, (const double *)nullptr, ws.w2.p, (int64_t)DHQR_NBV); \
    else                                                                                             \
    hipLaunchKernelGGL((k_gemm_tn<2, 1, KW_>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Top, \
                       (int64_t)DHQR_NBV, w1sum, (int64_t)DHQR_NBV, 1, (int64_t)0, (int64_t)KW_, ncols,  \
                       (int64_t)KW_, ws.w2.p, (int64_t)DHQR_NBV, (int64_t)0);                            \
    CHECK(prof_switch(c, CAT_AVW));                                                                              \
    /* (one event ends the previous section and starts this one) */                                                                   \
    const int64_t gx_ = (rows + 127) / 128;                                                          \
    const int swz_ = (gx_ >= 16 && ntiles >= 16) ? 1 : 0;                              \
    dim3 grid((unsigned)gx_, (unsigned)ntiles);                                                      \
    if (swz_) grid = dim3((unsigned)((((gx_ + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);      \
    launch_nn_sub<KW_>(c, vec, grid, V, ldv, (const double *)ws.w2.p, (int64_t)DHQR_NBV, C, ldc, rows, ncols, swz_, \
                       true);                                                                        \
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
// Factors the rows x w panel P in place, writes alpha[0:w], and leaves the V, T, T', alpha operands in pb.
// The reference algorithm column by column: used for partial / short panels and as the robust fallback.
static int32_t small_qr_launch(dhqr_ctx *c, int fit, const double *Asrc, int64_t lds, double *Adst, int64_t ldd, int64_t m,
                               int64_t n, double *alpha, unsigned long long *done = nullptr, unsigned long long epoch = 0);
static int32_t factor_panel_v2(dhqr_ctx *c, double *P, int64_t rows, int64_t w, int64_t ldp,
                               double *alpha, const PanelBuf &pb) {
  if (c->short_panel_small && rows <= 256 && rows >= w && (rows <= 224 || w <= 192)) {
    // r6: a panel of at most 256 rows -- the LAST panel of every square factorisation (128 rows) among them -- fits the
    // registers of one compute unit: ONE launch of the small route's kernel (dhqr_small.h, the barrier form: no give-up
    // answer inside a driver) instead of one launch per column (136 x 128: 0.8 ms -> 0.15), then V and T as after any panel.
    CHECK(prof_begin(c, CAT_PANEL));
    const bool was = c->profiling;
    c->profiling = false;
    const int keep = c->small_flags;
    c->small_flags = 0;
    int32_t rc = small_qr_launch(c, rows <= 128 ? 0 : (rows <= 224 ? 1 : 2), P, ldp, P, ldp, rows, w, alpha);
    c->small_flags = keep;
    if (rc == DHQR_OK) rc = panel_pack_and_t(c, P, rows, w, ldp, alpha, pb);
    c->profiling = was;
    CHECK(rc);
    if (c->profiling)
      for (int64_t j = 0; j + 1 < w; ++j) c->st.bytes_panel += 16.0 * (double)(rows - j) * (double)(w - j - 1);
    return prof_end(c);
  }
  const int64_t ldvw = pb.ldv;
  double *vt = pb.V;
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
  const int was_epoch = c->epoch;
  c->profiling = false;  // the nested T builds / GEMMs are accounted to the panel
  c->epoch = -1;         // robust path: runs unconditionally
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMemsetAsync(vt, 0, (size_t)(ldvw * (DHQR_NBV - 1) + panel_ldv(rows)) * sizeof(double), c->stream));
    for (int64_t j0 = 0; j0 < w; j0 += ib) {
      const int ncs = (int)std::min<int64_t>(ib, w - j0);
      const int64_t rows_s = rows - j0;
      double *Ps = P + j0 + j0 * ldp;
      const int nch = (int)((rows_s + PS_RC - 1) / PS_RC);
      const PanelBuf sub = vt_view(c->vts.p, rows_s);
      const int64_t ldvs = sub.ldv;
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
        CHECK(panel_build_t(c, rows_s, ncs, sub, kw));
        CHECK(panel_apply(c, sub, rows_s, P + j0 + (j0 + ncs) * ldp, w - j0 - ncs, ldp, 1, kw));
      }
    }
    {
      dim3 grid((unsigned)std::min<int64_t>((rows + 255) / 256, 64), (unsigned)w);
      hipLaunchKernelGGL(k_unpack_v, grid, dim3(256), 0, c->stream, P, ldp, rows, w, (const double *)vt, ldvw,
                         (const int *)nullptr, 0);
    }
    CHECK(panel_build_t(c, rows, w, pb));
    hipLaunchKernelGGL(k_commit_alpha, dim3(1), dim3(DHQR_NBV), 0, c->stream, (const double *)alpha, (int)w,
                       (double *)nullptr, pb.alpha, (const int *)nullptr, 0);
    LAUNCHCHECK();
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  c->epoch = was_epoch;
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
  const bool vec = (ldx % 2 == 0) && (rows % 2 == 0) && aligned16(X);
  int64_t nsplit, rps;
  if (c->gram_strips && rows >= 1024) {
    // r6: FOUR 32-row strips of the 128 x 128 result (blockIdx.z) over at most 64 row slabs instead of one 128-row tile over
    // up to 256: the same ~256 workgroups and the same K-loop time (the product is MFMA-bound either way), a QUARTER of the
    // split-K partials per element -- at 32768 rows 8 MiB instead of 32 MiB written by the product and read back by the
    // reduction, which is most of what the reduction launch costs on the panel chain.  The four strips of a slab run on
    // one XCD (workgroup ids 64 apart) and share the slab through its L2.
    pick_split(rows, 4, 256, 64, &nsplit, &rps, 256, 64);
    CHECK(ensure(c, c->spart, (size_t)nsplit * DHQR_NBV * DHQR_NBV));
    const dim3 grid(1, (unsigned)nsplit, 4);
    if (vec)
      hipLaunchKernelGGL((k_gemm_tn<2, 1, 32>), grid, dim3(256), 0, c->stream, X, ldx, X, ldx, 1, (int64_t)0, rows,
                         (int64_t)DHQR_NBV, rps, c->spart.p, (int64_t)DHQR_NBV, (int64_t)DHQR_NBV * DHQR_NBV);
    else
      hipLaunchKernelGGL((k_gemm_tn<1, 1, 32>), grid, dim3(256), 0, c->stream, X, ldx, X, ldx, 1, (int64_t)0, rows,
                         (int64_t)DHQR_NBV, rps, c->spart.p, (int64_t)DHQR_NBV, (int64_t)DHQR_NBV * DHQR_NBV);
  } else {
    pick_split(rows, 1, 512, 256, &nsplit, &rps, 512, 64);
    CHECK(ensure(c, c->spart, (size_t)nsplit * DHQR_NBV * DHQR_NBV));
    if (vec)
      hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, X, ldx, X, ldx,
                         1, (int64_t)0, rows, (int64_t)DHQR_NBV, rps, c->spart.p, (int64_t)DHQR_NBV,
                         (int64_t)DHQR_NBV * DHQR_NBV);
    else
      hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, X, ldx, X, ldx,
                         1, (int64_t)0, rows, (int64_t)DHQR_NBV, rps, c->spart.p, (int64_t)DHQR_NBV,
                         (int64_t)DHQR_NBV * DHQR_NBV);
  }
  hipLaunchKernelGGL(k_reduce_splits, dim3(DHQR_NBV * DHQR_NBV / 64), dim3(256), 0, c->stream,
                     (const double *)c->spart.p, (int)nsplit, (int64_t)DHQR_NBV * DHQR_NBV,
                     (int64_t)DHQR_NBV * DHQR_NBV, out);
  return DHQR_OK;
}
// out (rows x 128, ld ldo) = X (rows x 128, ld ldx) * Y with negY = -Y given (128 x 128, ld 128)
static int32_t mul128(dhqr_ctx *c, const double *X, int64_t ldx, int64_t rows, const double *negY, double *out,
                      int64_t ldo, const double *fix_alpha = nullptr) {
  const bool vec = (ldx % 2 == 0) && (ldo % 2 == 0) && (rows % 2 == 0) && aligned16(X) && aligned16(out);
  dim3 grid((unsigned)((rows + 127) / 128), 1);
  if (fix_alpha) {  // the panel's V = tril((X - alpha E) Y) in the product's epilogue (k_gemm_nn_vfix, dhqr_gemm.h)
    if (vec && ((rows + 127) / 128) < 512)
      hipLaunchKernelGGL((k_gemm_nn_vfix<2, 64>), dim3((unsigned)((rows + 63) / 64), 1), dim3(256), 0, c->stream, X, ldx, negY,
                         (int64_t)DHQR_NBV, out, ldo, rows, (int64_t)DHQR_NBV, fix_alpha);
    else if (vec)
      hipLaunchKernelGGL((k_gemm_nn_vfix<2, 128>), grid, dim3(256), 0, c->stream, X, ldx, negY, (int64_t)DHQR_NBV, out, ldo, rows,
                         (int64_t)DHQR_NBV, fix_alpha);
    else
      hipLaunchKernelGGL((k_gemm_nn_vfix<1, 128>), grid, dim3(256), 0, c->stream, X, ldx, negY, (int64_t)DHQR_NBV, out, ldo, rows,
                         (int64_t)DHQR_NBV, fix_alpha);
    return DHQR_OK;
  }
  launch_nn_sub<128, true>(c, vec, grid, X, ldx, negY, (int64_t)DHQR_NBV, out, ldo, rows, (int64_t)DHQR_NBV, 0, false);
  return DHQR_OK;
}

// device status block: stat[0] = first failed panel (INT_MAX none), stat[1] = breakdown flag of the panel in flight
__global__ void k_set_status(int *stat, int v0) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    stat[0] = v0;
    stat[1] = 0;
  }
}
static int32_t status_reset(dhqr_ctx *c) {
  hipLaunchKernelGGL(k_set_status, dim3(1), dim3(64), 0, c->stream, c->dstat, INT_MAX);
  LAUNCHCHECK();
  return DHQR_OK;
}
// host copy of stat[0] (synchronises c->stream)
static int32_t pipe_error_report(dhqr_ctx *c, int e);
static int32_t solve_pipelined(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha, double *db,
                               bool allow_persist);
// (the same round trip brings the pipeline error word of dhqr_common.h: a hand-over wait that expired inside any of the
// pass's launches is reported here, by every driver that reads its status once per pass)
static int32_t status_read(dhqr_ctx *c, int *first_failed) {
  HIPCHECK(hipMemcpyAsync(c->hflag, c->dstat, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (c->zflags)
    HIPCHECK(hipMemcpyAsync(c->hflag + 2, c->zflags + DHQR_PIPE_ERR_OFFSET, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  else
    c->hflag[2] = 0;
  HIPCHECK(hipStreamSynchronize(c->stream));
  *first_failed = c->hflag[0];
  if (c->hflag[2] != 0) return pipe_error_report(c, c->hflag[2]);
  return DHQR_OK;
}

// ---- TSQR tree (dhqr_tsqr.h), all launches on c->stream ------------------------------------------------------------
// Pair levels over `count` stacked 128 x 128 R factors.  cnt[0] = count, cnt[l] = ceil(cnt[l-1] / 2) down to 1; the
// reflectors of level l >= 1 live at Ybase + yoff[l] node blocks (256 x 128 each) when Ybase != nullptr.
struct TsqrLevels {
  std::vector<int64_t> cnt, yoff;
  explicit TsqrLevels(int64_t count) {
    cnt.push_back(count);
    yoff.push_back(0);
    int64_t off = 0;
    while (cnt.back() > 1) {
      const int64_t n = (cnt.back() + 1) / 2;
      yoff.push_back(off);
      off += n;
      cnt.push_back(n);
    }
  }
  int L() const { return (int)cnt.size() - 1; }
  static int64_t node_blocks(int64_t count) { return count + 64; }  // upper bound of sum_l cnt[l], l >= 1
};
static const size_t TSQR_NN = (size_t)DHQR_NBV * DHQR_NBV;
// bottom-up: Rroot <- R of the stacked blocks Rin[0 .. count); ping / pong: (count + 1) / 2 blocks each
static int32_t tsqr_pairs_up(dhqr_ctx *c, const double *Rin, int64_t count, double *Rroot, double *Ybase, double *ping,
                             double *pong) {
  if (count <= 0) {
    HIPCHECK(hipMemsetAsync(Rroot, 0, TSQR_NN * sizeof(double), c->stream));
    return DHQR_OK;
  }
  if (count == 1) {
    HIPCHECK(hipMemcpyAsync(Rroot, Rin, TSQR_NN * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return DHQR_OK;
  }
  const TsqrLevels lv(count);
  const double *in = Rin;
  double *bufs[2] = {ping, pong};
  for (int l = 1; l <= lv.L(); ++l) {
    double *out = (l == lv.L()) ? Rroot : bufs[l & 1];
    hipLaunchKernelGGL(k_tsqr_node, dim3((unsigned)lv.cnt[l]), dim3(1024), 0, c->stream, (const double *)nullptr, (int64_t)0,
                       (int64_t)0, in, (int)lv.cnt[l - 1], out, Ybase ? Ybase + (size_t)lv.yoff[l] * 2 * TSQR_NN : nullptr,
                       (int64_t)0);
    in = out;
  }
  LAUNCHCHECK();
  return DHQR_OK;
}
// top-down: *Cleaf <- the `count` C blocks of the bottom level, starting from Croot (nullptr: identity) at the root;
// ping / pong: `count` blocks each
static int32_t tsqr_pairs_down(dhqr_ctx *c, int64_t count, const double *Ybase, const double *Croot, double *ping,
                               double *pong, const double **Cleaf) {
  if (count <= 1) {
    if (Croot == nullptr) {
      hipLaunchKernelGGL(k_tsqr_identity, dim3((unsigned)(TSQR_NN / 256)), dim3(256), 0, c->stream, ping);
      LAUNCHCHECK();
      *Cleaf = ping;
    } else {
      *Cleaf = Croot;
    }
    return DHQR_OK;
  }
  const TsqrLevels lv(count);
  const double *in = Croot;
  double *bufs[2] = {ping, pong};
  for (int l = lv.L(); l >= 1; --l) {
    double *out = bufs[l & 1];
    hipLaunchKernelGGL(k_tsqr_apply, dim3((unsigned)lv.cnt[l]), dim3(1024), 0, c->stream, in,
                       Ybase + (size_t)lv.yoff[l] * 2 * TSQR_NN, (int64_t)0, 0, (int)lv.cnt[l - 1], out, (double *)nullptr,
                       (int64_t)0);
    in = out;
  }
  LAUNCHCHECK();
  *Cleaf = in;
  return DHQR_OK;
}
// Workspace of a local tree over `rows` rows inside c->tsq (after `reserve` doubles the caller keeps for itself).
struct TsqrLocal {
  int64_t rows = 0, nleaf = 0, ldy = 0;
  double *R0 = nullptr, *ping = nullptr, *pong = nullptr, *Yl = nullptr, *Yp = nullptr;
  static size_t elems(int64_t rows) {
    const size_t nl = (size_t)std::max<int64_t>(1, (rows + TSQR_LEAF - 1) / TSQR_LEAF);
    return (3 * nl + 2) * TSQR_NN + nl * 2 * TSQR_NN + (size_t)TsqrLevels::node_blocks((int64_t)nl) * 2 * TSQR_NN;
  }
  void place(double *base, int64_t rows_) {
    rows = rows_;
    nleaf = std::max<int64_t>(1, (rows + TSQR_LEAF - 1) / TSQR_LEAF);
    ldy = nleaf * TSQR_LEAF;
    R0 = base;
    ping = R0 + (size_t)nleaf * TSQR_NN;
    pong = ping + (size_t)(nleaf + 1) * TSQR_NN;
    Yl = pong + (size_t)(nleaf + 1) * TSQR_NN;
    Yp = Yl + (size_t)nleaf * 2 * TSQR_NN;
  }
};
// up: Rroot <- R of the local rows (keep: reflectors stored for tsqr_local_down)
static int32_t tsqr_local_up(dhqr_ctx *c, const TsqrLocal &t, const double *P, int64_t ldp, double *Rroot, bool keep) {
  if (t.rows <= 0) {
    HIPCHECK(hipMemsetAsync(Rroot, 0, TSQR_NN * sizeof(double), c->stream));
    if (keep) HIPCHECK(hipMemsetAsync(t.Yl, 0, (size_t)t.nleaf * 2 * TSQR_NN * sizeof(double), c->stream));
    return DHQR_OK;
  }
  double *lvl0 = (t.nleaf == 1) ? Rroot : t.R0;
  hipLaunchKernelGGL(k_tsqr_node, dim3((unsigned)t.nleaf), dim3(1024), 0, c->stream, P, ldp, t.rows, (const double *)nullptr, 0,
                     lvl0, keep ? t.Yl : (double *)nullptr, t.ldy);
  LAUNCHCHECK();
  if (t.nleaf == 1) return DHQR_OK;
  return tsqr_pairs_up(c, lvl0, t.nleaf, Rroot, keep ? t.Yp : nullptr, t.ping, t.pong);
}
// down: the local rows of the explicit Q -> t.Yl (leading dimension t.ldy), from Croot (nullptr: identity)
static int32_t tsqr_local_down(dhqr_ctx *c, const TsqrLocal &t, const double *Croot) {
  if (t.rows <= 0) return DHQR_OK;
  const double *Cleaf = nullptr;
  CHECK(tsqr_pairs_down(c, t.nleaf, t.Yp, Croot, t.ping, t.pong, &Cleaf));
  hipLaunchKernelGGL(k_tsqr_apply, dim3((unsigned)t.nleaf), dim3(1024), 0, c->stream, Cleaf, (const double *)t.Yl, t.ldy, 1, 0,
                     (double *)nullptr, t.Yl, t.ldy);
  LAUNCHCHECK();
  return DHQR_OK;
}
// R only (dhqr_tsqr_r_f64)
static int32_t tsqr_local_r(dhqr_ctx *c, const double *P, int64_t ldp, int64_t rows, double *Rout) {
  CHECK(ensure(c, c->tsq, TsqrLocal::elems(rows)));
  TsqrLocal t;
  t.place(c->tsq.p, rows);
  return tsqr_local_up(c, t, P, ldp, Rout, false);
}

// Enqueue the R-first factorisation of a full-width panel (w == 128, rows >= 256) WITHOUT waiting for its
// verification: nothing is written to P, alpha or pb.T/Tt/alpha unless the panel is accepted on the device
// (k_build_t); once a panel has failed every later commit / trailing update with epoch >= its index is a
// no-op, and the driver resumes from it with factor_panel_sync after its single final synchronisation.
// The 128 x 128 scratch matrices of a panel (R1, -R1^{-1}, R, Rref, -M^{-1}, G | alpha_tmp): TWO sets, by panel parity -- a
// panel's commit (reads Rref, alpha_tmp) may run on another stream while the next panel's chain fills the other set.
static inline size_t panel_rbuf_elems() { return 6 * (size_t)DHQR_NBV * DHQR_NBV + 1024; }
static inline double *panel_rbuf(dhqr_ctx *c, int panel_idx) { return c->rbuf.p + (size_t)(panel_idx & 1) * panel_rbuf_elems(); }
// commit of an accepted panel (device-side predicate): reflectors, R, alpha in one launch, on `stream`
// part: 0 everything; 1 only alpha -> the panel buffer's tail (what a broadcast of the buffer carries to the peers: must be
// in place BEFORE the broadcast); 2 everything else (reflectors and R -> the matrix, alpha -> the caller's vector)
static int32_t panel_commit_enqueue(dhqr_ctx *c, hipStream_t stream, double *P, int64_t rows, int64_t ldp, double *alpha,
                                    const PanelBuf &pb, int panel_idx, int part = 0) {
  const size_t NN = (size_t)DHQR_NBV * DHQR_NBV;
  const double *Rref = panel_rbuf(c, panel_idx) + 3 * NN, *altmp = panel_rbuf(c, panel_idx) + 6 * NN;
  if (part == 1) {
    hipLaunchKernelGGL(k_commit_alpha, dim3(1), dim3(DHQR_NBV), 0, stream, altmp, (int)DHQR_NBV, (double *)nullptr, pb.alpha,
                       (const int *)c->dstat, panel_idx);
  } else {
    dim3 grid((unsigned)std::min<int64_t>((rows + 255) / 256, 64), DHQR_NBV);
    hipLaunchKernelGGL(k_commit_panel, grid, dim3(256), 0, stream, P, ldp, rows, (const double *)pb.V, pb.ldv, Rref, altmp, alpha,
                       part == 2 ? (double *)nullptr : pb.alpha, (const int *)c->dstat, panel_idx);
  }
  LAUNCHCHECK();
  return DHQR_OK;
}
// ev_t != nullptr: the commit is left to the caller (panel_commit_enqueue on a stream that waits for ev_t, recorded here
// behind k_build_t; the caller records c->ev_commit[panel_idx & 1] behind the commit) -- r6: 12 us less on the lane per panel.
static int32_t panel_fast_enqueue(dhqr_ctx *c, double *P, int64_t rows, int64_t ldp, double *alpha, const PanelBuf &pb,
                                  int passes, int panel_idx, hipEvent_t ev_v = nullptr, hipEvent_t ev_t = nullptr) {
  const int64_t ldv = pb.ldv;
  const size_t NN = (size_t)DHQR_NBV * DHQR_NBV;
  CHECK(ensure(c, c->rbuf, 2 * panel_rbuf_elems()));
  if (passes == 2) CHECK(ensure(c, c->vts, (size_t)panel_elems(rows)));
  if (passes == 3) CHECK(ensure(c, c->tsq, TsqrLocal::elems(rows)));
  CHECK(ensure(c, c->sfull, NN));
  // this panel's scratch set was last read by the commit of panel_idx - 2 (a no-op wait unless that commit left the lane)
  if (c->ev_commit[panel_idx & 1]) HIPCHECK(hipStreamWaitEvent(c->stream, c->ev_commit[panel_idx & 1], 0));
  double *R1 = panel_rbuf(c, panel_idx), *negR1inv = R1 + NN, *Rf = R1 + 2 * NN, *Rref = R1 + 3 * NN, *negMinv = R1 + 4 * NN;
  double *G = R1 + 5 * NN, *altmp = R1 + 6 * NN;
  int *bflag = c->dstat + 1;
  CHECK(prof_begin(c, CAT_PANEL));
  const bool was = c->profiling;
  c->profiling = false;
  auto body = [&]() -> int32_t {
    const double *X = P;  // what the reflectors are reconstructed from: the panel, or its explicit Q factor
    int64_t ldx = ldp;
    if (passes == 3) {  // TSQR-HR (dhqr_tsqr.h): R_t and the explicit Q from the tree, replay of Q with R(Q) = I
      TsqrLocal t;
      t.place(c->tsq.p, rows);
      CHECK(tsqr_local_up(c, t, P, ldp, Rf, true));
      CHECK(tsqr_local_down(c, t, nullptr));
      hipLaunchKernelGGL(k_tsqr_identity, dim3((unsigned)(NN / 256)), dim3(256), 0, c->stream, R1);
      X = t.Yl;
      ldx = t.ldy;
      launch_recon_top(c, X, ldx, R1, G, Rref, negMinv);                            // G[0..128) = alpha(Q) = +-1
      hipLaunchKernelGGL(k_tsqr_sign_cols, dim3((unsigned)(NN / 256)), dim3(256), 0, c->stream, negMinv, (const double *)Rf);
    } else if (passes == 2) {
      CHECK(gram128(c, P, ldp, rows, G));                                          // G  = P'P
      double *Q1 = c->vts.p;  // rows x 128 scratch
      const int64_t ldq = panel_ldv(rows);
      launch_chol_inv(c, G, nullptr, R1, negR1inv, bflag);                          // R1, -R1^{-1}
      CHECK(mul128(c, P, ldp, rows, negR1inv, Q1, ldq));                           // Q1 = P R1^{-1}
      CHECK(gram128(c, Q1, ldq, rows, G));                                         // G2 = Q1'Q1
      launch_chol_inv(c, G, R1, Rf, nullptr, bflag);                                // R  = chol(G2) R1
      launch_recon_top(c, P, ldp, Rf, altmp, Rref, negMinv);                        // alpha, R_ref, -M^{-1}
    } else {  // R = chol(P'P), replay, -M^{-1} in one launch
      CHECK(gram128(c, P, ldp, rows, G));
      hipLaunchKernelGGL((k_panel_top<false>), dim3(1), dim3(1024), 0, c->stream, (const double *)G, (const double *)P, ldp, altmp,
                         Rref, negMinv, bflag);
    }
    if (c->fuse_fix) {
      CHECK(mul128(c, X, ldx, rows, negMinv, pb.V, ldv, passes == 3 ? G : altmp));  // Vw = tril((X - aE) M^{-1}), one launch
    } else {
      CHECK(mul128(c, X, ldx, rows, negMinv, pb.V, ldv));                            // Vw = X M^{-1}
      hipLaunchKernelGGL(k_recon_fix, dim3(NN / 256), dim3(256), 0, c->stream, pb.V, ldv,
                         (const double *)(passes == 3 ? G : altmp), (const double *)negMinv);  // Vw = tril((X - aE) M^{-1})
    }
    // pb.V holds the reflectors from here on (T, the verdict and the commit follow): what needs V alone may start
    if (ev_v) HIPCHECK(hipEventRecord(ev_v, c->stream));
    if (passes == 3)  // R = D R_t, alpha = diag(R)
      hipLaunchKernelGGL(k_tsqr_final_r, dim3(NN / 256), dim3(256), 0, c->stream, (const double *)Rf, (const double *)G, Rref,
                         altmp);
    CHECK(gram128(c, pb.V, ldv, rows, c->sfull.p));                                // S = V'V
    // T from S, fused with the acceptance decision (before the predicated commits)
    hipLaunchKernelGGL(k_build_t, dim3(1), dim3(1024), 0, c->stream, (const double *)c->sfull.p, (int)DHQR_NBV, pb.T, pb.Tt,
                       c->recon_tol, c->dstat, panel_idx, pb.alpha + DHQR_NBV, c->tt_keep);
    // commit (device-side predicate): reflectors, R, alpha in one launch; T is only ever read by accepted consumers
    if (ev_t) HIPCHECK(hipEventRecord(ev_t, c->stream));
    else CHECK(panel_commit_enqueue(c, c->stream, P, rows, ldp, alpha, pb, panel_idx));
    LAUNCHCHECK();
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  if (c->profiling)
    for (int64_t j = 0; j + 1 < DHQR_NBV; ++j) c->st.bytes_panel += 16.0 * (double)(rows - j) * (double)(DHQR_NBV - j - 1);
  CHECK(prof_end(c));
  return DHQR_OK;
}
static inline bool panel_fast_eligible(dhqr_ctx *c, int64_t rows, int64_t w) {
  return c->panel_impl == 3 && w == DHQR_NBV && rows >= 2 * DHQR_NBV;
}

// Robust, host-verified panel factorisation: fast path (then CholeskyQR2) with the verification read back,
// finally the column-by-column kernels on the untouched P.  One or two stream synchronisations; used by the
// simple (no look-ahead) driver, the dhqr_panel_* entry points and when an asynchronous run resumes.
static int32_t factor_panel_sync(dhqr_ctx *c, double *P, int64_t rows, int64_t w, int64_t ldp, double *alpha,
                                 const PanelBuf &pb) {
  if (panel_fast_eligible(c, rows, w)) {
    // ladder: Gram/Cholesky -> CholeskyQR2 -> TSQR tree (each verified on the device, nothing written on rejection)
    const int last_rung = std::max(c->cholqr_passes, c->tsqr_rung == 1 ? 3 : 2);
    for (int passes = c->cholqr_passes; passes <= last_rung; ++passes) {
      CHECK(status_reset(c));
      CHECK(panel_fast_enqueue(c, P, rows, ldp, alpha, pb, passes, 0));
      int failed = 0;
      CHECK(status_read(c, &failed));
      if (failed == INT_MAX) {
        c->n_fast++;
        if (passes == 3) c->n_tsqr++;
        return DHQR_OK;
      }
    }
    CHECK(status_reset(c));
    c->n_fallback++;
  }
  // (r6: a panel the R-first path does not take -- fewer than 128 columns -- of 257 .. 4608 rows goes through the K-reflector
  // passes of the unblocked path, ~w / 6 launches, instead of one launch per column: 2200 x 2000 5.34 -> 5.00 ms, 3000 x 300
  // 1.04 -> 0.97; above that height the per-column kernels win again, 8192 x 1000 3.24 against 3.42: profiles/r06_mid_sizes.txt)
  if (c->panel_impl >= 2 && !(c->partial_unblocked && rows > 256 && rows <= c->partial_unblocked_max_rows))
    return factor_panel_v2(c, P, rows, w, ldp, alpha, pb);
  CHECK(factor_unblocked_cols(c, P, rows, w, ldp, alpha, CAT_PANEL));
  return panel_pack_and_t(c, P, rows, w, ldp, alpha, pb);
}

// ---- two-panel trailing update (K = 256) ------------------------------------------------------
// C <- (I - V_b T_b' V_b')(I - V_a T_a' V_a') C in ONE pass over C for the subtraction:
//   W_a = T_a' (V_a' C),  W_b = T_b' (V_b' C - (V_b' V_a) W_a),  C -= [V_a V_b] [W_a; W_b].
// Vp = [V_a | V_b] (ldv x 256; V_b shifted down by 128 rows, zeros above), rows = rows of panel a.
// Halves the C read+write traffic of the NN GEMM per flop (0.125 -> 0.094 B/flop through the CU
// memory pipe), which is what bounds k_gemm_nn_sub; the TN pass (k_gemm_tn2) reads C once for both panels.
// ar / rows_b_in (row split, dhqr_rowsplit.h): Y is summed over the ranks of `ar` before T is applied (the "all-reduce of
// the cross-partition partial dots"), and V_b may start fewer than 128 rows below V_a on a rank that does not hold
// the pair's diagonal blocks (statistics only).
static int32_t comm_allreduce_sum(dhqr_comm *cm, double *dbuf, int64_t count, hipStream_t stream);

// Column chunks of a WIDE subtraction launch.  The look-ahead lane's single-workgroup kernels (k_panel_top, k_build_t: 1024
// threads, 135-141 KB of LDS) need an EMPTY CU; while a subtraction launch runs every CU holds two of its workgroups and a
// retiring one is replaced at once, so the lane stands still at its first such kernel until the launch has drained and
// finishes its chain afterwards, with the wide stream waiting (profiles/r03_panel_server.txt: ~1 ms per quad step at
// 32768^2, ~2 ms per pair step of the 262144 x 4096 row split).  Between two CHUNKS the CUs drain and the waiting kernel
// gets one: each boundary lets the lane past one more of its whole-CU kernels, for the price of one launch tail.
// Measured (profiles/r03_nn_chunks.txt): 32768^2 847.3 -> 841.5 ms, 16384^2 140.1 -> 138.5 ms, 262144 x 4096 row split
// 178.9 -> 174.0 ms (row chunks: 30 column tiles, 2048 row tiles).  At most nn_split chunks of at least nn_chunk_tiles (48) tiles each.
// Only launches of few column tiles and many row tiles are chunked, by ROWS (the row split, a rank's local block at P > 1);
// column chunks of the square single-GPU case gained 0.4 %, within the box-to-box spread (round 4), and were deleted.
static inline int64_t nn_row_chunks(const dhqr_ctx *c, int64_t row_tiles) {
  if (c->nn_split <= 1 || c->cur_ws != 0) return 1;
  return std::max<int64_t>(1, std::min<int64_t>(c->nn_split, row_tiles / c->nn_chunk_tiles));
}

// Y (256 x ncols, ld 256) = [V_a V_b]' C for ONE or TWO column tiles, in workgroups that fit beside a running wide
// subtraction.  k_gemm_tn2 needs a whole CU per workgroup (8 waves x 256 VGPRs, 110 KB of LDS): beside k_gemm_nn_quad,
// whose 256-thread workgroups sit two to a CU and are replaced one at a time, a launch of ~250 of them on the look-ahead
// lane found no CU until the subtraction had drained -- the r4 per-launch trace shows the lane's narrow V'C "running"
// 12.2 ms beside a 12.7 ms subtraction, and the ~0.9 ms of panel chain behind it running after the subtraction with the
// wide stream idle (45 x ~0.9 ms of 840 ms at 32768^2).  k_gemm_tn (256 threads, 220 VGPRs, 72 KB: the size of ONE
// subtraction workgroup) takes the slots as they free up and runs at idle speed there (profiles/r04_thin_lane_probe.txt:
// a 234-workgroup Gram launch 42 us beside the subtraction, 39 us alone).  Here: blockIdx.z = 0 / 1 = V_a / V_b, i.e. twice
// the workgroups of half the size, C read once more (a narrow update's C is a few MB).  part: nsplit x 256 x ncols.
static int32_t narrow_vtc(dhqr_ctx *c, const double *Vp, int64_t ldv, int64_t rows, const double *C, int64_t ldc,
                          int64_t ncols, bool vec, Buf &part, double *Y) {
  const int64_t ntiles = (ncols + 127) / 128, ld2 = 2 * DHQR_NBV, wstride = ld2 * ncols;
  int64_t nsplit, rps;
  pick_split(rows, 2 * ntiles, 512, 256 / ntiles, &nsplit, &rps, 512, 64);
  CHECK(ensure(c, part, (size_t)nsplit * (size_t)wstride));
  const dim3 grid((unsigned)ntiles, (unsigned)nsplit, 2);
  if (vec)
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), grid, dim3(256), 0, c->stream, Vp, ldv, C, ldc, 1, (int64_t)0, rows, ncols, rps, part.p,
                       ld2, wstride);
  else
    hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), grid, dim3(256), 0, c->stream, Vp, ldv, C, ldc, 1, (int64_t)0, rows, ncols, rps, part.p,
                       ld2, wstride);
  hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)((wstride + 63) / 64)), dim3(256), 0, c->stream, (const double *)part.p, (int)nsplit,
                     wstride, wstride, Y);
  return DHQR_OK;
}

// Y (256 x ncols, ld 256) = [V_a V_b]' C: k_gemm_tn2 (ONE pass over C for both panels, split-K over row slabs into
// ws.w1) + the deterministic split-K reduction.
static int32_t pair_vtc(dhqr_ctx *c, const double *Vp, int64_t ldv, int64_t rows, const double *C, int64_t ldc,
                        int64_t ncols, bool vec, double *Y) {
  const int64_t ntiles = (ncols + 127) / 128;
  if (ntiles <= 2) return narrow_vtc(c, Vp, ldv, rows, C, ldc, ncols, vec, c->ws[c->cur_ws].w1, Y);
  int64_t nsplit, rps;
  // k_gemm_tn2 workgroups have 512 threads and 110 KB of LDS: one per CU, 256 resident
  const int64_t slots = wide_slots(c, ncols);
  pick_split(rows, ntiles, slots, ntiles <= 2 ? 256 : 64, &nsplit, &rps, slots, ntiles <= 2 ? 64 : 128);
  if (ntiles >= c->tn_model_min_tiles) {
    // Wide launches: k_gemm_tn2 runs ONE workgroup per CU, all of the same size, so a launch takes
    // ceil(ntiles * ns / 256) rounds of rows / ns rows each -- 224 column tiles at ns = 1..4 idle an eighth of the chip,
    // at ns = 8 they are exactly seven full rounds.  Estimated time in "rows of one workgroup" (2.85e-7 s each at the
    // kernel's per-CU rate): rounds * slab + the split-K partials written and read back (ns * ncols * 4 KiB at ~3 TB/s)
    // + ~10 us of prologue / epilogue per round; smallest estimate wins (in situ the per-launch rate varied between
    // 53.7 and 63.6 TFLOP/s with the number of column tiles, profiles/r02_ab_gemm_variants.txt section 7).
    const int64_t cap = std::min<int64_t>(64, std::max<int64_t>(1, rows / 512));
    double best = 1e300;
    int64_t bns = 1;
    for (int64_t ns = 1; ns <= cap; ++ns) {
      if (ns > 1 && ntiles * ns > 3072) break;  // partial buffer: at most 12 rounds (cs_prepare / rs_prepare size it for that)
      int64_t r = (rows + ns - 1) / ns;
      r = (r + G_KT - 1) / G_KT * G_KT;
      const int64_t rounds = (ntiles * ns + slots - 1) / slots;
      const double est = (double)rounds * (double)r + (double)ns * (double)ncols * 4.79e-3 + 40.0 * (double)rounds;
      if (est < best) { best = est; bns = ns; }
    }
    int64_t r = (rows + bns - 1) / bns;
    r = (r + G_KT - 1) / G_KT * G_KT;
    rps = r;
    nsplit = std::max<int64_t>(1, (rows + r - 1) / r);
  }
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  const int64_t ld2 = 2 * DHQR_NBV, wstride = ld2 * ncols;
  if (ntiles >= c->tn_model_min_tiles && ntiles <= 1024 && rows >= 1024) {
    // stream-K (k_gemm_tn2<.., true>): 128-row fine units numbered tile-major, a contiguous range per workgroup
    const int64_t FU = 128, S = (rows + FU - 1) / FU, U = ntiles * S;
    const int64_t G = std::min<int64_t>(U, slots), q = (U + G - 1) / G, Gq = (U + q - 1) / q;
    // row groups (dhqr_gemm.h: tn2_sk_group_of): XCD x works inside row range x % R, so that its slice of [V_a V_b] stays in
    // its L2.  DHQR_TUNE tn2_rgroups = 1 / 2 / 4 / 8, 0 = by height.
    int R = c->tn2_rgroups;
    if (R == 0) R = rows >= c->tn2_rg8_rows ? 8 : rows >= c->tn2_rg4_rows ? 4 : rows >= c->tn2_rg2_rows ? 2 : 1;
    if (G < 64 || S < 4 * R) R = 1;
    if (R > 1) {
      int P = 0;
      for (int g = 0; g < R; ++g) P = std::max(P, tn2_sk_pieces(tn2_sk_group_of(g, R, G, S, ntiles, q)));
      CHECK(ensure(c, ws.w1, (size_t)R * (size_t)P * (size_t)wstride));
      if (vec)
        hipLaunchKernelGGL((k_gemm_tn2<2, true>), dim3((unsigned)G), dim3(512), 0, c->stream, Vp, ldv, C, ldc, rows, ncols, FU, ws.w1.p, wstride, q, R, P);
      else
        hipLaunchKernelGGL((k_gemm_tn2<1, true>), dim3((unsigned)G), dim3(512), 0, c->stream, Vp, ldv, C, ldc, rows, ncols, FU, ws.w1.p, wstride, q, R, P);
      hipLaunchKernelGGL(k_reduce_pieces, dim3((unsigned)((wstride + 255) / 256)), dim3(256), 0, c->stream, (const double *)ws.w1.p, S, q,
                         wstride, wstride, Y, R, P, G, ntiles);
      return DHQR_OK;
    }
    const int64_t pieces = (q >= S) ? 2 : (S + q - 1) / q + 1;
    CHECK(ensure(c, ws.w1, (size_t)pieces * (size_t)wstride));
    if (vec)
      hipLaunchKernelGGL((k_gemm_tn2<2, true>), dim3((unsigned)Gq), dim3(512), 0, c->stream, Vp, ldv, C, ldc, rows, ncols, FU, ws.w1.p, wstride, q, 1, 0);
    else
      hipLaunchKernelGGL((k_gemm_tn2<1, true>), dim3((unsigned)Gq), dim3(512), 0, c->stream, Vp, ldv, C, ldc, rows, ncols, FU, ws.w1.p, wstride, q, 1, 0);
    hipLaunchKernelGGL(k_reduce_pieces, dim3((unsigned)((wstride + 255) / 256)), dim3(256), 0, c->stream, (const double *)ws.w1.p, S, q,
                       wstride, wstride, Y, 1, 0, Gq, ntiles);
    return DHQR_OK;
  }
  CHECK(ensure(c, ws.w1, (size_t)nsplit * (size_t)wstride));
  // k_gemm_tn2 is persistent: its workgroups loop over the (column tile, row slab) units
  const dim3 gtn((unsigned)std::min<int64_t>(ntiles * nsplit, slots)), gred((unsigned)((wstride + 63) / 64));
  if (vec)
    hipLaunchKernelGGL((k_gemm_tn2<2>), gtn, dim3(512), 0, c->stream, Vp, ldv, C, ldc, rows, ncols, rps, ws.w1.p, wstride, (int64_t)0, 1, 0);
  else
    hipLaunchKernelGGL((k_gemm_tn2<1>), gtn, dim3(512), 0, c->stream, Vp, ldv, C, ldc, rows, ncols, rps, ws.w1.p, wstride, (int64_t)0, 1, 0);
  hipLaunchKernelGGL(k_reduce_splits, gred, dim3(256), 0, c->stream, (const double *)ws.w1.p, (int)nsplit, wstride, wstride, Y);
  return DHQR_OK;
}

// The T products of a pair: W (rows 0..255 of a matrix with leading dimension ldw) from Y (256 x ncols, ld 256; its
// lower half is overwritten):  W_a = T_a' Y_a,  Y_b -= (V_b' V_a) W_a,  W_b = T_b' Y_b.
static int32_t pair_tw(dhqr_ctx *c, double *Y, const double *Ta, const double *Tb, const double *Sba, double *W,
                       int64_t ldw, int64_t ncols) {
  const int64_t ntiles = (ncols + 127) / 128, ld2 = 2 * DHQR_NBV;
  double *Ya = Y, *Yb = Y + DHQR_NBV;  // rows 0..127 / 128..255 of Y
  if (ntiles <= 2) {
    // the lane's narrow update: all three T products in one launch (k_tw_fused, dhqr_gemm.h); T' of a panel sits right
    // behind its T in every operand buffer ([T | T' | alpha]: PanelBuf tails, row-split slots)
    const int64_t NN_ = (int64_t)DHQR_NBV * DHQR_NBV;
    hipLaunchKernelGGL((k_tw_fused<true>), dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, c->stream, (const double *)Y, ncols,
                       Ta + NN_, Tb + NN_, Sba, W, ldw);
    return DHQR_OK;
  }
  // W_a = T_a' Y_a  -> rows 0..127 of W
  hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Ta, (int64_t)DHQR_NBV,
                     (const double *)Ya, ld2, 1, (int64_t)0, (int64_t)DHQR_NBV, ncols, (int64_t)DHQR_NBV, W, ldw, (int64_t)0);
  // Y_b -= (V_b' V_a) W_a   (workspace: never predicated)
  launch_nn_sub<128>(c, true, dim3(1, (unsigned)ntiles), Sba, (int64_t)DHQR_NBV, (const double *)W, ldw, Yb, ld2,
                     (int64_t)DHQR_NBV, ncols, 0, false);
  // W_b = T_b' Y_b  -> rows 128..255 of W
  hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Tb, (int64_t)DHQR_NBV,
                     (const double *)Yb, ld2, 1, (int64_t)0, (int64_t)DHQR_NBV, ncols, (int64_t)DHQR_NBV, W + DHQR_NBV, ldw,
                     (int64_t)0);
  return DHQR_OK;
}

static int32_t pair_apply(dhqr_ctx *c, const double *Vp, int64_t ldv, int64_t rows, const double *Ta,
                          const double *Tb, const double *Sba, double *C, int64_t ncols, int64_t ldc,
                          dhqr_comm *ar = nullptr, int64_t rows_b_in = -1) {
  if (ncols <= 0) return DHQR_OK;
  const int64_t rows_b = rows_b_in >= 0 ? rows_b_in : rows - DHQR_NBV;
  const int64_t ntiles = (ncols + 127) / 128;
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  const int64_t ld2 = 2 * DHQR_NBV;
  CHECK(ensure(c, ws.w1r, (size_t)ld2 * (size_t)ncols));
  CHECK(ensure(c, ws.w2, (size_t)ld2 * (size_t)ncols));
  const bool vec = (ldc % 2 == 0) && (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(C) && aligned16(Vp);
  const int64_t wstride = ld2 * ncols;

  // Y = [V_a V_b]' C: ONE pass over C for both panels (stacked 256 x ncols result), then the split-K reduction
  CHECK(prof_begin(c, CAT_VTA));
  CHECK(pair_vtc(c, Vp, ldv, rows, C, ldc, ncols, vec, ws.w1r.p));
  CHECK(prof_switch(c, CAT_TW));

  /* (one event ends the previous section and starts this one) */
  if (ar) CHECK(comm_allreduce_sum(ar, ws.w1r.p, wstride, c->stream));
  CHECK(pair_tw(c, ws.w1r.p, Ta, Tb, Sba, ws.w2.p, ld2, ncols));
  CHECK(prof_switch(c, CAT_AVW));

  /* (one event ends the previous section and starts this one) */
  const int64_t gx = (rows + 127) / 128;
  if (ntiles < 48 && nn_row_chunks(c, gx) > 1) {
    // few column tiles but many row tiles (the row split's tall slabs): chunks of ROWS (same W, V and C from the chunk's row)
    const int64_t nrc = nn_row_chunks(c, gx), rpc = (gx + nrc - 1) / nrc * 128;
    for (int64_t r0 = 0; r0 < rows; r0 += rpc) {
      const int64_t nr = std::min(rpc, rows - r0), gxr = (nr + 127) / 128;
      const int swz = (gxr >= 16 && ntiles >= 16) ? 1 : 0;
      dim3 grid((unsigned)gxr, (unsigned)ntiles);
      if (swz) grid = dim3((unsigned)((((gxr + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
      launch_nn_sub<256>(c, vec, grid, Vp + r0, ldv, (const double *)ws.w2.p, ld2, C + r0, ldc, nr, ncols, swz, true);
    }
  } else {
    const int swz = (gx >= 16 && ntiles >= 16) ? 1 : 0;
    dim3 grid((unsigned)gx, (unsigned)ntiles);
    if (swz) grid = dim3((unsigned)((((gx + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
    launch_nn_sub<256>(c, vec, grid, Vp, ldv, (const double *)ws.w2.p, ld2, C, ldc, rows, ncols, swz, true);
  }
  CHECK(prof_end(c));
  if (c->profiling) {
    c->st.flops_gemm_vta += 2.0 * DHQR_NBV * ((double)rows + (double)rows_b) * (double)ncols;
    c->st.flops_gemm_avw += 2.0 * DHQR_NBV * ((double)rows + (double)rows_b) * (double)ncols;
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

// ---- four-panel trailing update (K = 512): two consecutive pairs (a, b), (c, d) in ONE pass over C for the subtraction ----
//   Y_1 = V_1' C,  Y_2 = V_2' C                     (V_1 = [V_a V_b], V_2 = [V_c V_d]: k_gemm_tn2 twice, C unchanged between)
//   W_1 = pair_tw(Y_1),  Y_2 -= (V_2' V_1) W_1,  W_2 = pair_tw(Y_2)        (= V_2' (C - V_1 W_1): the first pair applied)
//   C -= [V_1 V_2] [W_1; W_2]                       (k_gemm_nn_quad: C read and written once for 512 reflectors)
// V_2 starts 256 rows below V_1 and has the SAME leading dimension (cs_run gives every group buffer of a factorisation
// the ldv of the first panel when quads are on).  rows = rows of panel a.  Requires the 16-byte path (vec).
static int32_t quad_apply(dhqr_ctx *c, const double *V1, const double *V2, int64_t ldv, int64_t rows, const double *Ta,
                          const double *Tb, const double *Sba, const double *Tc, const double *Td, const double *Sdc,
                          const double *S21, double *C, int64_t ncols, int64_t ldc) {
  if (ncols <= 0) return DHQR_OK;
  const int64_t NB = DHQR_NBV, ld2 = 2 * NB, ld4 = 4 * NB, rows2 = rows - 2 * NB;
  const int64_t ntiles = (ncols + 127) / 128;
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  CHECK(ensure(c, ws.w1r, (size_t)ld4 * (size_t)ncols));
  CHECK(ensure(c, ws.w2, (size_t)ld4 * (size_t)ncols));
  double *Y1 = ws.w1r.p, *Y2 = ws.w1r.p + ld2 * ncols, *W = ws.w2.p;

  CHECK(prof_begin(c, CAT_VTA));
  CHECK(pair_vtc(c, V1, ldv, rows, C, ldc, ncols, true, Y1));
  CHECK(pair_vtc(c, V2, ldv, rows2, C + 2 * NB, ldc, ncols, true, Y2));
  CHECK(prof_switch(c, CAT_TW));

  /* (one event ends the previous section and starts this one) */
  CHECK(pair_tw(c, Y1, Ta, Tb, Sba, W, ld4, ncols));
  launch_nn_sub<256>(c, true, dim3(2, (unsigned)ntiles), S21, ld2, (const double *)W, ld4, Y2, ld2, ld2, ncols, 0, false);
  CHECK(pair_tw(c, Y2, Tc, Td, Sdc, W + ld2, ld4, ncols));
  CHECK(prof_switch(c, CAT_AVW));

  /* (one event ends the previous section and starts this one) */
  const int *st = pred_stat(c);
  const int64_t gx = (rows + 127) / 128;
  if (ntiles <= 2 && gx * ntiles < 512) {  // narrow (the lane / the head of a wide step): 64-row tiles
    hipLaunchKernelGGL((k_gemm_nn_quad<2, 64>), dim3((unsigned)((rows + 63) / 64), (unsigned)ntiles), dim3(256), 0, c->stream, V1,
                       V2 - 2 * NB, ldv, 2 * NB, (const double *)W, ld4, C, ldc, rows, ncols, 0, st, c->epoch);
  } else {
    const int swz = (gx >= 16 && ntiles >= 16) ? 1 : 0;
    dim3 grid((unsigned)gx, (unsigned)ntiles);
    if (swz) grid = dim3((unsigned)((((gx + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
    hipLaunchKernelGGL((k_gemm_nn_quad<2, 128>), grid, dim3(256), 0, c->stream, V1, V2 - 2 * NB, ldv, 2 * NB, (const double *)W, ld4, C,
                       ldc, rows, ncols, swz, st, c->epoch);
  }
  CHECK(prof_end(c));
  if (c->profiling) {
    const double f = 2.0 * NB * ((double)rows + (double)(rows - NB) + (double)rows2 + (double)(rows2 - NB)) * (double)ncols;
    c->st.flops_gemm_vta += f;
    c->st.flops_gemm_avw += f;
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

// S_21 = V_2' V_1 (256 x 256, ld 256) of two consecutive pair operands (V_2 starts 256 rows below V_1, same ldv): the
// cross term of quad_apply.  rows_a = rows of the first pair's first panel.
static int32_t quad_cross_gram(dhqr_ctx *c, const double *V1, const double *V2, int64_t ldv, int64_t rows_a, double *S21,
                               Buf *part = nullptr) {
  const int64_t NB = DHQR_NBV, rows2 = rows_a - 2 * NB, ld2 = 2 * NB;
  Buf &sp = part ? *part : c->spart;
  CHECK(narrow_vtc(c, V2, ldv, rows2, V1 + 2 * NB, ldv, ld2, true, sp, S21));
  LAUNCHCHECK();
  return DHQR_OK;
}

// S_ba = V_b' V_a (128 x 128) of a pair operand Vp = [V_a | V_b]: the cross term of pair_apply
static int32_t pair_cross_gram(dhqr_ctx *c, const double *Vp, int64_t ldv, int64_t rows_a, double *Sba, Buf *part = nullptr) {
  const int64_t NB = DHQR_NBV, rows_b = rows_a - NB;
  const size_t NN = (size_t)NB * NB;
  int64_t nsplit, rps;
  pick_split(rows_b, 1, 512, 256, &nsplit, &rps, 512, 64);
  Buf &sp = part ? *part : c->spart;
  CHECK(ensure(c, sp, (size_t)nsplit * NN));
  const double *Vb = Vp + NB + NB * ldv;
  const bool vec = (ldv % 2 == 0) && (rows_b % 2 == 0) && aligned16(Vp);
  if (vec)
    hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, Vb, ldv,
                       (const double *)(Vp + NB), ldv, 1, (int64_t)0, rows_b, (int64_t)NB, rps, sp.p, (int64_t)NB,
                       (int64_t)NN);
  else
    hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), dim3(1, (unsigned)nsplit), dim3(256), 0, c->stream, Vb, ldv,
                       (const double *)(Vp + NB), ldv, 1, (int64_t)0, rows_b, (int64_t)NB, rps, sp.p, (int64_t)NB,
                       (int64_t)NN);
  hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)(NN / 64)), dim3(256), 0, c->stream, (const double *)sp.p,
                     (int)nsplit, (int64_t)NN, (int64_t)NN, Sba);
  LAUNCHCHECK();
  return DHQR_OK;
}

// ---- simple blocked driver: no look-ahead, host-verified panels (DHQR_LOOKAHEAD=0, matrices of 1-2 panels) ----
static int32_t factor_blocked_simple(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  CHECK(ensure(c, c->vt, (size_t)panel_elems(m)));
  for (int64_t c0 = 0; c0 < n; c0 += DHQR_NBV) {
    const int64_t w = std::min<int64_t>(DHQR_NBV, n - c0), rows = m - c0;
    double *P = dA + c0 + c0 * lda;
    const PanelBuf pb = vt_view(c->vt.p, rows);
    c->tt_keep = c->tc_base ? c->tc_base + (c0 / DHQR_NBV) * (int64_t)(DHQR_NBV * DHQR_NBV) : nullptr;
    const int32_t rc = factor_panel_sync(c, P, rows, w, lda, dalpha + c0, pb);
    c->tt_keep = nullptr;
    CHECK(rc);
    if (c0 + w < n) CHECK(panel_apply(c, pb, rows, dA + c0 + (c0 + w) * lda, n - c0 - w, lda, 1));
  }
  return DHQR_OK;
}

#include "dhqr_comm.h"
#include "dhqr_hostio.h"
#include "dhqr_dist.h"
#include "dhqr_rowsplit.h"
#include "dhqr_mg.h"

static CsProblem cs_single(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  CsProblem pr;
  pr.c = c;
  pr.cm = nullptr;
  pr.A = dA;
  pr.m = m;
  pr.n = n;
  pr.lda = lda;
  pr.alpha = dalpha;
  pr.P = 1;
  pr.r = 0;
  pr.K = cs_nblocks(n);
  pr.ncl = n;
  return pr;
}

// ---- the single-workgroup route for small matrices (dhqr_small.h) -------------------------------------------------
// instantiations of k_small_qr_d<NR, NQ, EXTRA>: rows <= 16 NR, columns <= 32 NQ, NR * NQ doubles of matrix per lane
static inline int small_qr_fit(const dhqr_ctx *c, int64_t m, int64_t n) {
  if (!c->small_route || m < n || n < 1) return -1;
  if (m <= 128 && n <= 128) return 0;
  if (m <= 224 && n <= 224) return 1;
  if (m <= 256 && n <= 192) return 2;
  return -1;
}
static inline bool small_ldiv_fit(const dhqr_ctx *c, int64_t m, int64_t n) {
  return c->small_route && m <= SML_LDR && n >= 1 && n <= m;
}
static int32_t small_qr_launch(dhqr_ctx *c, int fit, const double *Asrc, int64_t lds, double *Adst, int64_t ldd, int64_t m,
                               int64_t n, double *alpha, unsigned long long *done, unsigned long long epoch) {
  // (k_small_qr_d: the reflectors are built by a ninth wave / by another wave than the column's owner, dhqr_small.h)
  if (fit == 0)
    hipLaunchKernelGGL((k_small_qr_d<8, 4, true>), dim3(1), dim3(SMB_THREADS), 0, c->stream, Asrc, lds, Adst, ldd, (int)m, (int)n, alpha, c->small_spin_limit, done, epoch);
  else if (fit == 1 && c->small_flags)  // (above 128 rows: no barrier in the column loop, LDS flags instead; DHQR_TUNE small_flags=0)
    hipLaunchKernelGGL((k_small_qr_d<14, 7, false, true>), dim3(1), dim3(SMQ_THREADS), 0, c->stream, Asrc, lds, Adst, ldd, (int)m, (int)n, alpha, c->small_spin_limit, done, epoch);
  else if (fit == 1)
    hipLaunchKernelGGL((k_small_qr_d<14, 7, false>), dim3(1), dim3(SMQ_THREADS), 0, c->stream, Asrc, lds, Adst, ldd, (int)m, (int)n, alpha, c->small_spin_limit, done, epoch);
  else if (c->small_flags)
    hipLaunchKernelGGL((k_small_qr_d<16, 6, false, true>), dim3(1), dim3(SMQ_THREADS), 0, c->stream, Asrc, lds, Adst, ldd, (int)m, (int)n, alpha, c->small_spin_limit, done, epoch);
  else
    hipLaunchKernelGGL((k_small_qr_d<16, 6, false>), dim3(1), dim3(SMQ_THREADS), 0, c->stream, Asrc, lds, Adst, ldd, (int)m, (int)n, alpha, c->small_spin_limit, done, epoch);
  LAUNCHCHECK();
  return DHQR_OK;
}
static int32_t small_ldiv_launch(dhqr_ctx *c, const double *A, int64_t lda, int64_t m, int64_t n, const double *alpha,
                                 const double *bin, double *bout, double *xout, double *Awork, unsigned long long *done = nullptr,
                                 unsigned long long epoch = 0) {
#define DHQR_SML(RPL_, CH_, AW_)                                                                                            \
  hipLaunchKernelGGL((k_small_ldiv<RPL_, CH_>), dim3(1), dim3(SML_THREADS), 0, c->stream, A, lda, (int)m, (int)n, alpha, bin, \
                     bout, xout, AW_, done, epoch)
  // (<= 128 rows: 64-column chunks straight from the caller's memory, no device copy of the factor)
  if (m <= 64) DHQR_SML(1, 64, (double *)nullptr);
  else if (m <= 128) DHQR_SML(2, 64, (double *)nullptr);
  else if (m <= 192) DHQR_SML(3, 16, Awork);
  else DHQR_SML(4, 16, Awork);
#undef DHQR_SML
  LAUNCHCHECK();
  return DHQR_OK;
}
// The host side of small_signal_done (dhqr_small.h): poll the pinned word, for at most ~2 ms; then (or when the kernel
// died) the stream's own synchronisation, which also reports an execution error.
static int32_t small_wait_done(dhqr_ctx *c, unsigned long long *done, unsigned long long epoch) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned i = 1;; ++i) {
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) == epoch) return DHQR_OK;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((i & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}
static int32_t small_pin_ensure(dhqr_ctx *c, size_t need) {
  if (need <= c->small_pin_cap) return DHQR_OK;
  c->small_pin_cap = 0;
  if (c->small_pin) {
    double *q = c->small_pin;
    c->small_pin = nullptr;
    HIPCHECK(hipHostFree(q));
  }
  need = (need + 4095) & ~(size_t)4095;
  HIPCHECK(hipHostMalloc((void **)&c->small_pin, (need + 8) * sizeof(double), hipHostMallocDefault));  // (+ the completion word)
  c->small_pin_cap = need;
  memset(c->small_pin + need, 0, 8 * sizeof(double));
  return DHQR_OK;
}
static inline void copy_cols(double *dst, int64_t ldd, const double *src, int64_t lds, int64_t m, int64_t n) {
  if (ldd == m && lds == m) {
    memcpy(dst, src, (size_t)m * (size_t)n * sizeof(double));
    return;
  }
  for (int64_t j = 0; j < n; ++j) memcpy(dst + j * ldd, src + j * lds, (size_t)m * sizeof(double));
}

// Every entry point runs on the context's device and restores the caller's current device on return (torch and
// other HIP users of the process read the current device from the runtime).
struct DeviceGuard {
  int prev = -1;
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
static int32_t enter_ctx(dhqr_ctx *c, DeviceGuard &g) {
  if (!c) return set_err(DHQR_EINVAL, "null context");
  int cur = -1;
  HIPCHECK(hipGetDevice(&cur));
  if (cur != c->device) {
    HIPCHECK(hipSetDevice(c->device));
    g.prev = cur;
  }
  return DHQR_OK;
}
#define ENTER(c_)   \
  DeviceGuard dg_; \
  CHECK(enter_ctx(c_, dg_))
// The reference's loops simply do not execute for a matrix without columns (src:127 `for j in Hl.colrange`,
// src:217 `for j in 1:n`): qr! returns an empty alpha, `\` an empty x.  Same here: n == 0 is a no-op.
static inline bool no_columns(int64_t m, int64_t n) { return n == 0 && m >= 0; }

static int32_t check_mat(const void *A, int64_t m, int64_t n, int64_t lda, bool need_tall) {
  if (!A) return set_err(DHQR_EINVAL, "null matrix pointer");
  if (m <= 0 || n <= 0) return set_err(DHQR_EINVAL, "m and n must be positive (m=%lld n=%lld)", (long long)m, (long long)n);
  if (need_tall && m < n) return set_err(DHQR_EINVAL, "m >= n required (m=%lld n=%lld)", (long long)m, (long long)n);
  if (lda < m) return set_err(DHQR_EINVAL, "leading dimension %lld < m=%lld", (long long)lda, (long long)m);
  // the MFMA kernels address a 128-column tile with 32-bit element offsets from the tile base
  if (lda > (int64_t)0xFFFFFFFFLL / 128)
    return set_err(DHQR_EINVAL, "leading dimension %lld too large (128 * ld must stay below 2^32 elements)", (long long)lda);
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
    const PanelBuf pb = vt_view(c->vt.p, rows);
    CHECK(panel_pack_and_t(c, P, rows, w, lda, dalpha ? dalpha + c0 : nullptr, pb));
    if (triangular) {
      if (nrhs - c0 > 0) CHECK(panel_apply(c, pb, rows, dB + c0 + c0 * ldb, nrhs - c0, ldb, trans));
    } else {
      CHECK(panel_apply(c, pb, rows, dB + c0, nrhs, ldb, trans));
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
  c->coop = prop.cooperativeLaunch != 0;
  g_live_ctx[device & 63].fetch_add(1);
  memset(&c->st, 0, sizeof(c->st));
  auto init = [&]() -> int32_t {  // any failure below releases what was created so far (dhqr_destroy)
    HIPCHECK(hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking));
    c->stream = c->own;
    {
      int lo = 0, hi = 0;
      HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // hi = numerically lowest = highest priority
      HIPCHECK(hipStreamCreateWithPriority(&c->hi, hipStreamNonBlocking, hi));
      if (const char *e = getenv("DHQR_LANE_SIDE")) c->lane_side = atoi(e) != 0;
      c->hi_priority = hi;  // c->hi2 is created by the single-rank driver on first use (cs_run)
    }
    if (const char *e = getenv("DHQR_LOOKAHEAD")) c->lookahead = atoi(e) != 0;
    {
      int ncu = 0;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) {
        c->rankk_wgs = ncu;
        c->ncu = ncu;
      }
    }
    if (const char *e = getenv("DHQR_SPARE_CUS")) {
      c->spare_cus = std::max(0, std::min(c->ncu - 8, atoi(e)));
      c->spare_cus_set = true;
    }
    if (const char *e = getenv("DHQR_QUAD")) c->quad = atoi(e) != 0;
    if (const char *e = getenv("DHQR_QUAD_MIN_COLS")) c->quad_min_cols = std::max<int64_t>(0, atoll(e));
    { long long v; if (tune_get("rankk_wgs", &v)) c->rankk_wgs = std::max(2, (int)v); }
    if (const char *e = getenv("DHQR_RANKK")) c->rankk = c->rankk_tall = c->rankk_xtall = std::min(5, std::max(1, atoi(e)));
    if (const char *e = getenv("DHQR_RANKK_MAX")) c->rankk_max = std::min(DHQR_RK_KMAX, std::max(1, atoi(e)));
    { long long v; if (tune_get("rankk_max_min_cols", &v)) c->rankk_max_min_cols = std::max(0, (int)v); }
    { long long v; if (tune_get("nn_chunk_tiles", &v)) c->nn_chunk_tiles = std::max(1, (int)v); }
    if (const char *e = getenv("DHQR_NN_SPLIT")) c->nn_split = std::min(16, std::max(1, atoi(e)));
    if (const char *e = getenv("DHQR_RANKK_PIPE")) c->rankk_pipe = std::min(2, std::max(0, atoi(e)));
    { long long v; if (tune_get("tn_min_tiles", &v)) c->tn_model_min_tiles = (int)v; }
    { long long v; if (tune_get("small_flags", &v)) c->small_flags = v != 0; }
    { long long v; if (tune_get("short_panel_small", &v)) c->short_panel_small = v != 0; }
    { long long v; if (tune_get("partial_unblocked", &v)) c->partial_unblocked = v != 0; }
    { long long v; if (tune_get("partial_unblocked_max_rows", &v)) c->partial_unblocked_max_rows = v; }
    { long long v; if (tune_get("small_spin_limit", &v)) c->small_spin_limit = (int)v; }
    { long long v; if (tune_get("tn2_rgroups", &v)) c->tn2_rgroups = (int)v; }
    { long long v; if (tune_get("tn2_rg8_rows", &v)) c->tn2_rg8_rows = v; }
    { long long v; if (tune_get("tn2_rg4_rows", &v)) c->tn2_rg4_rows = v; }
    { long long v; if (tune_get("tn2_rg2_rows", &v)) c->tn2_rg2_rows = v; }
    { long long v; if (tune_get("tn_spare", &v)) c->tn_spare = std::max(0, std::min((c->ncu - 8) / 2, (int)v)); }
    { long long v; if (tune_get("tn_spare_cols", &v)) c->tn_spare_cols = v; }
    if (const char *e = getenv("DHQR_PAIR")) c->pair = atoi(e) != 0;
    if (const char *e = getenv("DHQR_PAIR_MIN_N")) c->pair_min_n = atoll(e);
    HIPCHECK(hipHostMalloc((void **)&c->hflag, 4 * sizeof(int), hipHostMallocDefault));
    HIPCHECK(hipMalloc((void **)&c->dstat, 16 * sizeof(int)));
    HIPCHECK(hipMalloc((void **)&c->zflags, DHQR_PIPE_INTS * sizeof(int)));  // 128 flags + the error word (dhqr_common.h)
    HIPCHECK(hipMemsetAsync(c->zflags, 0, DHQR_PIPE_INTS * sizeof(int), c->stream));
    {
      int limit = DHQR_PIPE_SPIN_LIMIT;
      { long long v; if (tune_get("spin_limit", &v)) limit = std::max(1, (int)v); }
      HIPCHECK(hipMemcpyAsync(c->zflags + DHQR_PIPE_LIMIT_OFFSET, &limit, sizeof(int), hipMemcpyHostToDevice, c->stream));
      HIPCHECK(hipStreamSynchronize(c->stream));  // `limit` is a stack variable
    }
    if (const char *e = getenv("DHQR_SMALL")) c->small_route = atoi(e) != 0;
    if (const char *e = getenv("DHQR_ZPIPE")) c->zpipe = atoi(e) != 0;
    if (const char *e = getenv("DHQR_SOLVE_PIPE")) c->solve_pipe = std::max(0, std::min(3, atoi(e)));
    if (const char *e = getenv("DHQR_KEEP_T")) c->keep_t = atoi(e) != 0;
    { long long v; if (tune_get("qtb_vec", &v)) c->qtb_vec = (int)v; }
    { long long v; if (tune_get("gram_strips", &v)) c->gram_strips = v != 0; }
    { long long v; if (tune_get("fuse_fix", &v)) c->fuse_fix = v != 0; }
    { long long v; if (tune_get("commit_off", &v)) c->commit_off = (int)v; }
    hipLaunchKernelGGL(k_set_status, dim3(1), dim3(64), 0, c->stream, c->dstat, INT_MAX);
    LAUNCHCHECK();
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  const int32_t rc_init = init();
  if (rc_init != DHQR_OK) {
    (void)dhqr_destroy(c);
    return rc_init;
  }
  if (const char *e = getenv("DHQR_CHOLQR_PASSES")) c->cholqr_passes = atoi(e) == 2 ? 2 : 1;
  if (const char *e = getenv("DHQR_TSQR")) {  // 1: every panel through TSQR-HR
    if (atoi(e) != 0) c->cholqr_passes = 3;
  }
  if (const char *e = getenv("DHQR_TSQR_RUNG")) c->tsqr_rung = atoi(e) != 0 ? 1 : 0;
  if (const char *e = getenv("DHQR_PANEL")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 3) c->panel_impl = v;
  }
  *out = c;
  return DHQR_OK;
}

int32_t dhqr_destroy(dhqr_ctx *c) {
  if (!c) return DHQR_OK;
  g_live_ctx[c->device & 63].fetch_sub(1);
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipDeviceSynchronize();
  cs_state_free(c);
  rs_state_free(c);
  if (c->hio) {
    hio_free(*c->hio);
    delete c->hio;
    c->hio = nullptr;
  }
  Buf *bufs[] = {&c->vbuf, &c->vt, &c->vts, &c->ws[0].w1, &c->ws[0].w1r, &c->ws[0].w2, &c->ws[1].w1,
                 &c->ws[1].w1r, &c->ws[1].w2, &c->ws[2].w1, &c->ws[2].w1r, &c->ws[2].w2, &c->spart, &c->spart2, &c->sfull, &c->scratch, &c->pbuf, &c->rbuf, &c->tsq, &c->zsolve_lo, &c->host_mat, &c->sv_T, &c->sv_S, &c->sv_part, &c->sv_small, &c->tc_T, &c->tc_alpha, &c->small_dev, &c->sv_bkp};
  for (Buf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto &e : c->evs) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  if (c->hflag) (void)hipHostFree(c->hflag);
  if (c->small_pin) (void)hipHostFree(c->small_pin);
  if (c->dstat) (void)hipFree(c->dstat);
  if (c->zflags) (void)hipFree(c->zflags);
  for (hipEvent_t e : c->zev)
    if (e) (void)hipEventDestroy(e);
  if (c->hi) (void)hipStreamDestroy(c->hi);
  if (c->hi2) (void)hipStreamDestroy(c->hi2);
  if (c->cstream) (void)hipStreamDestroy(c->cstream);
  for (hipEvent_t e : c->ev_commit)
    if (e) (void)hipEventDestroy(e);
  if (c->own) (void)hipStreamDestroy(c->own);
  delete c;
  return DHQR_OK;
}

int32_t dhqr_set_stream(dhqr_ctx *c, void *s) {
  ENTER(c);
  c->stream = (hipStream_t)s;  // NULL is the device's default (null) stream, as torch uses it
  return DHQR_OK;
}
int32_t dhqr_use_own_stream(dhqr_ctx *c) {
  ENTER(c);
  c->stream = c->own;
  return DHQR_OK;
}
// A column pipeline whose bounded hand-over wait expired (dhqr_common.h) has produced wrong numbers instead of hanging
// the GPU: reported here, by the entry points that synchronise anyway.  c->stream must be idle.
static int32_t pipe_error_report(dhqr_ctx *c, int e) {
  HIPCHECK(hipMemsetAsync(c->zflags + DHQR_PIPE_ERR_OFFSET, 0, sizeof(int), c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return set_err(DHQR_EHIP, "an inter-workgroup pipeline (k_zpanel_pipe / rankk_lead_pipe / k_backsub_pipe, launch %d) gave up "
                 "waiting for a lower-indexed workgroup: the results of that call are invalid; DHQR_ZPIPE=0 / DHQR_RANKK_PIPE=0 / "
                 "DHQR_SOLVE_PIPE=0 select the kernels without inter-workgroup waits", e);
}
static int32_t pipe_error_check(dhqr_ctx *c) {
  if (!c->zflags) return DHQR_OK;
  int e = 0;
  HIPCHECK(hipMemcpy(&e, c->zflags + DHQR_PIPE_ERR_OFFSET, sizeof(int), hipMemcpyDeviceToHost));
  if (e == 0) return DHQR_OK;
  if (e == 0x7ffffffe && c->retry.valid) {
    // A wait of the Q'b kernels expired and the last solve took the persistent kernel (whose workgroups must all be resident:
    // something else held compute units).  Repeat it from the saved b with one launch per panel step -- no workgroup of
    // that form waits for a higher-indexed one -- and report only if that fails as well.
    const dhqr_ctx::SolveRetry r = c->retry;
    c->retry.valid = false;
    HIPCHECK(hipMemsetAsync(c->zflags + DHQR_PIPE_ERR_OFFSET, 0, sizeof(int), c->stream));
    HIPCHECK(hipMemcpyAsync(r.b, c->sv_bkp.p, (size_t)r.m * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    const bool was = c->profiling;
    c->profiling = false;
    const int32_t rc = solve_pipelined(c, r.A, r.m, r.n, r.lda, r.alpha, r.b, false);
    c->profiling = was;
    CHECK(rc);
    HIPCHECK(hipStreamSynchronize(c->stream));
    c->n_solve_retry++;
    HIPCHECK(hipMemcpy(&e, c->zflags + DHQR_PIPE_ERR_OFFSET, sizeof(int), hipMemcpyDeviceToHost));
    if (e == 0) return DHQR_OK;
  }
  return pipe_error_report(c, e);
}

int32_t dhqr_synchronize(dhqr_ctx *c) {
  ENTER(c);
  HIPCHECK(hipStreamSynchronize(c->stream));
  return pipe_error_check(c);
}
int32_t dhqr_trim(dhqr_ctx *c) {
  ENTER(c);
  HIPCHECK(hipStreamSynchronize(c->stream));
  HIPCHECK(hipDeviceSynchronize());  // the lane, side, comm and copy streams of this context
  Buf *bs[] = {&c->host_mat, &c->sv_T, &c->sv_S, &c->sv_part, &c->sv_small, &c->tc_T, &c->tc_alpha, &c->vts, &c->tsq, &c->zsolve_lo,
               &c->small_dev, &c->sv_bkp};
  c->retry.valid = false;
  if (c->small_pin) {
    (void)hipHostFree(c->small_pin);
    c->small_pin = nullptr;
    c->small_pin_cap = 0;
  }
  for (Buf *b : bs)
    if (b->p) {
      (void)hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
  c->tc_valid = false;
  c->sv_units_dev = nullptr;
  if (c->cs)
    for (Buf &b : c->cs->gbuf)
      if (b.p) {
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
      }
  if (c->hio) {
    hio_free(*c->hio);
    delete c->hio;
    c->hio = nullptr;
  }
  return pipe_error_check(c);
}
int32_t dhqr_set_profiling(dhqr_ctx *c, int32_t on) {
  ENTER(c);
  if (c->profiling && !on) CHECK(prof_resolve(c));
  c->profiling = on != 0;
  return DHQR_OK;
}
int32_t dhqr_reset_stats(dhqr_ctx *c) {
  ENTER(c);
  HIPCHECK(hipStreamSynchronize(c->stream));
  c->ev_used = 0;
  memset(&c->st, 0, sizeof(c->st));
  c->n_fast = c->n_fallback = 0;
  c->n_tsqr = 0;
  return DHQR_OK;
}
int32_t dhqr_get_stats(dhqr_ctx *c, dhqr_stats *out) {
  ENTER(c);
  if (!out) return set_err(DHQR_EINVAL, "null stats pointer");
  CHECK(prof_resolve(c));
  *out = c->st;
  return DHQR_OK;
}

// R (128 x 128, upper triangular, row signs as the tree produces them) of a device-resident rows x 128 panel.  Async.
int32_t dhqr_tsqr_r_f64(dhqr_ctx *c, const double *dP, int64_t rows, int64_t ldp, double *dR) {
  ENTER(c);
  if (!dP || !dR) return set_err(DHQR_EINVAL, "null pointer argument");
  if (rows < 1 || ldp < rows) return set_err(DHQR_EINVAL, "bad panel shape: rows=%lld ldp=%lld", (long long)rows, (long long)ldp);
  return tsqr_local_r(c, dP, ldp, rows, dR);
}

int32_t dhqr_set_r_source(dhqr_ctx *c, int32_t source) {
  if (!c || source < 1 || source > 3) return set_err(DHQR_EINVAL, "R source must be 1 (Cholesky), 2 (CholeskyQR2) or 3 (TSQR tree)");
  c->cholqr_passes = source;
  return DHQR_OK;
}

int32_t dhqr_set_tsqr_rung(dhqr_ctx *c, int32_t on) {
  if (!c) return set_err(DHQR_EINVAL, "null context");
  c->tsqr_rung = on != 0 ? 1 : 0;
  return DHQR_OK;
}

int32_t dhqr_set_small_route(dhqr_ctx *c, int32_t on) {
  if (!c) return set_err(DHQR_EINVAL, "null context");
  c->small_route = on != 0 ? 1 : 0;
  return DHQR_OK;
}

int32_t dhqr_get_solve_retries(dhqr_ctx *c, int64_t *n_retries) {
  if (!c || !n_retries) return set_err(DHQR_EINVAL, "null argument");
  *n_retries = c->n_solve_retry;
  return DHQR_OK;
}

int32_t dhqr_get_tsqr_count(dhqr_ctx *c, int64_t *n_tsqr) {
  if (!c || !n_tsqr) return set_err(DHQR_EINVAL, "null argument");
  *n_tsqr = c->n_tsqr;
  return DHQR_OK;
}

int32_t dhqr_get_panel_counters(dhqr_ctx *c, int64_t *n_fast, int64_t *n_fallback) {
  ENTER(c);
  if (n_fast) *n_fast = c->n_fast;
  if (n_fallback) *n_fallback = c->n_fallback;
  return DHQR_OK;
}

int32_t dhqr_fill_uniform_f64(dhqr_ctx *c, double *dA, int64_t rows, int64_t cols, int64_t lda,
                              uint64_t seed, int64_t global_m, int64_t row0, int64_t colblock,
                              int32_t nranks, int32_t rank) {
  ENTER(c);
  CHECK(check_mat(dA, rows, cols, lda, false));
  if (colblock <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || global_m < rows)
    return set_err(DHQR_EINVAL, "bad layout arguments to dhqr_fill_uniform_f64");
  if (c->tc_A == dA) c->tc_valid = false;  // kept T factors belong to what this call overwrites
  const int64_t total = rows * cols;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(k_fill_uniform, dim3(grid), dim3(256), 0, c->stream, dA, rows, cols, lda, seed,
                     global_m, row0, colblock, (int)nranks, (int)rank);
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_factor_f64(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha,
                        int32_t nb) {
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  if (!dalpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  if (nb != 0 && nb != DHQR_NB)
    return set_err(DHQR_EINVAL, "nb must be 0 (unblocked) or %d (blocked); got %d", DHQR_NB, nb);
  if (c->tc_A == dA) c->tc_valid = false;  // whatever was kept for this matrix is gone
  c->retry.valid = false;
  if (const int fit = small_qr_fit(c, m, n); fit >= 0) {  // the reference's algorithm in one launch, whatever nb says
    CHECK(prof_begin(c, CAT_RANK1));
    CHECK(small_qr_launch(c, fit, dA, lda, dA, lda, m, n, dalpha));
    if (c->profiling)
      for (int64_t j = 0; j + 1 < n; ++j) c->st.bytes_rank1 += 16.0 * (double)(m - j) * (double)(n - j - 1);
    return prof_end(c);
  }
  if (nb == 0) return factor_unblocked_cols(c, dA, m, n, lda, dalpha, CAT_RANK1);
  // kept T factors (solve_pipelined): every panel's k_build_t stores T' here as well
  const int64_t np = cs_nblocks(n);
  c->tc_valid = false;
  c->tc_base = nullptr;
  if (c->keep_t && c->solve_pipe) {
    CHECK(ensure(c, c->tc_T, (size_t)np * DHQR_NBV * DHQR_NBV));
    CHECK(ensure(c, c->tc_alpha, (size_t)n + 16));
    c->tc_base = c->tc_T.p;
  }
  int32_t rc;
  if (!c->lookahead || np < 3) {
    rc = factor_blocked_simple(c, dA, m, n, lda, dalpha);
  } else {
    const CsProblem pr = cs_single(c, dA, m, n, lda, dalpha);
    rc = cs_factor(pr);
  }
  if (rc == DHQR_OK && c->tc_base) {
    HIPCHECK(hipMemcpyAsync(c->tc_alpha.p, dalpha, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->tc_A = dA;
    c->tc_m = m;
    c->tc_n = n;
    c->tc_lda = lda;
    c->tc_valid = true;
  }
  c->tc_base = nullptr;
  c->tt_keep = nullptr;
  return rc;
}

int32_t dhqr_qr_f64(dhqr_ctx *c, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha,
                    int32_t nb) {
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  if (nb != 0 && nb != DHQR_NB)
    return set_err(DHQR_EINVAL, "nb must be 0 (unblocked) or %d (blocked); got %d", DHQR_NB, nb);
  if (const int fit = small_qr_fit(c, m, n); fit >= 0) {
    // memcpy -> ONE launch that reads and writes the pinned staging buffer across PCIe itself -> one synchronisation ->
    // memcpy (dhqr_small.h): no hipMemcpy, no device copy
    const size_t na = (size_t)m * (size_t)n;
    CHECK(small_pin_ensure(c, na + (size_t)n));
    double *pA = c->small_pin, *pal = pA + na;
    unsigned long long *done = reinterpret_cast<unsigned long long *>(c->small_pin + c->small_pin_cap);
    copy_cols(pA, m, hA, lda, m, n);
    CHECK(small_qr_launch(c, fit, pA, m, pA, m, m, n, pal, done, ++c->small_epoch));
    CHECK(small_wait_done(c, done, c->small_epoch));
    if (fit >= 1 && c->small_flags && std::isnan(pal[0])) {
      // the flag form's waits are bounded and it answers NaN when one expires (dhqr_small.h): once more, with a barrier
      // per column (a NaN in the caller's first column gives the same answer twice)
      copy_cols(pA, m, hA, lda, m, n);
      c->small_flags = 0;
      const int32_t rc = small_qr_launch(c, fit, pA, m, pA, m, m, n, pal);
      c->small_flags = 1;
      CHECK(rc);
      HIPCHECK(hipStreamSynchronize(c->stream));
    }
    copy_cols(hA, lda, pA, m, m, n);
    memcpy(halpha, pal, (size_t)n * sizeof(double));
    return DHQR_OK;
  }
  // The device copy of the matrix lives in the context between calls (hipMalloc + hipFree of 8 GiB cost ~0.3 s per call at
  // 32768^2, a third of the factorisation; `qr!` is typically called in a loop, test/runtests.jl:84); freed by dhqr_destroy.
  const int64_t ldd = (m + 1) & ~(int64_t)1;
  CHECK(ensure(c, c->host_mat, (size_t)ldd * (size_t)n + (size_t)n + (size_t)m + 32));  // (dhqr_ldiv_f64 keeps b behind alpha)
  double *dA = c->host_mat.p, *dal = dA + (((size_t)ldd * (size_t)n + 1) & ~(size_t)1);
  // Default: the plain three-phase form (one hipMemcpy2D up, factorisation, one down).  DHQR_HOSTIO=1: the staged form of
  // dhqr_hostio.h (every committed column block travels back while later panels are factored).  It lost every A/B on this
  // stack (32768^2: 1.16-1.25 s against 1.10-1.15 s, profiles/r04_hostio*.txt, r05_hostio*.txt): an asynchronous
  // device-to-host copy runs as a blit kernel whose grid covers the chip while it moves data at the PCIe rate, and the
  // factorisation stretches by what the download was meant to hide; a 16-workgroup copy-out kernel storing into the pinned
  // buffer was slower still (1.29 s).  Kept for stacks whose copies run on the DMA engines.
  static const bool overlap = [] { const char *e = getenv("DHQR_HOSTIO"); return e && atoi(e) == 1; }();
  int32_t rc = DHQR_OK;
  auto plain = [&]() -> int32_t {
    HIPCHECK(hipMemcpy2DAsync(dA, ldd * sizeof(double), hA, lda * sizeof(double), m * sizeof(double),
                              n, hipMemcpyHostToDevice, c->stream));
    // the reference factors whatever m x n block it is given; m odd is handled by the scalar path
    CHECK(dhqr_factor_f64(c, dA, m, n, ldd, dal, nb));
    HIPCHECK(hipMemcpy2DAsync(hA, lda * sizeof(double), dA, ldd * sizeof(double), m * sizeof(double),
                              n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(halpha, dal, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return pipe_error_check(c);
  };
  // dhqr_hostio.h: staged upload on two copy streams, every column block downloaded behind its panel's commit
  auto overlapped = [&]() -> int32_t {
    if (!c->hio) c->hio = new HostIo();
    HostIo &h = *c->hio;
    const int64_t K = (n + DHQR_NBV - 1) / DHQR_NBV;
#ifdef DHQR_HOSTIO_TRACE
    const bool trace = true;  // phase times on stderr (a -DDHQR_HOSTIO_TRACE build: tools/hostio_bench.py)
#else
    const bool trace = false;
#endif
    const auto tp0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); };
    CHECK(hio_upload(h, hA, m, n, lda, dA, ldd, c->stream));
    if (trace) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      fprintf(stderr, "[hostio] upload done at %.1f ms\n", since());
    }
    h.hA = hA;
    h.dA = dA;
    h.m = m;
    h.n = n;
    h.lda = lda;
    h.ldd = ldd;
    h.use = 0;
    h.done.assign((size_t)K, 0);
    const int64_t resumes = c->n_resume, fallbacks = c->n_fallback;
    c->panel_hook = hio_panel_hook;
    c->panel_hook_arg = &h;
    const int32_t rf = dhqr_factor_f64(c, dA, m, n, ldd, dal, nb);
    c->panel_hook = nullptr;
    c->panel_hook_arg = nullptr;
    if (trace) fprintf(stderr, "[hostio] factorisation returned at %.1f ms\n", since());
    if (rf != DHQR_OK) {
      (void)hio_drain(h);
      return rf;
    }
    // a block that left while a rejected panel was being redone may be stale: take everything again (rare)
    if (c->n_resume != resumes || c->n_fallback != fallbacks) {
      CHECK(hio_drain(h));
      h.done.assign((size_t)K, 0);
    }
    HIPCHECK(hipEventRecord(h.ev[0], c->stream));  // what the hook did not cover waits for the whole factorisation
    for (int i = 0; i < 2; ++i) HIPCHECK(hipStreamWaitEvent(h.s[i], h.ev[0], 0));
    for (int64_t k = 0; k < K; ++k)
      if (!h.done[(size_t)k]) CHECK(hio_download_block(h, k * DHQR_NBV, std::min<int64_t>(DHQR_NBV, n - k * DHQR_NBV), nullptr));
    HIPCHECK(hipMemcpyAsync(halpha, dal, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    CHECK(hio_drain(h));
    if (trace) fprintf(stderr, "[hostio] last block on the host at %.1f ms\n", since());
    return pipe_error_check(c);
  };
  // the staged path needs four pinned staging buffers; without them (hipHostMalloc failed) the plain form still works
  bool staged = overlap;
  if (staged) {
    if (!c->hio) c->hio = new HostIo();
    if (hio_init(*c->hio, ldd, n) != DHQR_OK) staged = false;
  }
  rc = staged ? overlapped() : plain();
  (void)hipStreamSynchronize(c->stream);
  if (c->hio) (void)hio_drain(*c->hio);
  return rc;
}

// The solve of dhqr_qtb.h on c->stream: batched Gram / T' pre-pass, one k_qtb_step launch per panel, one pipelined
// back-substitution launch.  Nothing synchronises; db[0:n] <- x.
static int32_t solve_pipelined(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha,
                               double *db, bool allow_persist) {
  const int np = (int)((n + DHQR_NBV - 1) / DHQR_NBV);
  c->retry.valid = false;
  const bool vec = (lda % 2 == 0) && (m % 2 == 0) && aligned16(dA) && aligned16(db);
  // ---- unit table of the Gram pre-pass: panel k = rows [128 k, m), slabs of rps rows
  if (c->sv_m != m || c->sv_n != n) {
    int64_t total = 0;
    for (int k = 0; k < np; ++k) total += m - (int64_t)k * DHQR_NBV;
    int64_t rps = ((total / 1024 + 15) / 16) * 16;
    rps = std::min<int64_t>(std::max<int64_t>(rps, 256), 4096);
    c->sv_units.assign((size_t)np + 1, 0);
    for (int k = 0; k < np; ++k)
      c->sv_units[(size_t)k + 1] = c->sv_units[(size_t)k] + (int)((m - (int64_t)k * DHQR_NBV + rps - 1) / rps);
    c->sv_rps = rps;
    c->sv_m = m;
    c->sv_n = n;
    c->sv_units_dev = nullptr;
  }
  const int nunits = c->sv_units[(size_t)np];
  const int VEC = (c->qtb_vec == 1 || !vec) ? 1 : (c->qtb_vec == 2 ? 2 : (m >= 16384 ? 2 : 1));
  const int64_t SS = 64 * VEC;
  const int64_t maxsl = std::max<int64_t>(8, std::min<int64_t>(c->ncu, 256));  // slabs = workgroups: at most one per CU
  const int64_t sl = SS * ((m + SS * maxsl - 1) / (SS * maxsl));
  const int64_t nsl = (m + sl - 1) / sl;
  CHECK(ensure(c, c->sv_T, (size_t)np * QTB_NB2));
  CHECK(ensure(c, c->sv_S, (size_t)np * QTB_NB2));
  CHECK(ensure(c, c->sv_part, (size_t)nunits * QTB_NB2));
  const size_t n_ypart = (size_t)std::max<int64_t>(nsl, (m + 63) / 64) * QTB_NB, n_w = (size_t)(np + 1) * QTB_NB;
  const size_t n_ints = 3 * (size_t)(np + 1) + (size_t)np + 2;  // unit table | arrival counters | w flags | block flags | kept-T flag
  CHECK(ensure(c, c->sv_small, n_ypart + n_w + (n_ints + 1) / 2 + 16));
  double *ypart = c->sv_small.p, *wbuf = ypart + n_ypart;
  int *units = reinterpret_cast<int *>(wbuf + n_w), *counter = units + (np + 1), *wflag = counter + (np + 1),
      *flags = wflag + (np + 1), *kept = flags + np;
  if (c->sv_units_dev != units) {  // the table stays on the device between calls on the same shape
    HIPCHECK(hipMemcpyAsync(units, c->sv_units.data(), (size_t)(np + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->sv_units_dev = units;
  }
  HIPCHECK(hipMemsetAsync(counter, 0, (size_t)(3 * np + 3) * sizeof(int), c->stream));  // (... and *kept = 0)
  // kept T factors: this context's last blocked factorisation was of this very matrix -> compare alpha on the device
  const bool maybe_kept = c->keep_t && c->tc_valid && c->tc_A == dA && c->tc_m == m && c->tc_n == n && c->tc_lda == lda &&
                          c->tc_T.p != nullptr;
  if (maybe_kept)
    hipLaunchKernelGGL(k_qtb_same_alpha, dim3(1), dim3(1024), 0, c->stream, dalpha, (const double *)c->tc_alpha.p, n, kept);
  const double *Tt_kept = maybe_kept ? c->tc_T.p : c->sv_T.p;
  // ---- pre-pass (independent of b): S_k = V_k'V_k, T_k' = (I + striu(S_k))^{-T}
  if (vec)
    hipLaunchKernelGGL((k_gemm_tn_gram_batch<2>), dim3((unsigned)nunits), dim3(256), 0, c->stream, dA, lda, m, n, c->sv_rps,
                       (const int *)units, np, c->sv_part.p, (const int *)kept);
  else
    hipLaunchKernelGGL((k_gemm_tn_gram_batch<1>), dim3((unsigned)nunits), dim3(256), 0, c->stream, dA, lda, m, n, c->sv_rps,
                       (const int *)units, np, c->sv_part.p, (const int *)kept);
  hipLaunchKernelGGL(k_qtb_sum_gram, dim3((unsigned)np, 16), dim3(256), 0, c->stream, (const double *)c->sv_part.p,
                     (const int *)units, n, c->sv_S.p, (const int *)kept);
  hipLaunchKernelGGL(k_build_t_batch, dim3((unsigned)np), dim3(1024), 0, c->stream, (const double *)c->sv_S.p, n, c->sv_T.p,
                     (const int *)kept);
  // ---- b <- Q'b (src:215-242): panel step k updates by panel k-1 and forms the dots of panel k.  One persistent launch
  // when every workgroup is certain to be resident (one per CU at most, this context alone on the device, a real device:
  // the CPU emulator runs workgroups one after the other and reports no cooperative launch), else one launch per step.
  int *err = c->zflags + DHQR_PIPE_ERR_OFFSET;
  const bool persist = allow_persist &&
                       (c->solve_pipe == 3 || (c->solve_pipe == 1 && c->coop && g_live_ctx[c->device & 63].load() == 1));
  // persistent form: one 64 VEC-row slab per workgroup (VEC = 1 with 4 waves up to 64 rows x #CU, VEC = 2 with 8 waves beyond)
  const int pVEC = (!vec || m <= 64 * (int64_t)c->ncu) ? 1 : 2;
  const int64_t pnsl = (m + 64 * pVEC - 1) / (64 * pVEC);
  if (persist && pnsl <= (int64_t)c->ncu && c->qtb_vec <= 0) {
    // what a repetition needs (pipe_error_check): b as it is now, and the arguments
    CHECK(ensure(c, c->sv_bkp, (size_t)m));
    HIPCHECK(hipMemcpyAsync(c->sv_bkp.p, db, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->retry.valid = true;
    c->retry.A = dA;
    c->retry.alpha = dalpha;
    c->retry.b = db;
    c->retry.m = m;
    c->retry.n = n;
    c->retry.lda = lda;
    if (pVEC == 2)
      hipLaunchKernelGGL((k_qtb_persist<2, 8>), dim3((unsigned)pnsl), dim3(512), 0, c->stream, dA, lda, m, n, np, db,
                         (const double *)c->sv_T.p, Tt_kept, (const int *)kept, wbuf, ypart, counter, wflag, err);
    else
      hipLaunchKernelGGL((k_qtb_persist<1, 4>), dim3((unsigned)pnsl), dim3(256), 0, c->stream, dA, lda, m, n, np, db,
                         (const double *)c->sv_T.p, Tt_kept, (const int *)kept, wbuf, ypart, counter, wflag, err);
  } else {
    for (int k = 0; k <= np; ++k) {
      const int64_t rfirst = (int64_t)(k >= 1 ? k - 1 : 0) * DHQR_NBV;
      const unsigned grid = (unsigned)(nsl - rfirst / sl);
      if (VEC == 2)
        hipLaunchKernelGGL((k_qtb_step<2>), dim3(grid), dim3(256), 0, c->stream, dA, lda, m, n, k, np, sl, db,
                           (const double *)c->sv_T.p, Tt_kept, (const int *)kept, wbuf, ypart, counter, err, (int64_t)0, (double *)nullptr);
      else
        hipLaunchKernelGGL((k_qtb_step<1>), dim3(grid), dim3(256), 0, c->stream, dA, lda, m, n, k, np, sl, db,
                           (const double *)c->sv_T.p, Tt_kept, (const int *)kept, wbuf, ypart, counter, err, (int64_t)0, (double *)nullptr);
    }
  }
  // ---- back substitution (src:244-282): one pipelined launch
  hipLaunchKernelGGL(k_backsub_pipe, dim3((unsigned)np), dim3(BSP_THREADS), 0, c->stream, dA, lda, dalpha, db, n, flags,
                     c->zflags + DHQR_PIPE_ERR_OFFSET);
  return DHQR_OK;
}

int32_t dhqr_solve_f64(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda,
                       const double *dalpha, double *db) {
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  if (!dalpha || !db) return set_err(DHQR_EINVAL, "null alpha or b pointer");
  CHECK(prof_begin(c, CAT_SOLVE));
  if (small_ldiv_fit(c, m, n)) {  // one single-workgroup launch (dhqr_small.h)
    CHECK(small_ldiv_launch(c, dA, lda, m, n, dalpha, db, db, nullptr, nullptr));
    return prof_end(c);
  }
  const bool was = c->profiling;
  c->profiling = false;  // the solve is timed as one group
  int32_t rc;
  if (c->solve_pipe) {
    rc = solve_pipelined(c, dA, m, n, lda, dalpha, db, true);
  } else {
    rc = apply_q_impl(c, dA, m, n, lda, dalpha, db, 1, m, 1, false);  // src:215-242
    if (rc == DHQR_OK) {
      for (int64_t hi = n; hi > 0; hi -= BS_NB) {  // src:244-282
        const int64_t lo = std::max<int64_t>(0, hi - BS_NB);
        hipLaunchKernelGGL(k_backsub_diag, dim3(1), dim3(64), 0, c->stream, dA, lda, dalpha, db, lo, hi);
        if (lo > 0)
          hipLaunchKernelGGL(k_backsub_update, dim3((unsigned)((lo + 255) / 256)), dim3(256), 0,
                             c->stream, dA, lda, db, lo, hi);
      }
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
  ENTER(c);
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
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  if (small_ldiv_fit(c, m, n)) {  // (same form as dhqr_qr_f64's small route: the kernel works on the pinned staging buffer)
    const size_t na = (size_t)m * (size_t)n;
    CHECK(small_pin_ensure(c, na + 2 * (size_t)n + (size_t)m));
    double *pA = c->small_pin, *pal = pA + na, *pb = pal + n, *px = pb + m;
    copy_cols(pA, m, hA, lda, m, n);
    memcpy(pal, halpha, (size_t)n * sizeof(double));
    memcpy(pb, hb, (size_t)m * sizeof(double));  // src:318 copy of b
    // the kernel first brings the factor from the pinned buffer into device memory (its chunk pipeline would otherwise pay
    // a PCIe round trip per chunk)
    CHECK(ensure(c, c->small_dev, (size_t)SML_LDR * SML_LDR));
    unsigned long long *done = reinterpret_cast<unsigned long long *>(c->small_pin + c->small_pin_cap);
    CHECK(small_ldiv_launch(c, pA, m, m, n, pal, pb, pb, px, c->small_dev.p, done, ++c->small_epoch));
    CHECK(small_wait_done(c, done, c->small_epoch));
    memcpy(hx, px, (size_t)n * sizeof(double));  // src:320
    return DHQR_OK;
  }
  // the device copy of the context (shared with dhqr_qr_f64: no 8 GiB hipMalloc / hipFree per call) and the staged upload
  // of dhqr_hostio.h; the factor is uploaded again -- the caller may have changed it since qr! (H.A is the caller's memory)
  const int64_t ldd = (m + 1) & ~(int64_t)1;
  const size_t mat = ((size_t)ldd * (size_t)n + 1) & ~(size_t)1;
  CHECK(ensure(c, c->host_mat, mat + (size_t)n + (size_t)m + 32));
  double *dA = c->host_mat.p, *dal = dA + mat, *db = dal + ((n + 1) & ~(int64_t)1);
  if (c->tc_A == dA) c->tc_valid = false;  // the caller's factor is uploaded afresh: nothing kept applies to it
  // the same switch and the same default as dhqr_qr_f64: the plain copy unless DHQR_HOSTIO=1
  static const bool overlap = [] { const char *e = getenv("DHQR_HOSTIO"); return e && atoi(e) == 1; }();
  auto body = [&]() -> int32_t {
    bool staged = overlap;
    if (staged) {
      if (!c->hio) c->hio = new HostIo();
      if (hio_init(*c->hio, ldd, n) != DHQR_OK) staged = false;
    }
    if (staged) {
      CHECK(hio_upload(*c->hio, hA, m, n, lda, dA, ldd, c->stream));
    } else {
      HIPCHECK(hipMemcpy2DAsync(dA, ldd * sizeof(double), hA, lda * sizeof(double), m * sizeof(double),
                                n, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHECK(hipMemcpyAsync(dal, halpha, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(db, hb, m * sizeof(double), hipMemcpyHostToDevice, c->stream));  // src:318 copy of b
    CHECK(dhqr_solve_f64(c, dA, m, n, ldd, dal, db));
    HIPCHECK(hipStreamSynchronize(c->stream));
    CHECK(pipe_error_check(c));  // a synchronous entry point reports an expired inter-workgroup wait itself (dhqr.h), after
                                 // repeating a solve whose persistent kernel could not get all its workgroups resident
    HIPCHECK(hipMemcpyAsync(hx, db, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));  // src:320
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  if (c->hio) (void)hio_drain(*c->hio);
  return rc;
}

int32_t dhqr_partialdot_f64(dhqr_ctx *c, const double *da, const double *db, int64_t lo, int64_t hi,
                            double *hout) {
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_zptr(dA, "matrix"));
  CHECK(check_zptr(dalpha, "alpha"));
  double2 *A = reinterpret_cast<double2 *>(dA), *al = reinterpret_cast<double2 *>(dalpha);
  if (c->zpipe && n <= 128 && m <= 8192) {  // a panel: one column-pipelined launch (k_zpanel_pipe, dhqr_complex.h)
    const int epoch = ++c->zepoch;
    CHECK(prof_begin(c, CAT_RANK1));
#define DHQR_ZPP(T_, E_, G_) hipLaunchKernelGGL((k_zpanel_pipe<T_, E_, G_>), dim3((unsigned)((n + G_ - 1) / G_)), dim3(T_), 0, c->stream, A, lda, m, (int)n, al, c->zflags, epoch)
    // ONE column per workgroup: measured (profiles/r03_c64_panel_pipeline.txt), several columns per workgroup make the
    // hand-overs rarer but serialise their dot / update rounds on one CU's FP64 pipes -- slower at every height
    if (m <= 512) DHQR_ZPP(256, 2, 1);
    else if (m <= 1024) DHQR_ZPP(256, 4, 1);
    else if (m <= 2048) DHQR_ZPP(256, 8, 1);
    else if (m <= 4096) DHQR_ZPP(512, 8, 1);
    else DHQR_ZPP(1024, 8, 1);
#undef DHQR_ZPP
    CHECK(prof_end(c));
    if (c->profiling)
      for (int64_t j = 0; j + 1 < n; ++j) c->st.bytes_rank1 += 32.0 * (double)(m - j) * (double)(n - j - 1);
    LAUNCHCHECK();
    return DHQR_OK;
  }
  const size_t vlen = (size_t)((m + 15) & ~(int64_t)15);  // complex elements per staging vector
  CHECK(ensure(c, c->vbuf, 4 * vlen));
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

// One ComplexF64 panel: factor (src:122-148,171-213 inside the panel) and, when `build` is set, embed its reflectors as a
// real 2 rows x 128 operand and build T (the last panel is applied to nothing: build = false).
static int32_t zpanel_make(dhqr_ctx *c, double *P, int64_t rows, int64_t w, int64_t lda, double *al, const PanelBuf &pb, bool build) {
  CHECK(dhqr_factor_c64(c, P, rows, w, lda, al));
  if (!build) return DHQR_OK;
  const int64_t npad = panel_ldv(2 * rows);
  CHECK(prof_begin(c, CAT_TBUILD));
  {
    dim3 grid((unsigned)std::min<int64_t>((npad / 2 + 255) / 256, 64), (unsigned)DHQR_ZNB);
    hipLaunchKernelGGL(k_zpack_emb, grid, dim3(256), 0, c->stream, reinterpret_cast<const double2 *>(P), lda, rows, (int)w,
                       pb.V, pb.ldv, npad);
  }
  CHECK(panel_build_t(c, 2 * rows, -2 * w, pb));  // negative: strict upper part at the 2 x 2 block level
  CHECK(prof_end(c));
  return DHQR_OK;
}

// Blocked ComplexF64 factorisation: panels of DHQR_ZNB = 64 complex columns are factored by the unblocked complex
// kernels, their block reflector is applied to the trailing matrix by the Float64 MFMA kernels through the real
// embedding (dhqr_complex.h).  nb = 0: unblocked (dhqr_factor_c64), nb = 64: blocked.
int32_t dhqr_factor_c64_nb(dhqr_ctx *c, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha, int32_t nb) {
  if (nb == 0) return dhqr_factor_c64(c, dA, m, n, lda, dalpha);
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  if (nb != DHQR_ZNB) return set_err(DHQR_EINVAL, "ComplexF64: nb must be 0 (unblocked) or %d (blocked); got %d", DHQR_ZNB, nb);
  CHECK(check_mat(dA, m, n, 2 * lda, false));  // the MFMA kernels see the interleaved storage as a real matrix with ld = 2 lda
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_zptr(dA, "matrix"));
  CHECK(check_zptr(dalpha, "alpha"));
  const int64_t ZB = DHQR_ZNB;
  CHECK(ensure(c, c->vt, 2 * (size_t)panel_elems(2 * m)));  // two panel operand buffers (look-ahead: panel k + 1 is built while k is applied)
  double *vtb[2] = {c->vt.p, c->vt.p + panel_elems(2 * m)};
  auto make_panel = [&](int64_t c0, const PanelBuf &pb) -> int32_t {
    const int64_t w = std::min<int64_t>(ZB, n - c0);
    return zpanel_make(c, dA + 2 * (c0 + c0 * lda), m - c0, w, lda, dalpha + 2 * c0, pb, n - c0 - w > 0);
  };
  // Look-ahead (DHQR_LOOKAHEAD=0 or per-column panels: plain loop).  With one launch per COLUMN a lane was slower than
  // no lane (profiles/r02_ab_c64_blocked_lookahead.txt: every launch waited for a CU behind the wide update's GEMM
  // workgroups); the column-pipelined panel kernel is ONE launch per panel, waits once, and then holds its <= 64 CUs
  // while the wide update runs on the others.
  // Panels taller than the pipelined kernel's 8192 rows are factored one launch per column: no lane for them (plain loop);
  // the lane takes over at the first panel whose SUCCESSOR fits.
  const bool lane = c->lookahead && c->zpipe && n > 2 * ZB;
  int64_t cstart = 0;  // first panel of the look-ahead phase
  while (cstart < n && (!lane || m - cstart - ZB > 8192)) {
    const int64_t w = std::min<int64_t>(ZB, n - cstart), rows = m - cstart;
    const PanelBuf pb = vt_view(vtb[0], 2 * rows);
    CHECK(make_panel(cstart, pb));
    const int64_t ncols = n - cstart - w;
    if (ncols > 0) CHECK(panel_apply(c, pb, 2 * rows, dA + 2 * (cstart + (cstart + w) * lda), ncols, 2 * lda, 1));
    cstart += ZB;
  }
  if (cstart < n && n - cstart <= ZB) {  // one panel left: applied to nothing
    CHECK(make_panel(cstart, vt_view(vtb[0], 2 * (m - cstart))));
    cstart = n;
  }
  if (cstart >= n) {
    LAUNCHCHECK();
    return DHQR_OK;
  }
  // lane (high priority): panel k -> the columns of panel k + 1, then panel k + 1;  caller's stream: panel k -> beyond
  if (!c->zev[0])
    for (int i = 0; i < 5; ++i) HIPCHECK(hipEventCreateWithFlags(&c->zev[i], hipEventDisableTiming));
  hipEvent_t evP[2] = {c->zev[0], c->zev[1]}, evW[2] = {c->zev[2], c->zev[3]}, evS = c->zev[4];
  hipStream_t sW = c->stream, sL = c->hi;
  auto on = [&](hipStream_t st, int wsi) { c->stream = st; c->cur_ws = wsi; };
  {  // the workspaces of both streams, sized up front: nothing is (re)allocated while the two streams run
    const size_t NN = (size_t)DHQR_NBV * DHQR_NBV, ncmax = (size_t)std::max<int64_t>(n, 2 * DHQR_NBV);
    for (int wsi = 0; wsi < 2; ++wsi) {
      CHECK(ensure(c, c->ws[wsi].w1, NN * (wsi == 0 ? 2 * ((ncmax + 127) / 128) + 2600 : 520)));  // split-K partials: pick_split stops at 4 x 512 workgroups
      CHECK(ensure(c, c->ws[wsi].w1r, (size_t)DHQR_NBV * ncmax));
      CHECK(ensure(c, c->ws[wsi].w2, (size_t)DHQR_NBV * ncmax));
    }
    CHECK(ensure(c, c->spart, 256 * NN));
    CHECK(ensure(c, c->sfull, NN));
  }
  auto body = [&]() -> int32_t {
    HIPCHECK(hipEventRecord(evS, sW));
    HIPCHECK(hipStreamWaitEvent(sL, evS, 0));
    on(sL, 1);
    CHECK(make_panel(cstart, vt_view(vtb[0], 2 * (m - cstart))));
    HIPCHECK(hipEventRecord(evP[0], sL));
    int64_t k = 0;
    for (int64_t c0 = cstart; c0 < n; c0 += ZB, ++k) {
      const int64_t w = std::min<int64_t>(ZB, n - c0), rows = m - c0;
      const int64_t c1 = c0 + w;  // first column of panel k + 1
      if (c1 >= n) break;
      const int64_t w1 = std::min<int64_t>(ZB, n - c1);
      const PanelBuf pb = vt_view(vtb[k & 1], 2 * rows);
      // lane: the columns of panel k + 1 carry every panel before k once the wide update of k - 1 has finished
      on(sL, 1);
      if (k >= 1) HIPCHECK(hipStreamWaitEvent(sL, evW[(k - 1) & 1], 0));
      CHECK(panel_apply(c, pb, 2 * rows, dA + 2 * (c0 + c1 * lda), w1, 2 * lda, 1));
      CHECK(make_panel(c1, vt_view(vtb[(k + 1) & 1], 2 * (m - c1))));  // its buffer was last read by the wide update of k - 1
      HIPCHECK(hipEventRecord(evP[(k + 1) & 1], sL));
      // caller's stream: panel k -> everything beyond panel k + 1
      on(sW, 0);
      HIPCHECK(hipStreamWaitEvent(sW, evP[k & 1], 0));
      const int64_t ncols = n - c1 - w1;
      if (ncols > 0) CHECK(panel_apply(c, pb, 2 * rows, dA + 2 * (c0 + (c1 + w1) * lda), ncols, 2 * lda, 1));
      HIPCHECK(hipEventRecord(evW[k & 1], sW));
    }
    on(sW, 0);
    HIPCHECK(hipEventRecord(evS, sL));
    HIPCHECK(hipStreamWaitEvent(sW, evS, 0));  // join: the caller's stream owns the result
    return DHQR_OK;
  };
  const int32_t rc = body();
  on(sW, 0);
  CHECK(rc);
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_qr_c64_nb(dhqr_ctx *c, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha, int32_t nb) {
  ENTER(c);
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
    CHECK(dhqr_factor_c64_nb(c, dA, m, n, m, dal, nb));
    HIPCHECK(hipMemcpy2DAsync(hA, lda * esz, dA, m * esz, m * esz, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(halpha, dal, n * esz, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return pipe_error_check(c);
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dal);
  return rc;
}

int32_t dhqr_qr_c64(dhqr_ctx *c, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha) {
  ENTER(c);
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
    return pipe_error_check(c);
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dal);
  return rc;
}

int32_t dhqr_solve_c64(dhqr_ctx *c, const double *dA, int64_t m, int64_t n, int64_t lda,
                       const double *dalpha, double *db) {
  ENTER(c);
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_zptr(dA, "matrix"));
  CHECK(check_zptr(dalpha, "alpha"));
  CHECK(check_zptr(db, "b"));
  const double2 *A = reinterpret_cast<const double2 *>(dA), *al = reinterpret_cast<const double2 *>(dalpha);
  double2 *b = reinterpret_cast<double2 *>(db);
  // b carried in double-double (dhqr_complex.h, "the solve with b carried in double-double"): with the plain-double solve
  // the reference's acceptance statistic scored like the reference itself, up to 14 x LAPACK (round 4), so that path is gone
  CHECK(ensure(c, c->zsolve_lo, (size_t)(2 * m + 16)));
  double2 *bl = reinterpret_cast<double2 *>(c->zsolve_lo.p);
  HIPCHECK(hipMemsetAsync(bl, 0, (size_t)m * sizeof(double2), c->stream));
  CHECK(prof_begin(c, CAT_SOLVE));
  for (int64_t j = 0; j < n; ++j) {  // src:215-224: reflectors in column order
    if (m - j <= 2048)
      hipLaunchKernelGGL((k_zqtb_col_dd<256>), dim3(1), dim3(256), 0, c->stream, A + j * lda, b, bl, m, j);
    else
      hipLaunchKernelGGL((k_zqtb_col_dd<1024>), dim3(1), dim3(1024), 0, c->stream, A + j * lda, b, bl, m, j);
  }
  for (int64_t hi = n; hi > 0; hi -= ZBS_NB) {  // src:244-254
    const int64_t lo = std::max<int64_t>(0, hi - ZBS_NB);
    hipLaunchKernelGGL(k_zbacksub_diag_dd, dim3(1), dim3(64), 0, c->stream, A, lda, al, b, bl, lo, hi);
    if (lo > 0)
      hipLaunchKernelGGL(k_zbacksub_update_dd, dim3((unsigned)((lo + 255) / 256)), dim3(256), 0, c->stream, A, lda, b, bl, lo, hi);
  }
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

int32_t dhqr_ldiv_c64(dhqr_ctx *c, const double *hA, int64_t m, int64_t n, int64_t lda,
                      const double *halpha, const double *hb, double *hx) {
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
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
  ENTER(c);
  CHECK(check_mat(dA, m, n, lda, true));
  CHECK(check_mat(dB, m, nrhs, ldb, false));
  return apply_q_impl(c, dA, m, n, lda, nullptr, dB, nrhs, ldb, trans ? 1 : 0, false);
}

int32_t dhqr_residual_f64(dhqr_ctx *c, const double *dAfact, int64_t m, int64_t n, int64_t lda,
                          const double *dalpha, const double *dAorig, int64_t ldo, double *dwork,
                          double *hrel) {
  ENTER(c);
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
  ENTER(c);
  CHECK(check_mat(dP, rows, ncols, ldp, true));
  if (ncols > DHQR_NB) return set_err(DHQR_EINVAL, "panel wider than %d", DHQR_NB);
  if (!dVT || !aligned16(dVT)) return set_err(DHQR_EINVAL, "dVT must be a 16-byte aligned device buffer");
  // alpha is produced in a small scratch vector and packed into the tail of dVT by factor_panel
  CHECK(ensure(c, c->scratch, 4096));
  double *al = c->scratch.p + 3072;
  return factor_panel_sync(c, dP, rows, ncols, ldp, al, vt_view(dVT, rows));
}

int32_t dhqr_panel_pack_f64(dhqr_ctx *c, const double *dP, int64_t rows, int64_t ncols, int64_t ldp,
                            double *dVT) {
  ENTER(c);
  CHECK(check_mat(dP, rows, ncols, ldp, true));
  if (ncols > DHQR_NB) return set_err(DHQR_EINVAL, "panel wider than %d", DHQR_NB);
  if (!dVT || !aligned16(dVT)) return set_err(DHQR_EINVAL, "dVT must be a 16-byte aligned device buffer");
  const PanelBuf pb = vt_view(dVT, rows);
  HIPCHECK(hipMemsetAsync(pb.alpha, 0, (DHQR_NBV + DHQR_STATW) * sizeof(double), c->stream));
  return panel_pack_and_t(c, dP, rows, ncols, ldp, nullptr, pb);
}

int32_t dhqr_form_r0_f64(dhqr_ctx *c, const double *dA, int64_t m, int64_t cols, int64_t lda,
                         const double *dalpha, double *dW, int64_t ldw, int64_t colblock,
                         int32_t nranks, int32_t rank) {
  ENTER(c);
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
  ENTER(c);
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

int32_t dhqr_panel_apply_f64(dhqr_ctx *c, const double *dVT, int64_t rows, double *dC, int64_t ncols,
                             int64_t ldc, int32_t trans) {
  ENTER(c);
  if (!dVT || !aligned16(dVT)) return set_err(DHQR_EINVAL, "dVT must be a 16-byte aligned device buffer");
  if (ncols == 0) return DHQR_OK;
  CHECK(check_mat(dC, rows, ncols, ldc, false));
  return panel_apply(c, vt_view(dVT, rows), rows, dC, ncols, ldc, trans ? 1 : 0);
}

#ifdef DHQR_BENCH_BUILD
}  // extern "C"
#include "dhqr_bench.h"  // micro-benchmarks and the MFMA layout probe: libdhqr_bench.so only (include/dhqr_bench.h)
extern "C" {
#endif

// ============================================================ multi-GPU: communicators (dhqr_comm.h)
// Second channel for the row-split look-ahead lane (its small latency-bound collectives overtake the wide stream's large
// all-reduce).  Default: ON for the LOCAL transport (a second mailbox; measured), OFF for RCCL -- two communicators whose
// kernels compete for CUs the persistent GEMMs hold have never run on hardware with more than one rank, so the first
// multi-GPU run uses ONE ordered channel; DHQR_LANE_CHANNEL=1 turns the second RCCL communicator on, =0 forces it off
// for every transport.
static bool lane_channel_wanted(int kind) {
  if (const char *e = getenv("DHQR_LANE_CHANNEL")) return atoi(e) != 0;
  return kind != COMM_RCCL;
}
// a context that drives a rank of a multi-GPU RCCL job keeps one CU per XCD out of its persistent launches (see spare_cus)
static void comm_bind_rccl(dhqr_ctx *c, int kind, int nranks) {
  if (kind == COMM_RCCL && nranks > 1 && !c->spare_cus_set) c->spare_cus = std::min(8, std::max(0, c->ncu - 8));
}
static int32_t comm_new(dhqr_comm **out, dhqr_ctx *c, int kind, int nranks, int rank) {
  comm_bind_rccl(c, kind, nranks);
  dhqr_comm *cm = new dhqr_comm();
  cm->ctx = c;
  cm->kind = kind;
  cm->nranks = nranks;
  cm->rank = rank;
  {  // doubles; broadcasts below it stay single ncclBroadcast calls
    long long v;
    if (tune_get("sag_min", &v) && v > 0) cm->bcast_sag_min = v;
  }
  *out = cm;
  return DHQR_OK;
}

int32_t dhqr_comm_unique_id(void *id128) {
  if (!id128) return set_err(DHQR_EINVAL, "null id buffer");
  CHECK(rccl_load());
  ncclUniqueId id;
  RCCLCHECK(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == DHQR_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return DHQR_OK;
}

int32_t dhqr_comm_create_rank(dhqr_comm **out, dhqr_ctx *c, int32_t nranks, int32_t rank, const void *id128) {
  if (!out) return set_err(DHQR_EINVAL, "null comm out-pointer");
  *out = nullptr;
  ENTER(c);
  if (nranks < 1 || nranks > DHQR_MAX_RANKS || rank < 0 || rank >= nranks)
    return set_err(DHQR_EINVAL, "bad rank %d of %d", rank, nranks);
  if (nranks == 1) return comm_new(out, c, COMM_SELF, 1, 0);
  if (!id128) return set_err(DHQR_EINVAL, "null unique id");
  CHECK(rccl_load());
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  // everything after the first ncclCommInitRank runs inside `build`: on any failure the communicators created so far
  // are destroyed and *out stays null (a half-built handle never reaches the caller)
  ncclComm_t nc = nullptr, nc2 = nullptr;
  dhqr_comm *cm = nullptr;
  void *dbuf = nullptr;
  auto build = [&]() -> int32_t {
    RCCLCHECK(g_rccl.CommInitRank(&nc, nranks, id, rank));
    if (g_rccl.CommCount) {
      int cnt = 0;
      RCCLCHECK(g_rccl.CommCount(nc, &cnt));
      if (cnt != nranks) return set_err(DHQR_ECOMM, "RCCL communicator has %d ranks, %d requested", cnt, nranks);
    }
    CHECK(comm_new(&cm, c, COMM_RCCL, nranks, rank));
    cm->nccl = nc;
    nc = nullptr;  // owned by cm from here
    // second channel for the look-ahead lane of the row-split driver: rank 0 draws another unique id and ships it over
    // the first communicator (no change for the host layer)
    if (lane_channel_wanted(COMM_RCCL)) {
      ncclUniqueId id2;
      memset(&id2, 0, sizeof(id2));
      if (rank == 0) RCCLCHECK(g_rccl.GetUniqueId(&id2));
      HIPCHECK(hipMalloc(&dbuf, sizeof(id2)));
      HIPCHECK(hipMemcpyAsync(dbuf, &id2, sizeof(id2), hipMemcpyHostToDevice, c->stream));
      RCCLCHECK(g_rccl.Broadcast(dbuf, dbuf, sizeof(id2), ncclChar, 0, cm->nccl, c->stream));
      HIPCHECK(hipMemcpyAsync(&id2, dbuf, sizeof(id2), hipMemcpyDeviceToHost, c->stream));
      HIPCHECK(hipStreamSynchronize(c->stream));
      RCCLCHECK(g_rccl.CommInitRank(&nc2, nranks, id2, rank));
      CHECK(comm_new(&cm->lane, c, COMM_RCCL, nranks, rank));
      cm->lane->nccl = nc2;
      nc2 = nullptr;
    }
    return comm_tune_bcast(cm, c->stream);  // ring broadcast vs scatter + all-gather, measured on this node
  };
  const int32_t rc = build();
  if (dbuf) (void)hipFree(dbuf);
  if (rc != DHQR_OK) {
    if (nc) (void)g_rccl.CommDestroy(nc);
    if (nc2) (void)g_rccl.CommDestroy(nc2);
    if (cm) (void)comm_free(cm);  // destroys cm->nccl and the lane
    return rc;
  }
  *out = cm;
  return DHQR_OK;
}

int32_t dhqr_comm_create_callbacks(dhqr_comm **out, dhqr_ctx *c, int32_t nranks, int32_t rank, dhqr_bcast_fn bcast,
                                   dhqr_allreduce_fn allreduce, void *user) {
  if (!out) return set_err(DHQR_EINVAL, "null comm out-pointer");
  *out = nullptr;
  ENTER(c);
  if (nranks < 1 || nranks > DHQR_MAX_RANKS || rank < 0 || rank >= nranks)
    return set_err(DHQR_EINVAL, "bad rank %d of %d", rank, nranks);
  if (nranks > 1 && (!bcast || !allreduce)) return set_err(DHQR_EINVAL, "null callback");
  CHECK(comm_new(out, c, nranks == 1 ? COMM_SELF : COMM_CALLBACK, nranks, rank));
  (*out)->cb_bcast = bcast;
  (*out)->cb_allreduce = allreduce;
  (*out)->cb_user = user;
  return DHQR_OK;
}

int32_t dhqr_comm_destroy(dhqr_comm *cm) { return comm_free(cm); }

int32_t dhqr_comm_info(dhqr_comm *cm, int32_t *kind, int32_t *nranks, int32_t *rank, int64_t *bytes_bcast) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (kind) *kind = cm->kind;
  if (nranks) *nranks = cm->nranks;
  if (rank) *rank = cm->rank;
  if (bytes_bcast) *bytes_bcast = cm->bytes_bcast;
  return DHQR_OK;
}

int32_t dhqr_comm_counters(dhqr_comm *cm, int64_t *out4) {
  if (!cm || !out4) return set_err(DHQR_EINVAL, "null pointer argument");
  out4[0] = cm->n_bcast;
  out4[1] = cm->bytes_bcast;
  out4[2] = cm->n_allreduce + (cm->lane ? cm->lane->n_allreduce : 0);
  out4[3] = cm->bytes_allreduce + (cm->lane ? cm->lane->bytes_allreduce : 0);
  return DHQR_OK;
}

// Device time this rank spent in its collectives (hipEvent pairs around every broadcast / all-reduce, the wait for the
// peers included) since the last call or since timing was switched on: out4 = {broadcast ms, broadcasts timed,
// all-reduce ms, all-reduces timed}, the row-split lane's channel included.  on = 1 / 0 starts / stops collecting, -1
// leaves it as it is.  Synchronises the device.
int32_t dhqr_comm_timing(dhqr_comm *cm, int32_t on, double *out4) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  ENTER(cm->ctx);  // the rank's device for the duration of the call; the caller's current device is restored on return
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  dhqr_comm *chs[2] = {cm, cm->lane};
  for (dhqr_comm *ch : chs) {
    if (!ch) continue;
    if (ch->tev_used > 0) HIPCHECK(hipDeviceSynchronize());
    for (size_t i = 0; i < ch->tev_used; ++i) {
      float t = 0.f;
      HIPCHECK(hipEventElapsedTime(&t, ch->tev[i].a, ch->tev[i].b));
      acc[2 * ch->tev[i].kind] += (double)t;
      acc[2 * ch->tev[i].kind + 1] += 1.0;
    }
    ch->tev_used = 0;
    if (on >= 0) ch->timing = on != 0;
  }
  if (out4)
    for (int i = 0; i < 4; ++i) out4[i] = acc[i];
  return DHQR_OK;
}

int32_t dhqr_comm_rccl_nranks(dhqr_comm *cm, int32_t *main_channel, int32_t *lane_channel) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  auto count = [&](dhqr_comm *x, int32_t *o) -> int32_t {
    if (!o) return DHQR_OK;
    *o = 0;  // 0: not an RCCL channel (or no ncclCommCount in this RCCL build)
    if (!x || x->kind != COMM_RCCL || !x->nccl || !g_rccl.CommCount) return DHQR_OK;
    int n = 0;
    RCCLCHECK(g_rccl.CommCount(x->nccl, &n));
    *o = n;
    return DHQR_OK;
  };
  CHECK(count(cm, main_channel));
  return count(cm->lane, lane_channel);
}

int32_t dhqr_comm_get_bcast_tuning(dhqr_comm *cm, int32_t *algo, double *ms_ring, double *ms_sag) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (algo) *algo = cm->bcast_algo;
  if (ms_ring) *ms_ring = cm->tune_ms[0];
  if (ms_sag) *ms_sag = cm->tune_ms[1];
  return DHQR_OK;
}

// ============================================================ multi-GPU: SPMD column-split entry points
int64_t dhqr_cs_local_cols(int64_t n, int32_t nranks, int32_t rank) {
  if (n < 0 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  return cs_local_cols(n, nranks, rank);
}
void dhqr_cs_contiguous_range(int64_t n, int32_t nranks, int32_t rank, int64_t *lo, int64_t *hi) {
  cs_contig_range(n, nranks, rank, lo, hi);
}

static int32_t cs_check(dhqr_comm *cm, const void *dA, int64_t m, int64_t n, int64_t lda, CsProblem *pr, bool need_A = true) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  pr->c = cm->ctx;
  pr->cm = cm;
  pr->P = cm->nranks;
  pr->r = cm->rank;
  pr->m = m;
  pr->n = n;
  pr->lda = lda;
  pr->K = cs_nblocks(n);
  pr->ncl = cs_local_cols(n, pr->P, pr->r);
  pr->A = const_cast<double *>((const double *)dA);
  pr->alpha = nullptr;
  if (need_A && pr->ncl > 0) CHECK(check_mat(dA, m, pr->ncl, lda, false));
  return DHQR_OK;
}

int32_t dhqr_cs_fill_uniform_f64(dhqr_comm *cm, double *dA, int64_t m, int64_t n, int64_t lda, uint64_t seed) {
  CsProblem pr;
  CHECK(cs_check(cm, dA, m, n, lda, &pr));
  if (pr.ncl == 0) return DHQR_OK;
  return dhqr_fill_uniform_f64(pr.c, dA, m, pr.ncl, lda, seed, m, 0, CS_CB, pr.P, pr.r);
}

int32_t dhqr_cs_factor_f64(dhqr_comm *cm, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  CsProblem pr;
  CHECK(cs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dalpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  pr.alpha = dalpha;
  return cs_factor(pr);
}

int32_t dhqr_cs_residual_f64(dhqr_comm *cm, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha,
                             uint64_t seed, double *dW, double *dA0, double *hrel) {
  CsProblem pr;
  CHECK(cs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dalpha || !hrel || (pr.ncl > 0 && (!dW || !dA0))) return set_err(DHQR_EINVAL, "null pointer argument");
  pr.alpha = const_cast<double *>(dalpha);
  return cs_residual(pr, seed, dW, dA0, hrel);
}

int32_t dhqr_cs_solve_f64(dhqr_comm *cm, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha,
                          double *db, double *dwork) {
  CsProblem pr;
  CHECK(cs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dalpha || !db || !dwork) return set_err(DHQR_EINVAL, "null pointer argument");
  pr.alpha = const_cast<double *>(dalpha);
  return cs_solve(pr, db, dwork);
}

int32_t dhqr_cs_load_contiguous_f64(dhqr_comm *cm, double *dA, int64_t m, int64_t n, int64_t lda, const double *dBlock,
                                    int64_t ldb, double *dstage) {
  CsProblem pr;
  CHECK(cs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dstage) return set_err(DHQR_EINVAL, "null staging buffer");
  return cs_load_contiguous(pr, dBlock, ldb, dstage);
}
int32_t dhqr_cs_store_contiguous_f64(dhqr_comm *cm, const double *dA, int64_t m, int64_t n, int64_t lda, double *dBlock,
                                     int64_t ldb, double *dstage) {
  CsProblem pr;
  CHECK(cs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dstage) return set_err(DHQR_EINVAL, "null staging buffer");
  return cs_store_contiguous(pr, dBlock, ldb, dstage);
}

// qr!(A::DArray) (src:115-120, 311-315) for ONE process of a multi-process job: hBlock is this process's
// contiguous column block of the m x n matrix (DistributedArrays' default split, dhqr_cs_contiguous_range), on
// the HOST; it is overwritten with the factored columns; halpha (n) receives the replicated alpha (the
// reference's SharedArray, src:301-304).  Collective over the communicator.
int32_t dhqr_cs_qr_darray_f64(dhqr_comm *cm, double *hBlock, int64_t m, int64_t n, int64_t ldb, double *halpha) {
  CsProblem pr;
  CHECK(cs_check(cm, nullptr, m, n, m, &pr, false));
  ENTER(pr.c);
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  int64_t lo, hi;
  cs_contig_range(n, pr.P, pr.r, &lo, &hi);
  const int64_t wr = hi - lo, wmax = n / pr.P + 1;
  if (wr > 0 && (!hBlock || ldb < m)) return set_err(DHQR_EINVAL, "bad local block");
  const int64_t ldd = (m + 1) & ~(int64_t)1;
  double *dA = nullptr, *dBlk = nullptr, *dStage = nullptr, *dal = nullptr;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&dA, (size_t)ldd * std::max<int64_t>(pr.ncl, 1) * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dBlk, (size_t)m * std::max<int64_t>(wr, 1) * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dStage, (size_t)m * std::max<int64_t>(wmax, DHQR_NBV) * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dal, (size_t)n * sizeof(double)));
    pr.A = dA;
    pr.lda = ldd;
    pr.alpha = dal;
    dhqr_ctx *c = pr.c;
    if (wr > 0)
      HIPCHECK(hipMemcpy2DAsync(dBlk, m * sizeof(double), hBlock, ldb * sizeof(double), m * sizeof(double), wr,
                                hipMemcpyHostToDevice, c->stream));
    CHECK(cs_load_contiguous(pr, dBlk, m, dStage));
    CHECK(cs_factor(pr));
    CHECK(cs_store_contiguous(pr, dBlk, m, dStage));
    if (wr > 0)
      HIPCHECK(hipMemcpy2DAsync(hBlock, ldb * sizeof(double), dBlk, m * sizeof(double), m * sizeof(double), wr,
                                hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(halpha, dal, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  const int32_t rc = body();
  (void)hipDeviceSynchronize();
  double *ps[] = {dA, dBlk, dStage, dal};
  for (double *p : ps)
    if (p) (void)hipFree(p);
  return rc;
}

// `H \ b` (src:317-321 with src:226-230, 256-270) for ONE process of a multi-process job holding a factored DArray: hBlock is
// this process's contiguous column block of the FACTORED m x n matrix (what dhqr_cs_qr_darray_f64 left there), halpha the
// replicated alpha, hb the right-hand side (m, the same on every process: the reference's SharedArray copy of b, src:318);
// hx (n) receives x on every process.  Nothing is modified on the host.  The block is moved into the block-cyclic layout
// and solved by cs_solve: one broadcast of b's tail per panel for Q'b, one all-reduce of the partial dots + one broadcast
// of the solved block per panel for the back substitution (instead of the reference's n x np scalar RPCs, src:260-267).
int32_t dhqr_cs_ldiv_darray_f64(dhqr_comm *cm, const double *hBlock, int64_t m, int64_t n, int64_t ldb, const double *halpha,
                                const double *hb, double *hx) {
  CsProblem pr;
  CHECK(cs_check(cm, nullptr, m, n, m, &pr, false));
  ENTER(pr.c);
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  int64_t lo, hi;
  cs_contig_range(n, pr.P, pr.r, &lo, &hi);
  const int64_t wr = hi - lo, wmax = n / pr.P + 1;
  if (wr > 0 && (!hBlock || ldb < m)) return set_err(DHQR_EINVAL, "bad local block");
  const int64_t ldd = (m + 1) & ~(int64_t)1, mpad = (m + 15) & ~(int64_t)15;
  double *dA = nullptr, *dBlk = nullptr, *dStage = nullptr, *dal = nullptr, *dvec = nullptr;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&dA, (size_t)ldd * std::max<int64_t>(pr.ncl, 1) * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dBlk, (size_t)m * std::max<int64_t>(wr, 1) * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dStage, (size_t)m * std::max<int64_t>(wmax, DHQR_NBV) * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dal, (size_t)n * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dvec, (size_t)(2 * mpad + 2 * DHQR_NBV + 16) * sizeof(double)));
    pr.A = dA;
    pr.lda = ldd;
    pr.alpha = dal;
    dhqr_ctx *c = pr.c;
    if (wr > 0)
      HIPCHECK(hipMemcpy2DAsync(dBlk, m * sizeof(double), hBlock, ldb * sizeof(double), m * sizeof(double), wr,
                                hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(dal, halpha, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(dvec, hb, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));  // src:318 copy of b
    CHECK(cs_load_contiguous(pr, dBlk, m, dStage));
    CHECK(cs_solve(pr, dvec, dvec + mpad));
    HIPCHECK(hipMemcpyAsync(hx, dvec, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));  // src:320
    HIPCHECK(hipStreamSynchronize(c->stream));
    return pipe_error_check(c);  // (the solve of dhqr_qtb.h has bounded inter-workgroup waits)
  };
  const int32_t rc = body();
  (void)hipDeviceSynchronize();
  double *ps[] = {dA, dBlk, dStage, dal, dvec};
  for (double *p : ps)
    if (p) (void)hipFree(p);
  return rc;
}

// ============================================================ multi-GPU: single-process handle (dhqr_mg.h)
int32_t dhqr_mg_create(dhqr_mg **out, const int32_t *devices, int32_t ndev) {
  if (!out) return set_err(DHQR_EINVAL, "null out-pointer");
  *out = nullptr;
  if (ndev < 1 || ndev > DHQR_MAX_RANKS) return set_err(DHQR_EINVAL, "ndev must be in [1, %d]", DHQR_MAX_RANKS);
  int prev = -1;
  (void)hipGetDevice(&prev);
  dhqr_mg *g = new dhqr_mg();
  g->ndev = ndev;
  g->rk.resize(ndev);
  g->rc.assign(ndev, DHQR_OK);
  g->err.assign(ndev, "");
  for (int r = 0; r < ndev; ++r) g->dev.push_back(devices ? devices[r] : r);
  auto init = [&]() -> int32_t {
    for (int r = 0; r < ndev; ++r) CHECK(dhqr_create(&g->rk[r].c, g->dev[r]));
    bool distinct = true;
    for (int a = 0; a < ndev; ++a)
      for (int b = a + 1; b < ndev; ++b)
        if (g->dev[a] == g->dev[b]) distinct = false;
    int want = COMM_RCCL;  // the product default: RCCL over xGMI
    if (const char *e = getenv("DHQR_TRANSPORT")) {
      if (!strcmp(e, "local")) want = COMM_LOCAL;
      else if (!strcmp(e, "rccl")) want = COMM_RCCL;
    }
    if (ndev == 1) want = COMM_SELF;
    else if (!distinct) want = COMM_LOCAL;  // RCCL cannot put two ranks on one device
    std::vector<ncclComm_t> nc(ndev, nullptr);
    if (want == COMM_RCCL) {
      bool ok = rccl_load() == DHQR_OK;
      if (ok) {
        const ncclResult_t r = g_rccl.CommInitAll(nc.data(), ndev, g->dev.data());
        if (r != ncclSuccess) {
          ok = false;
          fprintf(stderr, "libdhqr: ncclCommInitAll failed (%s); using peer copies\n", g_rccl.GetErrorString(r));
        }
      }
      if (!ok) {
        if (const char *e = getenv("DHQR_TRANSPORT"))
          if (!strcmp(e, "rccl")) return set_err(DHQR_ECOMM, "RCCL requested (DHQR_TRANSPORT=rccl) but not available: %s", g_err);
        want = COMM_LOCAL;
      }
    }
    LocalWorld *w = nullptr;
    if (want == COMM_LOCAL) {
      CHECK(local_world_create(&w, g->dev.data(), ndev));
      w->refs.store(ndev);
      for (int a = 0; a < ndev; ++a) {  // peer access for the device-to-device copies (ignored when already on / same device)
        (void)hipSetDevice(g->dev[a]);
        for (int b = 0; b < ndev; ++b)
          if (g->dev[a] != g->dev[b]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->dev[a], g->dev[b]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(g->dev[b], 0);
          }
        (void)hipGetLastError();
      }
    }
    for (int r = 0; r < ndev; ++r) {
      CHECK(comm_new(&g->rk[r].cm, g->rk[r].c, want, ndev, r));
      g->rk[r].cm->nccl = nc[r];
      g->rk[r].cm->world = w;
    }
    if (ndev > 1 && lane_channel_wanted(want)) {  // second channel (look-ahead lane of the row-split driver)
      std::vector<ncclComm_t> nc2(ndev, nullptr);
      LocalWorld *w2 = nullptr;
      bool ok = true;
      if (want == COMM_RCCL) {
        const ncclResult_t r = g_rccl.CommInitAll(nc2.data(), ndev, g->dev.data());
        if (r != ncclSuccess) {
          ok = false;
          fprintf(stderr, "libdhqr: second ncclCommInitAll failed (%s); the lane shares the first channel\n", g_rccl.GetErrorString(r));
        }
      } else {
        CHECK(local_world_create(&w2, g->dev.data(), ndev));
        w2->refs.store(ndev);
      }
      if (ok)
        for (int r = 0; r < ndev; ++r) {
          CHECK(comm_new(&g->rk[r].cm->lane, g->rk[r].c, want, ndev, r));
          g->rk[r].cm->lane->nccl = nc2[r];
          g->rk[r].cm->lane->world = w2;
        }
    }
    g->transport = want;
    for (int r = 0; r < ndev; ++r) g->rk[r].th = std::thread(mg_worker, g, r);
    if (want == COMM_RCCL)  // ring broadcast vs scatter + all-gather, measured on this node (collective: rank threads)
      CHECK(mg_run(g, [g](int r) -> int32_t { return comm_tune_bcast(g->rk[r].cm, g->rk[r].c->stream); }));
    return DHQR_OK;
  };
  const int32_t rc = init();
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc != DHQR_OK) {
    (void)dhqr_mg_destroy(g);
    return rc;
  }
  *out = g;
  return DHQR_OK;
}

int32_t dhqr_mg_destroy(dhqr_mg *g) {
  if (!g) return DHQR_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  bool threads = false;
  for (auto &k : g->rk) threads |= k.th.joinable();
  if (threads) {
    (void)mg_free_matrix(g);
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->quit = true;
    }
    g->cv_job.notify_all();
    for (auto &k : g->rk)
      if (k.th.joinable()) k.th.join();
  }
  for (auto &k : g->rk) {
    if (k.cm) (void)comm_free(k.cm);
    if (k.c) (void)dhqr_destroy(k.c);
  }
  delete g;
  if (prev >= 0) (void)hipSetDevice(prev);
  return DHQR_OK;
}

int32_t dhqr_mg_get_bcast_tuning(dhqr_mg *g, int32_t *algo, double *ms_ring, double *ms_sag) {
  if (!g || g->rk.empty()) return set_err(DHQR_EINVAL, "null handle");
  return dhqr_comm_get_bcast_tuning(g->rk[0].cm, algo, ms_ring, ms_sag);
}

int32_t dhqr_mg_comm_counters(dhqr_mg *g, int32_t rank, int64_t *out4) {
  if (!g || rank < 0 || rank >= g->ndev) return set_err(DHQR_EINVAL, "bad arguments");
  return dhqr_comm_counters(g->rk[rank].cm, out4);
}
int32_t dhqr_mg_comm_timing(dhqr_mg *g, int32_t rank, int32_t on, double *out4) {
  if (!g || rank < 0 || rank >= g->ndev) return set_err(DHQR_EINVAL, "bad arguments");
  return dhqr_comm_timing(g->rk[rank].cm, on, out4);
}
int32_t dhqr_mg_rccl_nranks(dhqr_mg *g, int32_t *main_channel, int32_t *lane_channel) {
  if (!g || g->rk.empty()) return set_err(DHQR_EINVAL, "null handle");
  return dhqr_comm_rccl_nranks(g->rk[0].cm, main_channel, lane_channel);
}
int32_t dhqr_mg_info(dhqr_mg *g, int32_t *ndev, int32_t *transport, int64_t *m, int64_t *n) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (ndev) *ndev = g->ndev;
  if (transport) *transport = g->transport;
  if (m) *m = g->m;
  if (n) *n = g->n;
  return DHQR_OK;
}

// Device-resident m x n matrix, block-cyclic columns over the devices (+ replicated alpha).
int32_t dhqr_mg_alloc_f64(dhqr_mg *g, int64_t m, int64_t n) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  CHECK(mg_free_matrix(g));
  g->m = m;
  g->n = n;
  g->rowsplit = false;
  return mg_run(g, [g, m, n](int r) -> int32_t {
    MgRank &k = g->rk[r];
    k.ncl = cs_local_cols(n, g->ndev, r);
    k.lda = (m + 1) & ~(int64_t)1;
    const size_t elems = (size_t)k.lda * std::max<int64_t>(k.ncl, 1);
    if (hipMalloc((void **)&k.A, elems * sizeof(double)) != hipSuccess)
      return set_err(DHQR_ENOMEM, "hipMalloc of the %lld x %lld local block failed", (long long)m, (long long)k.ncl);
    k.capA = elems;
    if (hipMalloc((void **)&k.alpha, (size_t)n * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc of alpha failed");
    HIPCHECK(hipMemsetAsync(k.alpha, 0, (size_t)n * sizeof(double), k.c->stream));
    return cs_prepare(mg_problem(g, r));
  });
}

int32_t dhqr_mg_fill_uniform_f64(dhqr_mg *g, uint64_t seed) {
  if (!g || !g->m || g->rowsplit) return set_err(DHQR_EINVAL, "no column-split matrix allocated");
  return mg_run(g, [g, seed](int r) -> int32_t {
    MgRank &k = g->rk[r];
    if (k.ncl == 0) return DHQR_OK;
    return dhqr_fill_uniform_f64(k.c, k.A, g->m, k.ncl, k.lda, seed, g->m, 0, CS_CB, g->ndev, r);
  });
}

// householder!(A, alpha) over all devices; returns when every device has finished.
int32_t dhqr_mg_factor_f64(dhqr_mg *g) {
  if (!g || !g->m || g->rowsplit) return set_err(DHQR_EINVAL, "no column-split matrix allocated");
  return mg_run(g, [g](int r) -> int32_t {
    const CsProblem pr = mg_problem(g, r);
    CHECK(cs_factor(pr));
    HIPCHECK(hipStreamSynchronize(pr.c->stream));
    return DHQR_OK;
  });
}

int32_t dhqr_mg_residual_f64(dhqr_mg *g, uint64_t seed, double *hrel) {
  if (!g || !g->m || g->rowsplit) return set_err(DHQR_EINVAL, "no column-split matrix allocated");
  if (!hrel) return set_err(DHQR_EINVAL, "null output");
  CHECK(mg_run(g, [g, seed](int r) -> int32_t {
    MgRank &k = g->rk[r];
    const size_t elems = (size_t)g->m * std::max<int64_t>(k.ncl, 1);
    if (!k.W && hipMalloc((void **)&k.W, elems * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
    if (!k.A0 && hipMalloc((void **)&k.A0, elems * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
    return cs_residual(mg_problem(g, r), seed, k.W, k.A0, &k.resid);
  }));
  *hrel = g->rk[0].resid;
  return DHQR_OK;
}

// Host matrix <-> the block-cyclic device blocks.
static int32_t mg_transfer(dhqr_mg *g, double *hA, int64_t lda, double *halpha, bool upload) {
  if (g->rowsplit) return set_err(DHQR_EINVAL, "the handle holds a row-split matrix (use dhqr_mg_rs_transfer_f64)");
  return mg_run(g, [g, hA, lda, halpha, upload](int r) -> int32_t {
    MgRank &k = g->rk[r];
    const CsProblem pr = mg_problem(g, r);
    const int64_t NB = DHQR_NBV, K = cs_nblocks(g->n);
    for (int64_t b = 0; b < K; ++b) {
      if (!pr.mine(b)) continue;
      const int64_t lc = pr.lcol(b), w = std::min<int64_t>(NB, g->n - b * NB);
      if (upload)
        HIPCHECK(hipMemcpy2DAsync(k.A + lc * k.lda, k.lda * sizeof(double), hA + b * NB * lda, lda * sizeof(double),
                                  g->m * sizeof(double), w, hipMemcpyHostToDevice, k.c->stream));
      else
        HIPCHECK(hipMemcpy2DAsync(hA + b * NB * lda, lda * sizeof(double), k.A + lc * k.lda, k.lda * sizeof(double),
                                  g->m * sizeof(double), w, hipMemcpyDeviceToHost, k.c->stream));
    }
    if (halpha) {
      if (upload)
        HIPCHECK(hipMemcpyAsync(k.alpha, halpha, (size_t)g->n * sizeof(double), hipMemcpyHostToDevice, k.c->stream));
      else if (r == 0)
        HIPCHECK(hipMemcpyAsync(halpha, k.alpha, (size_t)g->n * sizeof(double), hipMemcpyDeviceToHost, k.c->stream));
    }
    HIPCHECK(hipStreamSynchronize(k.c->stream));
    return DHQR_OK;
  });
}
int32_t dhqr_mg_upload_f64(dhqr_mg *g, const double *hA, int64_t lda, const double *halpha) {
  if (!g || !g->m) return set_err(DHQR_EINVAL, "no matrix allocated");
  if (!hA || lda < g->m) return set_err(DHQR_EINVAL, "bad host matrix");
  return mg_transfer(g, const_cast<double *>(hA), lda, const_cast<double *>(halpha), true);
}
int32_t dhqr_mg_download_f64(dhqr_mg *g, double *hA, int64_t lda, double *halpha) {
  if (!g || !g->m) return set_err(DHQR_EINVAL, "no matrix allocated");
  if (!hA || lda < g->m) return set_err(DHQR_EINVAL, "bad host matrix");
  return mg_transfer(g, hA, lda, halpha, false);
}

// qr!(A; ndev) (src:311-315): host in / host out over all devices of the handle.
int32_t dhqr_mg_qr_f64(dhqr_mg *g, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  if (g->m != m || g->n != n) CHECK(dhqr_mg_alloc_f64(g, m, n));
  CHECK(mg_transfer(g, hA, lda, nullptr, true));
  CHECK(dhqr_mg_factor_f64(g));
  return mg_transfer(g, hA, lda, halpha, false);
}

// solve_householder!(b, H, alpha) with the factored matrix resident in the handle: hx[0:n] <- x; hb (m) is not modified.
int32_t dhqr_mg_solve_f64(dhqr_mg *g, const double *hb, double *hx) {
  if (!g || !g->m || g->rowsplit) return set_err(DHQR_EINVAL, "no column-split matrix allocated");
  if (!hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  return mg_run(g, [g, hb, hx](int r) -> int32_t {
    MgRank &k = g->rk[r];
    const int64_t m = g->m;
    if (!k.vec && hipMalloc((void **)&k.vec, (size_t)(2 * m + 2 * DHQR_NBV + 16) * sizeof(double)) != hipSuccess)
      return set_err(DHQR_ENOMEM, "hipMalloc failed");
    double *db = k.vec, *du = k.vec + ((m + 15) & ~(int64_t)15);
    HIPCHECK(hipMemcpyAsync(db, hb, (size_t)m * sizeof(double), hipMemcpyHostToDevice, k.c->stream));  // src:318 copy of b
    CHECK(cs_solve(mg_problem(g, r), db, du));
    if (r == 0) HIPCHECK(hipMemcpyAsync(hx, db, (size_t)g->n * sizeof(double), hipMemcpyDeviceToHost, k.c->stream));
    HIPCHECK(hipStreamSynchronize(k.c->stream));
    return pipe_error_check(k.c);
  });
}

// `H \ b` (src:317-321) for a factored HOST matrix: uploads (hA, halpha), solves, returns x.
int32_t dhqr_mg_ldiv_f64(dhqr_mg *g, const double *hA, int64_t m, int64_t n, int64_t lda, const double *halpha,
                         const double *hb, double *hx) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  if (g->m != m || g->n != n) CHECK(dhqr_mg_alloc_f64(g, m, n));
  CHECK(mg_transfer(g, const_cast<double *>(hA), lda, const_cast<double *>(halpha), true));
  return dhqr_mg_solve_f64(g, hb, hx);
}

// ---- ComplexF64 column split (dhqr_zdist.h): cyclic blocks of DHQR_ZNB = 64 complex columns
#include "dhqr_zdist.h"
int64_t dhqr_cs_local_cols_c64(int64_t n, int32_t nranks, int32_t rank) {
  if (n < 0 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  return zcs_local_cols(n, nranks, rank);
}

// householder!(A, alpha) (src:215-294) for ComplexF64 columns distributed over the communicator's ranks.
int32_t dhqr_cs_factor_c64(dhqr_comm *cm, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (no_columns(m, n)) return DHQR_OK;
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  ENTER(cm->ctx);
  const int64_t ncl = zcs_local_cols(n, cm->nranks, cm->rank);
  if (ncl > 0) {
    CHECK(check_mat(dA, m, ncl, 2 * lda, false));  // the MFMA kernels see the interleaved storage as a real matrix with ld = 2 lda
    CHECK(check_mat(dA, m, ncl, lda, false));
    CHECK(check_zptr(dA, "matrix"));
  }
  CHECK(check_zptr(dalpha, "alpha"));
  return zcs_factor(cm->ctx, cm, dA, m, n, lda, dalpha);
}

// qr!(A::DArray{ComplexF64}) (src:115-120, 311-315) for one process: its contiguous host column block in, the factored
// block + the replicated alpha out (the ComplexF64 method of dhqr_cs_qr_darray_f64).
int32_t dhqr_cs_qr_darray_c64(dhqr_comm *cm, double *hBlock, int64_t m, int64_t n, int64_t ldb, double *halpha) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (no_columns(m, n)) return DHQR_OK;
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  dhqr_ctx *c = cm->ctx;
  ENTER(c);
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  const int P = cm->nranks, r = cm->rank;
  int64_t lo, hi;
  cs_contig_range(n, P, r, &lo, &hi);
  const int64_t wr = hi - lo, wmax = std::max<int64_t>(n / P + 1, DHQR_ZNB), ncl = zcs_local_cols(n, P, r);
  if (wr > 0 && (!hBlock || ldb < m)) return set_err(DHQR_EINVAL, "bad local block");
  const size_t esz = 2 * sizeof(double);
  double *dA = nullptr, *dBlk = nullptr, *dStage = nullptr, *dal = nullptr;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&dA, (size_t)m * std::max<int64_t>(ncl, 1) * esz));
    HIPCHECK(hipMalloc((void **)&dBlk, (size_t)m * std::max<int64_t>(wr, 1) * esz));
    HIPCHECK(hipMalloc((void **)&dStage, (size_t)m * wmax * esz));
    HIPCHECK(hipMalloc((void **)&dal, (size_t)n * esz));
    HIPCHECK(hipMemsetAsync(dal, 0, (size_t)n * esz, c->stream));
    if (wr > 0) HIPCHECK(hipMemcpy2DAsync(dBlk, m * esz, hBlock, ldb * esz, m * esz, wr, hipMemcpyHostToDevice, c->stream));
    CHECK(zcs_convert(c, cm, dA, m, n, m, dBlk, m, dStage, true));
    CHECK(zcs_factor(c, cm, dA, m, n, m, dal));
    CHECK(zcs_convert(c, cm, dA, m, n, m, dBlk, m, dStage, false));
    if (wr > 0) HIPCHECK(hipMemcpy2DAsync(hBlock, ldb * esz, dBlk, m * esz, m * esz, wr, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(halpha, dal, (size_t)n * esz, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return pipe_error_check(c);
  };
  const int32_t rc = body();
  (void)hipDeviceSynchronize();
  double *ps[] = {dA, dBlk, dStage, dal};
  for (double *p : ps)
    if (p) (void)hipFree(p);
  return rc;
}

// solve_householder!(b, H, alpha) (src:226-282) for ComplexF64 on the cyclic 64-column split: db (m complex, the same on
// every rank) is overwritten, x = db[0:n] on every rank; dwork: dhqr_cs_solve_work_c64(m, nranks) complex of scratch (b is
// carried in double-double).  Asynchronous on the context's stream.
int64_t dhqr_cs_solve_work_c64(int64_t m, int32_t nranks) { return (m < 0 || nranks < 1) ? -1 : zcs_solve_work(m, nranks); }
int32_t dhqr_cs_solve_c64(dhqr_comm *cm, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha, double *db,
                          double *dwork) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (no_columns(m, n)) return DHQR_OK;
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  ENTER(cm->ctx);
  const int64_t ncl = zcs_local_cols(n, cm->nranks, cm->rank);
  if (ncl > 0) {
    CHECK(check_mat(dA, m, ncl, lda, false));
    CHECK(check_zptr(dA, "matrix"));
  }
  CHECK(check_zptr(dalpha, "alpha"));
  CHECK(check_zptr(db, "b"));
  CHECK(check_zptr(dwork, "work"));
  return zcs_solve(cm->ctx, cm, dA, m, n, lda, dalpha, db, dwork);
}

// `H \ b` (src:317-321) for one process holding its contiguous block of a factored DArray{ComplexF64} (the ComplexF64
// method of dhqr_cs_ldiv_darray_f64): host block + replicated alpha + b in, x (n complex) out on every process.
int32_t dhqr_cs_ldiv_darray_c64(dhqr_comm *cm, const double *hBlock, int64_t m, int64_t n, int64_t ldb, const double *halpha,
                                const double *hb, double *hx) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (no_columns(m, n)) return DHQR_OK;
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  dhqr_ctx *c = cm->ctx;
  ENTER(c);
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  const int P = cm->nranks, r = cm->rank;
  int64_t lo, hi;
  cs_contig_range(n, P, r, &lo, &hi);
  const int64_t wr = hi - lo, wmax = std::max<int64_t>(n / P + 1, DHQR_ZNB), ncl = zcs_local_cols(n, P, r);
  if (wr > 0 && (!hBlock || ldb < m)) return set_err(DHQR_EINVAL, "bad local block");
  const size_t esz = 2 * sizeof(double);
  double *dA = nullptr, *dBlk = nullptr, *dStage = nullptr, *dal = nullptr, *dvec = nullptr;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&dA, (size_t)m * std::max<int64_t>(ncl, 1) * esz));
    HIPCHECK(hipMalloc((void **)&dBlk, (size_t)m * std::max<int64_t>(wr, 1) * esz));
    HIPCHECK(hipMalloc((void **)&dStage, (size_t)m * wmax * esz));
    HIPCHECK(hipMalloc((void **)&dal, (size_t)n * esz));
    HIPCHECK(hipMalloc((void **)&dvec, (size_t)(m + zcs_solve_work(m, P)) * esz));
    if (wr > 0) HIPCHECK(hipMemcpy2DAsync(dBlk, m * esz, hBlock, ldb * esz, m * esz, wr, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(dal, halpha, (size_t)n * esz, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(dvec, hb, (size_t)m * esz, hipMemcpyHostToDevice, c->stream));  // src:318 copy of b
    CHECK(zcs_convert(c, cm, dA, m, n, m, dBlk, m, dStage, true));
    CHECK(zcs_solve(c, cm, dA, m, n, m, dal, dvec, dvec + 2 * m));
    HIPCHECK(hipMemcpyAsync(hx, dvec, (size_t)n * esz, hipMemcpyDeviceToHost, c->stream));  // src:320
    HIPCHECK(hipStreamSynchronize(c->stream));
    return DHQR_OK;
  };
  const int32_t rc = body();
  (void)hipDeviceSynchronize();
  double *ps[] = {dA, dBlk, dStage, dal, dvec};
  for (double *p : ps)
    if (p) (void)hipFree(p);
  return rc;
}

// `H \ b` (src:317-321) for a factored ComplexF64 HOST matrix over all devices of the handle: every rank uploads its cyclic
// 64-column blocks of (hA, halpha) and the ranks solve together (zcs_solve); x (n complex) comes back from rank 0.
int32_t dhqr_mg_ldiv_c64(dhqr_mg *g, const double *hA, int64_t m, int64_t n, int64_t lda, const double *halpha, const double *hb,
                         double *hx) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha || !hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  return mg_run(g, [g, hA, m, n, lda, halpha, hb, hx](int r) -> int32_t {
    MgRank &k = g->rk[r];
    dhqr_ctx *c = k.c;
    const int P = g->ndev;
    const int64_t ZB = DHQR_ZNB, K = zcs_npanels(n), ncl = zcs_local_cols(n, P, r);
    const size_t esz = 2 * sizeof(double);
    double *dA = nullptr, *dal = nullptr, *dvec = nullptr;
    auto body = [&]() -> int32_t {
      if (hipMalloc((void **)&dA, (size_t)m * (size_t)std::max<int64_t>(ncl, 1) * esz) != hipSuccess)
        return set_err(DHQR_ENOMEM, "hipMalloc of the %lld x %lld complex block failed", (long long)m, (long long)ncl);
      if (hipMalloc((void **)&dal, (size_t)n * esz) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc of alpha failed");
      if (hipMalloc((void **)&dvec, (size_t)(m + zcs_solve_work(m, P)) * esz) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc of b failed");
      for (int64_t b = r; b < K; b += P) {
        const int64_t w = std::min<int64_t>(ZB, n - b * ZB);
        HIPCHECK(hipMemcpy2DAsync(dA + 2 * (b / P) * ZB * m, m * esz, hA + 2 * b * ZB * lda, lda * esz, m * esz, w,
                                  hipMemcpyHostToDevice, c->stream));
      }
      HIPCHECK(hipMemcpyAsync(dal, halpha, (size_t)n * esz, hipMemcpyHostToDevice, c->stream));
      HIPCHECK(hipMemcpyAsync(dvec, hb, (size_t)m * esz, hipMemcpyHostToDevice, c->stream));  // src:318 copy of b
      CHECK(zcs_solve(c, P > 1 ? k.cm : nullptr, dA, m, n, m, dal, dvec, dvec + 2 * m));
      if (r == 0) HIPCHECK(hipMemcpyAsync(hx, dvec, (size_t)n * esz, hipMemcpyDeviceToHost, c->stream));  // src:320
      HIPCHECK(hipStreamSynchronize(c->stream));
      return DHQR_OK;
    };
    const int32_t rc = body();
    if (rc != DHQR_OK) (void)hipDeviceSynchronize();
    double *ps[] = {dA, dal, dvec};
    for (double *p : ps)
      if (p) (void)hipFree(p);
    return rc;
  });
}

// qr!(A; ndev) (src:311-315) for a ComplexF64 host matrix over all devices of the handle: host in / host out (H in
// place of A, alpha), the format dhqr_ldiv_c64 solves with.  The device blocks live for the duration of the call.
int32_t dhqr_mg_qr_c64(dhqr_mg *g, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (no_columns(m, n)) return DHQR_OK;
  CHECK(check_mat(hA, m, n, lda, true));
  if (!halpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  return mg_run(g, [g, hA, m, n, lda, halpha](int r) -> int32_t {
    MgRank &k = g->rk[r];
    dhqr_ctx *c = k.c;
    const int P = g->ndev;
    const int64_t ZB = DHQR_ZNB, K = zcs_npanels(n), ncl = zcs_local_cols(n, P, r);
    const size_t esz = 2 * sizeof(double);
    double *dA = nullptr, *dal = nullptr;
    auto body = [&]() -> int32_t {
      if (hipMalloc((void **)&dA, (size_t)m * (size_t)std::max<int64_t>(ncl, 1) * esz) != hipSuccess)
        return set_err(DHQR_ENOMEM, "hipMalloc of the %lld x %lld complex block failed", (long long)m, (long long)ncl);
      if (hipMalloc((void **)&dal, (size_t)n * esz) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc of alpha failed");
      HIPCHECK(hipMemsetAsync(dal, 0, (size_t)n * esz, c->stream));
      for (int64_t b = r; b < K; b += P) {
        const int64_t w = std::min<int64_t>(ZB, n - b * ZB);
        HIPCHECK(hipMemcpy2DAsync(dA + 2 * (b / P) * ZB * m, m * esz, hA + 2 * b * ZB * lda, lda * esz, m * esz, w,
                                  hipMemcpyHostToDevice, c->stream));
      }
      CHECK(zcs_factor(c, P > 1 ? k.cm : nullptr, dA, m, n, m, dal));
      for (int64_t b = r; b < K; b += P) {
        const int64_t w = std::min<int64_t>(ZB, n - b * ZB);
        HIPCHECK(hipMemcpy2DAsync(hA + 2 * b * ZB * lda, lda * esz, dA + 2 * (b / P) * ZB * m, m * esz, m * esz, w,
                                  hipMemcpyDeviceToHost, c->stream));
      }
      if (r == 0) HIPCHECK(hipMemcpyAsync(halpha, dal, (size_t)n * esz, hipMemcpyDeviceToHost, c->stream));
      HIPCHECK(hipStreamSynchronize(c->stream));
      return pipe_error_check(c);
    };
    const int32_t rc = body();
    if (rc != DHQR_OK) (void)hipDeviceSynchronize();
    if (dA) (void)hipFree(dA);
    if (dal) (void)hipFree(dal);
    return rc;
  });
}

int32_t dhqr_mg_set_profiling(dhqr_mg *g, int32_t on) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  return mg_run(g, [g, on](int r) -> int32_t { return dhqr_set_profiling(g->rk[r].c, on); });
}
int32_t dhqr_mg_reset_stats(dhqr_mg *g) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  return mg_run(g, [g](int r) -> int32_t { return dhqr_reset_stats(g->rk[r].c); });
}
int32_t dhqr_mg_get_stats(dhqr_mg *g, int32_t rank, dhqr_stats *out, int64_t *n_fast, int64_t *n_fallback, int64_t *bytes_bcast) {
  if (!g || rank < 0 || rank >= g->ndev || !out) return set_err(DHQR_EINVAL, "bad arguments");
  CHECK(mg_run(g, [g, rank, out](int r) -> int32_t { return r == rank ? dhqr_get_stats(g->rk[r].c, out) : DHQR_OK; }));
  if (n_fast) *n_fast = g->rk[rank].c->n_fast;
  if (n_fallback) *n_fallback = g->rk[rank].c->n_fallback;
  if (bytes_bcast) *bytes_bcast = g->rk[rank].cm->bytes_bcast;
  return DHQR_OK;
}


// ============================================================ multi-GPU: row split (BASELINE configs[4]), SPMD
void dhqr_rs_row_range(int64_t m, int32_t nranks, int32_t rank, int64_t *row0, int64_t *mloc) {
  rs_row_range(m, nranks, rank, row0, mloc);
}
static int32_t rs_check(dhqr_comm *cm, const void *dA, int64_t m, int64_t n, int64_t lda, RsProblem *pr) {
  if (!cm) return set_err(DHQR_EINVAL, "null communicator");
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  pr->c = cm->ctx;
  pr->cm = cm;
  pr->P = cm->nranks;
  pr->r = cm->rank;
  pr->m = m;
  pr->n = n;
  pr->lda = lda;
  rs_row_range(m, pr->P, pr->r, &pr->row0, &pr->mloc);
  pr->A = const_cast<double *>((const double *)dA);
  pr->alpha = nullptr;
  if (pr->mloc > 0) CHECK(check_mat(dA, pr->mloc, n, lda, false));
  return DHQR_OK;
}
int32_t dhqr_rs_fill_uniform_f64(dhqr_comm *cm, double *dA, int64_t m, int64_t n, int64_t lda, uint64_t seed) {
  RsProblem pr;
  CHECK(rs_check(cm, dA, m, n, lda, &pr));
  if (pr.mloc == 0) return DHQR_OK;
  return dhqr_fill_uniform_f64(pr.c, dA, pr.mloc, n, lda, seed, m, pr.row0, DHQR_NBV, 1, 0);
}
int32_t dhqr_rs_factor_f64(dhqr_comm *cm, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha) {
  RsProblem pr;
  CHECK(rs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dalpha) return set_err(DHQR_EINVAL, "null alpha pointer");
  pr.alpha = dalpha;
  return rs_factor(pr);
}
int32_t dhqr_rs_residual_f64(dhqr_comm *cm, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha,
                             uint64_t seed, double *dB, double *dA0, double *hrel) {
  RsProblem pr;
  CHECK(rs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dalpha || !hrel || (pr.mloc > 0 && (!dB || !dA0))) return set_err(DHQR_EINVAL, "null pointer argument");
  pr.alpha = const_cast<double *>(dalpha);
  return rs_residual(pr, seed, dB, dA0, hrel);
}
int32_t dhqr_rs_solve_f64(dhqr_comm *cm, const double *dA, int64_t m, int64_t n, int64_t lda, const double *dalpha,
                          double *db, double *dx) {
  RsProblem pr;
  CHECK(rs_check(cm, dA, m, n, lda, &pr));
  ENTER(pr.c);
  if (!dalpha || !dx || (pr.mloc > 0 && !db)) return set_err(DHQR_EINVAL, "null pointer argument");
  pr.alpha = const_cast<double *>(dalpha);
  return rs_solve(pr, db, dx);
}

// single-process handle, row split: device-resident m x n matrix, 128-row aligned slabs over the devices
static RsProblem mg_rs_problem(dhqr_mg *g, int r) {
  RsProblem pr;
  MgRank &k = g->rk[r];
  pr.c = k.c;
  pr.cm = k.cm;
  pr.A = k.A;
  pr.m = g->m;
  pr.n = g->n;
  pr.lda = k.lda;
  pr.alpha = k.alpha;
  pr.P = g->ndev;
  pr.r = r;
  rs_row_range(g->m, g->ndev, r, &pr.row0, &pr.mloc);
  return pr;
}
int32_t dhqr_mg_rs_alloc_f64(dhqr_mg *g, int64_t m, int64_t n) {
  if (!g) return set_err(DHQR_EINVAL, "null handle");
  if (m <= 0 || n <= 0 || m < n) return set_err(DHQR_EINVAL, "m >= n >= 1 required (m=%lld n=%lld)", (long long)m, (long long)n);
  CHECK(mg_free_matrix(g));
  g->m = m;
  g->n = n;
  g->rowsplit = true;
  return mg_run(g, [g, m, n](int r) -> int32_t {
    MgRank &k = g->rk[r];
    int64_t row0, mloc;
    rs_row_range(m, g->ndev, r, &row0, &mloc);
    k.ncl = n;
    k.lda = std::max<int64_t>((mloc + 1) & ~(int64_t)1, 2);
    const size_t elems = (size_t)k.lda * n;
    if (hipMalloc((void **)&k.A, elems * sizeof(double)) != hipSuccess)
      return set_err(DHQR_ENOMEM, "hipMalloc of the %lld x %lld local slab failed", (long long)mloc, (long long)n);
    k.capA = elems;
    if (hipMalloc((void **)&k.alpha, (size_t)n * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc of alpha failed");
    HIPCHECK(hipMemsetAsync(k.alpha, 0, (size_t)n * sizeof(double), k.c->stream));
    RsWork w;
    return rs_prepare(mg_rs_problem(g, r), &w);
  });
}
int32_t dhqr_mg_rs_fill_uniform_f64(dhqr_mg *g, uint64_t seed) {
  if (!g || !g->m || !g->rowsplit) return set_err(DHQR_EINVAL, "no row-split matrix allocated");
  return mg_run(g, [g, seed](int r) -> int32_t {
    const RsProblem pr = mg_rs_problem(g, r);
    if (pr.mloc == 0) return DHQR_OK;
    return dhqr_fill_uniform_f64(pr.c, pr.A, pr.mloc, pr.n, pr.lda, seed, pr.m, pr.row0, DHQR_NBV, 1, 0);
  });
}
int32_t dhqr_mg_rs_factor_f64(dhqr_mg *g) {
  if (!g || !g->m || !g->rowsplit) return set_err(DHQR_EINVAL, "no row-split matrix allocated");
  return mg_run(g, [g](int r) -> int32_t {
    const RsProblem pr = mg_rs_problem(g, r);
    CHECK(rs_factor(pr));
    HIPCHECK(hipStreamSynchronize(pr.c->stream));
    return DHQR_OK;
  });
}
int32_t dhqr_mg_rs_residual_f64(dhqr_mg *g, uint64_t seed, double *hrel) {
  if (!g || !g->m || !g->rowsplit) return set_err(DHQR_EINVAL, "no row-split matrix allocated");
  if (!hrel) return set_err(DHQR_EINVAL, "null output");
  CHECK(mg_run(g, [g, seed](int r) -> int32_t {
    MgRank &k = g->rk[r];
    const RsProblem pr = mg_rs_problem(g, r);
    const size_t elems = (size_t)std::max<int64_t>(pr.mloc, 1) * pr.n;
    if (!k.W && hipMalloc((void **)&k.W, elems * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
    if (!k.A0 && hipMalloc((void **)&k.A0, elems * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
    return rs_residual(pr, seed, k.W, k.A0, &k.resid);
  }));
  *hrel = g->rk[0].resid;
  return DHQR_OK;
}
// host rows <-> device slabs (hA: m x n column-major, lda); halpha from rank 0 on download
int32_t dhqr_mg_rs_transfer_f64(dhqr_mg *g, double *hA, int64_t lda, double *halpha, int32_t upload) {
  if (!g || !g->m || !g->rowsplit) return set_err(DHQR_EINVAL, "no row-split matrix allocated");
  if (!hA || lda < g->m) return set_err(DHQR_EINVAL, "bad host matrix");
  return mg_run(g, [g, hA, lda, halpha, upload](int r) -> int32_t {
    const RsProblem pr = mg_rs_problem(g, r);
    if (pr.mloc > 0) {
      if (upload)
        HIPCHECK(hipMemcpy2DAsync(pr.A, pr.lda * sizeof(double), hA + pr.row0, lda * sizeof(double), pr.mloc * sizeof(double),
                                  pr.n, hipMemcpyHostToDevice, pr.c->stream));
      else
        HIPCHECK(hipMemcpy2DAsync(hA + pr.row0, lda * sizeof(double), pr.A, pr.lda * sizeof(double), pr.mloc * sizeof(double),
                                  pr.n, hipMemcpyDeviceToHost, pr.c->stream));
    }
    if (halpha && !upload && r == 0)
      HIPCHECK(hipMemcpyAsync(halpha, pr.alpha, (size_t)pr.n * sizeof(double), hipMemcpyDeviceToHost, pr.c->stream));
    if (halpha && upload)
      HIPCHECK(hipMemcpyAsync(pr.alpha, halpha, (size_t)pr.n * sizeof(double), hipMemcpyHostToDevice, pr.c->stream));
    HIPCHECK(hipStreamSynchronize(pr.c->stream));
    return DHQR_OK;
  });
}
// `H \ b` on the resident row-split factorisation: hb (m) is not modified, hx (n) <- x
int32_t dhqr_mg_rs_solve_f64(dhqr_mg *g, const double *hb, double *hx) {
  if (!g || !g->m || !g->rowsplit) return set_err(DHQR_EINVAL, "no row-split matrix allocated");
  if (!hb || !hx) return set_err(DHQR_EINVAL, "null pointer argument");
  return mg_run(g, [g, hb, hx](int r) -> int32_t {
    MgRank &k = g->rk[r];
    const RsProblem pr = mg_rs_problem(g, r);
    const size_t need = (size_t)std::max<int64_t>(pr.mloc, 1) + (size_t)pr.n + 64;
    if (!k.vec && hipMalloc((void **)&k.vec, need * sizeof(double)) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc failed");
    double *db = k.vec, *dx = k.vec + ((std::max<int64_t>(pr.mloc, 1) + 15) & ~(int64_t)15);
    if (pr.mloc > 0)
      HIPCHECK(hipMemcpyAsync(db, hb + pr.row0, (size_t)pr.mloc * sizeof(double), hipMemcpyHostToDevice, pr.c->stream));
    CHECK(rs_solve(pr, db, dx));
    if (r == 0) HIPCHECK(hipMemcpyAsync(hx, dx, (size_t)pr.n * sizeof(double), hipMemcpyDeviceToHost, pr.c->stream));
    HIPCHECK(hipStreamSynchronize(pr.c->stream));
    return pipe_error_check(pr.c);
  });
}

}  // extern "C"
