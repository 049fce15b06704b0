// dhqr_dist.h -- the blocked factorisation driver, written once for P >= 1 ranks (included by dhqr_api.hip).
//
// Replaces householder!(A::DArray, alpha) (src:115-120: owners visited one after the other, every reflector
// shipped to every process, src:141-143) by an SPMD program over a 1-D BLOCK-CYCLIC column split: cyclic block =
// CS_CB = 256 columns = TWO panels; rank r owns the panel pairs q with q % P == r, stored contiguously: the trailing
// columns of every rank are a suffix of its local storage.  Both panels of a pair live on one rank, so the second is
// brought up to date and factored without waiting for a broadcast, and the broadcast of the first runs behind it: per
// pair ONE broadcast sits on the critical chain (narrow update -> panel a -> panel b -> broadcast of b) instead of two
// (tools/scaling_model.py: 8 GPUs 5.7x -> 6.2x at 100 GB/s, 4.8x -> 5.8x at 50 GB/s).  At P == 1 it is the single-GPU
// look-ahead driver.
//
// Panels are grouped: a group is a PAIR of full-width panels (a, b = a+1) applied to the trailing matrix in one
// pass (K = 256 MFMA update, pair_apply) or a single panel (last odd / partial panel, or pairing disabled).
// Per rank three streams:
//   lane  (high priority)  the critical chain: bring the blocks of the NEXT group's panels this rank owns up to
//                          date, factor them (asynchronously verified R-first path), assemble the pair;
//   comm                   one broadcast per panel of its operands [T | T' | alpha | status | V] (root = owner);
//   wide  (caller's)       apply group g to the local blocks beyond group g+1 -- first the blocks of group g+2
//                          ("head": the lane needs them next), then the rest.
// Nothing in the loop waits on the host: panels are committed on the device (k_build_t) and the driver
// reads one status word after the last launch; a rejected panel (ill-conditioned for CholeskyQR) makes every
// later matrix update a no-op and the run resumes from that panel with the host-verified robust path.
//
// Group buffer (ring of CS_NGB, sized for the first group):
//   [ tail_a | V_a : ldv x 128 | V_b : ldv x 128 | tail_b | S_ba ]      ldv = panel_ldv(rows of panel a)
// V_b is stored shifted down by 128 rows (zeros above) so [V_a | V_b] is the K = 256 operand; panel a is
// broadcast as the contiguous range [tail_a | V_a], panel b as [V_b | tail_b]: no repacking on any rank.
#pragma once

#define CS_NGB 6   // group buffers in flight (a quad step keeps two of them until its wide update has finished)
#define CS_EVR 8   // event ring length (groups); panels use 2 * CS_EVR

struct CsState {
  bool init = false;
  hipStream_t comm = nullptr;
  hipEvent_t ev_group[CS_EVR], ev_wide[CS_EVR], ev_head[CS_EVR], ev_lane[CS_EVR];
  hipEvent_t ev_ready[2 * CS_EVR], ev_recv[2 * CS_EVR];
  hipEvent_t ev_v[2 * CS_EVR], ev_y[2 * CS_EVR], ev_x[CS_EVR];  // lane side stream: V of a panel final / Y = V_a' C_b done / a group's cross terms done
  hipEvent_t ev_t[2 * CS_EVR];  // a panel's T and verdict are final (k_build_t done): what its off-lane commit waits for
  hipEvent_t ev_start = nullptr, ev_end = nullptr;
  int64_t ticket[2 * CS_EVR];
  Buf gbuf[CS_NGB];
  Buf vt;  // legacy packed panel buffer for re-applied panels (resume, residual, solve)
};

static int32_t cs_state_init(dhqr_ctx *c) {
  if (!c->cs) c->cs = new CsState();
  CsState &s = *c->cs;
  if (s.init) return DHQR_OK;
  int lo = 0, hi = 0;
  HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIPCHECK(hipStreamCreateWithPriority(&s.comm, hipStreamNonBlocking, hi));
  for (int i = 0; i < CS_EVR; ++i) {
    HIPCHECK(hipEventCreateWithFlags(&s.ev_group[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_wide[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_head[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_lane[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_x[i], hipEventDisableTiming));
  }
  for (int i = 0; i < 2 * CS_EVR; ++i) {
    HIPCHECK(hipEventCreateWithFlags(&s.ev_ready[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_recv[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_v[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_y[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_t[i], hipEventDisableTiming));
    s.ticket[i] = -1;
  }
  HIPCHECK(hipEventCreateWithFlags(&s.ev_start, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&s.ev_end, hipEventDisableTiming));
  s.init = true;
  return DHQR_OK;
}
static void cs_state_free(dhqr_ctx *c) {
  if (!c->cs) return;
  CsState &s = *c->cs;
  if (s.init) {
    for (int i = 0; i < CS_EVR; ++i) {
      (void)hipEventDestroy(s.ev_group[i]);
      (void)hipEventDestroy(s.ev_wide[i]);
      (void)hipEventDestroy(s.ev_head[i]);
      (void)hipEventDestroy(s.ev_lane[i]);
      (void)hipEventDestroy(s.ev_x[i]);
    }
    for (int i = 0; i < 2 * CS_EVR; ++i) {
      (void)hipEventDestroy(s.ev_ready[i]);
      (void)hipEventDestroy(s.ev_recv[i]);
      (void)hipEventDestroy(s.ev_v[i]);
      (void)hipEventDestroy(s.ev_y[i]);
      (void)hipEventDestroy(s.ev_t[i]);
    }
    (void)hipEventDestroy(s.ev_start);
    (void)hipEventDestroy(s.ev_end);
    (void)hipStreamDestroy(s.comm);
  }
  for (Buf &b : s.gbuf)
    if (b.p) (void)hipFree(b.p);
  if (s.vt.p) (void)hipFree(s.vt.p);
  delete c->cs;
  c->cs = nullptr;
}

// ---- block-cyclic column map (cyclic block = CS_CB = DHQR_CS_BLOCK columns = two panels) -----------
#define CS_CB ((int64_t)DHQR_CS_BLOCK)
static_assert(DHQR_CS_BLOCK == 2 * DHQR_NBV, "the cyclic block is a pair of panels");
static inline int64_t cs_nblocks(int64_t n) { return (n + DHQR_NBV - 1) / DHQR_NBV; }  // panels
static inline int64_t cs_local_cols(int64_t n, int P, int r) {
  const int64_t K = cs_nblocks(n);
  int64_t cols = 0;
  for (int64_t k = 0; k < K; ++k)
    if ((k / 2) % P == r) cols += std::min<int64_t>(DHQR_NBV, n - k * DHQR_NBV);
  return cols;
}

struct CsProblem {
  dhqr_ctx *c;
  dhqr_comm *cm;   // nullptr: single rank
  double *A;       // local block: m x cs_local_cols(n, P, r), leading dimension lda
  int64_t m, n, lda;
  double *alpha;   // n doubles, replicated on every rank
  int P, r;
  int64_t K, ncl;
  int quads_ok = -1;  // P > 1: do ALL ranks meet the quad steps' alignment conditions (cs_agree_quads; -1: not asked yet)
  int64_t width(int64_t k) const { return std::min<int64_t>(DHQR_NBV, n - k * DHQR_NBV); }
  int owner(int64_t k) const { return (int)((k / 2) % P); }      // panel k belongs to the pair k / 2
  bool mine(int64_t k) const { return owner(k) == r; }
  int64_t lcol(int64_t k) const { return ((k / 2) / P) * CS_CB + (k % 2) * DHQR_NBV; }   // local column of an owned panel
  int64_t first_local_ge(int64_t k) const {  // smallest panel index >= k this rank owns (may be >= K)
    const int64_t q = k / 2;
    if (q % P == r) return k;
    return 2 * (q + (((int64_t)r - q % P) % P + P) % P);
  }
  // local columns holding the global blocks >= k: [*lo, ncl)
  int64_t local_from(int64_t k) const {
    const int64_t kk = first_local_ge(k);
    return kk < K ? lcol(kk) : ncl;
  }
};

struct CsGroup {
  int64_t a;      // first panel
  int np;         // 1 or 2 panels
  int64_t last() const { return a + np - 1; }
};

struct CsGroupBuf {
  double *tailA, *VA, *VB, *tailB, *Sba, *S21;  // S21 (256 x 256): V_2' V_1 when this pair is the second of a quad step
  int64_t ldv, rows_a;
  PanelBuf pa() const { return tail_view(VA, ldv, tailA); }
  PanelBuf pb() const { return tail_view(VB + DHQR_NBV, ldv, tailB); }  // V_b proper starts 128 rows down
  double *region(int idx) const { return idx == 0 ? tailA : VB; }
  int64_t region_elems() const { return ldv * DHQR_NBV + panel_tail_elems(); }
};
static inline size_t cs_gbuf_elems(int64_t m) {
  return (size_t)(2 * panel_tail_elems() + 2 * panel_ldv(m) * DHQR_NBV + 5 * (int64_t)DHQR_NBV * DHQR_NBV);
}
// ldv_fixed > 0: every group of the factorisation uses this leading dimension (quad steps: the K = 512 update reads the
// reflectors of two group buffers with one stride)
static inline CsGroupBuf cs_gbuf_view(double *base, int64_t rows_a, int64_t ldv_fixed = 0) {
  CsGroupBuf g;
  g.rows_a = rows_a;
  g.ldv = ldv_fixed > 0 ? ldv_fixed : panel_ldv(rows_a);
  g.tailA = base;
  g.VA = base + panel_tail_elems();
  g.VB = g.VA + g.ldv * DHQR_NBV;
  g.tailB = g.VB + g.ldv * DHQR_NBV;
  g.Sba = g.tailB + panel_tail_elems();
  g.S21 = g.Sba + (int64_t)DHQR_NBV * DHQR_NBV;
  return g;
}

// A step of the wide stream: one group, or a QUAD = two consecutive pairs applied to the trailing matrix in one K = 512
// pass (quad_apply: C is read three times and written once for four panels instead of four times and twice; the K loop
// of the subtraction is twice as long per tile prologue / epilogue).  The group that follows a step receives the step
// from the lane (narrow update), everything beyond from the wide stream.
struct CsStep {
  int g0, ng;  // first group, number of groups (1 or 2)
};
// Groups and steps of a pass that starts at panel kstart.  Quads: 16-byte path (on EVERY rank: the plan is part of the SPMD
// program, cs_agree_quads), and at least c->quad_min_cols columns PER RANK to the right of the quad (beyond that the panel
// chain, not the wide stream, bounds the factorisation and pairs keep the chain shorter).
static void cs_plan(const CsProblem &pr, int64_t kstart, std::vector<CsGroup> &groups, std::vector<CsStep> &steps) {
  const dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV;
  const bool pairing = c->pair && (pr.n >= c->pair_min_n || pr.P > 1);
  groups.clear();
  steps.clear();
  for (int64_t k = kstart; k < pr.K;) {
    CsGroup g;
    g.a = k;
    g.np = (pairing && k + 1 < pr.K && pr.width(k) == NB && pr.width(k + 1) == NB) ? 2 : 1;
    groups.push_back(g);
    k += g.np;
  }
  const bool quads = c->quad && (pr.P == 1 ? (pr.m % 2 == 0 && pr.lda % 2 == 0 && aligned16(pr.A)) : pr.quads_ok == 1);
  const int G = (int)groups.size();
  for (int g = 0; g < G;) {
    CsStep st;
    st.g0 = g;
    st.ng = 1;
    if (quads && g + 1 < G && groups[g].np == 2 && groups[g + 1].np == 2 &&
        (pr.n - (groups[g + 1].last() + 1) * NB) / pr.P >= c->quad_min_cols)
      st.ng = 2;
    steps.push_back(st);
    g += st.ng;
  }
}

// One asynchronous pass over the panels [kstart, K).  `robust_first`: panel kstart is factored with the
// host-verified path (resume after a rejected panel).  Returns in *failed the index of the first rejected panel
// (INT_MAX: none).  fast_idx collects the panels this rank enqueued on the fast path.
static int32_t cs_run(const CsProblem &pr, int64_t kstart, bool robust_first, int *failed, std::vector<int64_t> &fast_idx) {
  dhqr_ctx *c = pr.c;
  dhqr_comm *cm = pr.cm;
  CsState &S = *c->cs;
  const int64_t NB = DHQR_NBV, K = pr.K, m = pr.m, lda = pr.lda;
  const int P = pr.P;
  // ---- groups and steps
  std::vector<CsGroup> groups;
  std::vector<CsStep> steps;
  cs_plan(pr, kstart, groups, steps);
  const int G = (int)groups.size(), NS = (int)steps.size();
  if (G == 0) {
    *failed = INT_MAX;
    return DHQR_OK;
  }
  std::vector<int> step_of((size_t)G);
  bool any_quad = false;
  for (int si = 0; si < NS; ++si)
    for (int q = 0; q < steps[si].ng; ++q) {
      step_of[(size_t)(steps[si].g0 + q)] = si;
      any_quad = any_quad || steps[si].ng == 2;
    }
  const int64_t ldv_fixed = any_quad ? panel_ldv(m - groups[0].a * NB) : 0;
  auto gview = [&](int g) { return cs_gbuf_view(S.gbuf[g % CS_NGB].p, m - groups[g].a * NB, ldv_fixed); };

  // (single rank only: with the communication stream of P > 1 a fifth busy stream shares a hardware queue -- rank threads
  // sharing one GPU, 32768^2 at 2 ranks: 904 -> 971 ms with it, profiles/r04_ab_side_stream_logical_ranks.txt)
  const bool want_side = c->lane_side && P == 1;
  if (want_side && !c->hi2) HIPCHECK(hipStreamCreateWithPriority(&c->hi2, hipStreamNonBlocking, c->hi_priority));
  // r6: the commit of an accepted panel (12 us of copies into the matrix) leaves the lane at P > 1: it runs on the
  // communication stream BEHIND the panel's broadcast (which sends the group buffer, not the matrix); the broadcast -- the
  // critical chain of the column split -- leaves at "T and verdict final".  (One rank: DHQR_TUNE commit_off=1 puts it on a
  // stream of its own, which lost badly: see dhqr_ctx::commit_off.)
  const bool commit_off = c->commit_off > 0 || (c->commit_off < 0 && P > 1);
  if (commit_off) {
    if (P == 1 && !c->cstream) HIPCHECK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    for (hipEvent_t &e : c->ev_commit)
      if (!e) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  hipStream_t sW = c->stream, sL = c->hi, sC = S.comm, sX = want_side ? c->hi2 : nullptr;
  hipStream_t sK = commit_off ? (P == 1 ? c->cstream : sC) : nullptr;
  // Lane side stream (r4): what needs a panel's V but not its T runs on sX beside the panel's second Gram product, k_build_t
  // and the commit -- Y = V_a' C_b for the pair's second panel, the pair's cross term V_b' V_a, the quad's V_2' V_1.  Only
  // for panels this rank factors itself on the asynchronous fast path (a received panel has no "V final" event).
  const bool side = c->lane_side && sX != nullptr;
  auto on = [&](hipStream_t s, int wsi) { c->stream = s; c->cur_ws = wsi; };
  const int saved_epoch = c->epoch;
  auto apply_group = [&](int g, int64_t lstart, int64_t ncols) -> int32_t {  // group g -> local columns [lstart, lstart+ncols)
    if (ncols <= 0) return DHQR_OK;
    const CsGroup &gr = groups[g];
    const CsGroupBuf gb = gview(g);
    double *C = pr.A + gr.a * NB + lstart * lda;
    c->epoch = (int)gr.last();
    int32_t rc;
    if (gr.np == 2)
      rc = pair_apply(c, gb.VA, gb.ldv, gb.rows_a, gb.pa().T, gb.pb().T, gb.Sba, C, ncols, lda);
    else
      rc = panel_apply(c, gb.pa(), gb.rows_a, C, ncols, lda, 1);
    return rc;
  };
  auto apply_step = [&](int si, int64_t lstart, int64_t ncols) -> int32_t {  // step si -> local columns [lstart, lstart+ncols)
    const CsStep &st = steps[si];
    if (st.ng == 1) return apply_group(st.g0, lstart, ncols);
    if (ncols <= 0) return DHQR_OK;
    const CsGroupBuf g1 = gview(st.g0), g2 = gview(st.g0 + 1);
    c->epoch = (int)groups[st.g0 + 1].last();
    return quad_apply(c, g1.VA, g2.VA, g1.ldv, g1.rows_a, g1.pa().T, g1.pb().T, g1.Sba, g2.pa().T, g2.pb().T, g2.Sba, g2.S21,
                      pr.A + groups[st.g0].a * NB + lstart * lda, ncols, lda);
  };
  // group h's last panel went through the asynchronous fast path on this rank with a "V final" event (ev_v) recorded
  std::vector<char> v_final((size_t)G, 0);
  // Does wide step si apply itself to the blocks of group glast + 2 FIRST, as a separate head with its own event?  At P > 1
  // always (that group's owner needs its block early: its wide launches are short and its lane is not shut out for long).
  // At P == 1 the only candidate is the second pair of a quad: six latency-bound launches on 256 columns (~0.55 ms, 45
  // times per 32768^2) where the same columns would be two column tiles more of the wide launches -- yet folding the head
  // into the wide launches was SLOWER on the same box (r4, profiles/r04_ab_quad_head.txt: the wide kernels gain 8 ms, the
  // total loses 3 ms at 32768^2, 3 ms at 16384^2, 4 ms at 24576^2) and the switch that did it is gone.
  auto has_head = [&](int si) -> bool {
    const int gl = steps[si].g0 + steps[si].ng - 1;
    if (gl + 2 >= G) return false;
    if (P > 1) return true;
    return steps[step_of[(size_t)(gl + 2)]].ng == 2 && steps[step_of[(size_t)(gl + 2)]].g0 == gl + 1;
  };
  // Lane sections outside the panel factorisations (narrow updates, cross terms) are never timed: every timed section is
  // two event records on the lane -- the critical chain -- and a profiled 32768^2 run had ~2000 of them per factorisation
  // (~4 ms of bubbles inside bench.py's timed region, r4); the panel factorisations keep their own pair.
  auto lane_begin = [&](bool &was) -> int32_t {
    was = c->profiling;
    c->profiling = false;
    return DHQR_OK;
  };
  auto lane_end = [&](bool was) -> int32_t {
    c->profiling = was;
    return DHQR_OK;
  };

  // produce group h: owners update + factor their panels (lane), everybody takes part in the broadcasts (comm),
  // the pair cross term is built (lane).  prev = h - 1 (or -1 for the first group).
  auto produce = [&](int h) -> int32_t {
    const CsGroup &gr = groups[h];
    const CsGroupBuf gb = gview(h);
    const int prev = h - 1;
    const int sh = step_of[(size_t)h];
    const bool second_of_quad = steps[sh].ng == 2 && h == steps[sh].g0 + 1;
    // writers into this buffer slot must wait for the readers of group h - CS_NGB
    auto guard = [&](hipStream_t s) -> int32_t {
      if (h >= CS_NGB) {
        HIPCHECK(hipStreamWaitEvent(s, S.ev_wide[step_of[(size_t)(h - CS_NGB)] % CS_EVR], 0));
        HIPCHECK(hipStreamWaitEvent(s, S.ev_lane[(h - CS_NGB + 1) % CS_EVR], 0));
        if (cm)
          for (int idx = 0; idx < groups[h - CS_NGB].np; ++idx) {
            const int64_t y = groups[h - CS_NGB].a + idx;
            if (pr.mine(y)) CHECK(comm_wait_consumed(cm, S.ticket[y % (2 * CS_EVR)], s));
          }
      }
      return DHQR_OK;
    };
    bool merged_update = false;
    bool v_event[2] = {false, false};  // ev_v recorded for panel idx of this group (this rank, fast path)
    for (int idx = 0; idx < gr.np; ++idx) {
      const int64_t x = gr.a + idx, w = pr.width(x), rows = m - x * NB;
      const PanelBuf pbx = idx == 0 ? gb.pa() : gb.pb();
      const int pe = (int)(x % (2 * CS_EVR));
      if (pr.mine(x)) {
        on(sL, 1);
        CHECK(guard(sL));
        const int64_t lc = pr.lcol(x);
        bool was = false;
        if (prev >= 0 && !merged_update) {
          // The block receives here what the wide stream leaves to the lane: the whole previous STEP when this group
          // opens a step (the block must carry everything before that step: the wide update of step sh - 2), or the
          // quad's first pair when this group is the quad's second (the block is the head of step sh - 1).
          if (second_of_quad) {
            if (sh >= 1) HIPCHECK(hipStreamWaitEvent(sL, (has_head(sh - 1) ? S.ev_head : S.ev_wide)[(sh - 1) % CS_EVR], 0));
          } else if (sh >= 2) {
            // the head of wide step sh - 2 is the group two behind that step's last one: this group when step sh - 1 is a
            // single group, the SECOND pair of step sh - 1 when that is a quad (then this block is in the rest)
            const bool in_head = P > 1 && has_head(sh - 2) && h == steps[sh - 2].g0 + steps[sh - 2].ng + 1;
            HIPCHECK(hipStreamWaitEvent(sL, (in_head ? S.ev_head : S.ev_wide)[(sh - 2) % CS_EVR], 0));
          }
          int64_t ncols = w;
          if (gr.np == 2 && idx == 0 && pr.mine(x + 1) && pr.lcol(x + 1) == lc + w) {
            ncols += pr.width(x + 1);  // both panels are local and adjacent (a whole cyclic block): one 256-column update
            merged_update = true;
          }
          CHECK(lane_begin(was));
          const int32_t rc = second_of_quad ? apply_group(prev, lc, ncols) : apply_step(sh - 1, lc, ncols);
          CHECK(lane_end(was));
          CHECK(rc);
        }
        if (idx == 1) {  // panel a of this group -> block b
          if (!pr.mine(gr.a)) HIPCHECK(hipStreamWaitEvent(sL, S.ev_recv[(int)(gr.a % (2 * CS_EVR))], 0));
          double *Cb = pr.A + gr.a * NB + lc * lda;
          const int pa_ = (int)(gr.a % (2 * CS_EVR));
          CHECK(lane_begin(was));
          c->epoch = (int)gr.a;
          int32_t rc = DHQR_OK;
          if (side && v_event[0] && (merged_update || prev < 0)) {
            // Y = V_a' C_b on the side stream as soon as V_a is final (block b was brought up to date together with block a,
            // before panel a was factored -- merged_update -- so ev_v is recorded behind that on the lane); T_a' Y and the
            // subtraction follow on the lane once panel a is verified and committed.  Both calls see workspace 2.
            on(sX, 2);
            HIPCHECK(hipStreamWaitEvent(sX, S.ev_v[pa_], 0));
            rc = panel_apply(c, gb.pa(), gb.rows_a, Cb, w, lda, 1, DHQR_NBV, 1);
            HIPCHECK(hipEventRecord(S.ev_y[pa_], sX));
            on(sL, 2);
            HIPCHECK(hipStreamWaitEvent(sL, S.ev_y[pa_], 0));
            if (rc == DHQR_OK) rc = panel_apply(c, gb.pa(), gb.rows_a, Cb, w, lda, 1, DHQR_NBV, 2);
            on(sL, 1);
          } else {
            rc = panel_apply(c, gb.pa(), gb.rows_a, Cb, w, lda, 1);
          }
          CHECK(lane_end(was));
          CHECK(rc);
          hipLaunchKernelGGL(k_zero_rows, dim3(DHQR_NBV), dim3(128), 0, sL, gb.VB, gb.ldv, (int)NB);
        }
        double *Pp = pr.A + x * NB + lc * lda;
        c->epoch = saved_epoch;
        c->tt_keep = (c->tc_base && P == 1) ? c->tc_base + x * (NB * NB) : nullptr;  // kept T factors (dhqr_api.hip)
        bool deferred = false;
        if (panel_fast_eligible(c, rows, w) && !(robust_first && x == kstart)) {
          deferred = sK != nullptr;
          CHECK(panel_fast_enqueue(c, Pp, rows, lda, pr.alpha + x * NB, pbx, c->cholqr_passes, (int)x, side ? S.ev_v[pe] : nullptr,
                                   deferred ? S.ev_t[pe] : nullptr));
          v_event[idx] = side;
          fast_idx.push_back(x);
        } else {
          // Short / partial panels (the last one or two of a factorisation) and the panel a resumed run starts
          // with use kernels that write unconditionally.  Inside a run they may only execute if no earlier
          // panel was rejected, which needs the one status read of the lane the asynchronous path avoids.
          bool run_it = true;
          if (x != kstart) {
            int f = 0;
            CHECK(status_read(c, &f));  // the lane has already waited for every earlier panel's status
            run_it = (f == INT_MAX);
          }
          if (run_it) CHECK(factor_panel_sync(c, Pp, rows, w, lda, pr.alpha + x * NB, pbx));
          hipLaunchKernelGGL(k_set_statword, dim3(1), dim3(64), 0, sL, (const int *)c->dstat, pbx.alpha + DHQR_NBV);
        }
        c->tt_keep = nullptr;
        auto commit_on = [&](hipStream_t s, int part) -> int32_t {
          CHECK(panel_commit_enqueue(c, s, Pp, rows, lda, pr.alpha + x * NB, pbx, (int)x, part));
          if (part != 1) HIPCHECK(hipEventRecord(c->ev_commit[x & 1], s));
          return DHQR_OK;
        };
        if (deferred && P == 1) {  // commit on its own stream behind "T and verdict final"
          HIPCHECK(hipStreamWaitEvent(sK, S.ev_t[pe], 0));
          CHECK(commit_on(sK, 0));
          HIPCHECK(hipEventRecord(S.ev_ready[pe], sK));
        } else if (!deferred) {
          HIPCHECK(hipEventRecord(S.ev_ready[pe], sL));
        }
        // host-in / host-out drop-in: the column block of a committed panel is final and may leave for the host
        if (c->panel_hook && P == 1) CHECK(c->panel_hook(c->panel_hook_arg, x, S.ev_ready[pe]));
        if (cm && P > 1) {
          HIPCHECK(hipStreamWaitEvent(sC, deferred ? S.ev_t[pe] : S.ev_ready[pe], 0));
          if (deferred) CHECK(commit_on(sC, 1));  // alpha -> the buffer's tail: travels with the broadcast
          CHECK(comm_bcast(cm, gb.region(idx), gb.region_elems(), pr.r, sC, &S.ticket[pe]));
          HIPCHECK(hipEventRecord(S.ev_recv[pe], sC));
          if (deferred) CHECK(commit_on(sC, 2));  // the rest behind the broadcast: the peers need the group buffer, not the matrix
        }
      } else {
        CHECK(guard(sC));
        CHECK(comm_bcast(cm, gb.region(idx), gb.region_elems(), pr.owner(x), sC, nullptr));
        hipLaunchKernelGGL(k_adopt_status, dim3(1), dim3(64), 0, sC, (const double *)(pbx.alpha + DHQR_NBV), c->dstat);
        hipLaunchKernelGGL(k_commit_alpha, dim3(1), dim3(DHQR_NBV), 0, sC, (const double *)pbx.alpha, (int)w,
                           pr.alpha + x * NB, (double *)nullptr, (const int *)c->dstat, (int)x);
        HIPCHECK(hipEventRecord(S.ev_recv[pe], sC));
      }
    }
    // assemble the group on the lane: needs every panel of the group locally
    on(sL, 1);
    for (int idx = 0; idx < gr.np; ++idx)
      if (!pr.mine(gr.a + idx)) HIPCHECK(hipStreamWaitEvent(sL, S.ev_recv[(int)((gr.a + idx) % (2 * CS_EVR))], 0));
    if (gr.np == 2 && gr.last() + 1 < K) {
      bool was = false;
      CHECK(lane_begin(was));
      int32_t rc = DHQR_OK;
      if (side && v_event[1]) {
        // the cross terms need the reflectors of this pair (and of the quad's first pair, complete long ago), not their T:
        // on the side stream behind "V_b final", beside panel b's verification and commit; the lane waits for them here
        const int pb_ = (int)(gr.last() % (2 * CS_EVR));
        on(sX, 2);
        HIPCHECK(hipStreamWaitEvent(sX, S.ev_v[pb_], 0));
        if (!pr.mine(gr.a)) HIPCHECK(hipStreamWaitEvent(sX, S.ev_recv[(int)(gr.a % (2 * CS_EVR))], 0));
        if (second_of_quad) HIPCHECK(hipStreamWaitEvent(sX, S.ev_lane[(h - 1) % CS_EVR], 0));  // the first pair assembled
        rc = pair_cross_gram(c, gb.VA, gb.ldv, gb.rows_a, gb.Sba, &c->spart2);
        if (rc == DHQR_OK && second_of_quad) {
          const CsGroupBuf g1 = gview(h - 1);
          rc = quad_cross_gram(c, g1.VA, gb.VA, gb.ldv, g1.rows_a, gb.S21, &c->spart2);
        }
        HIPCHECK(hipEventRecord(S.ev_x[h % CS_EVR], sX));
        on(sL, 1);
        HIPCHECK(hipStreamWaitEvent(sL, S.ev_x[h % CS_EVR], 0));
      } else {
        rc = pair_cross_gram(c, gb.VA, gb.ldv, gb.rows_a, gb.Sba);
        if (rc == DHQR_OK && second_of_quad) {
          const CsGroupBuf g1 = gview(h - 1);
          rc = quad_cross_gram(c, g1.VA, gb.VA, gb.ldv, g1.rows_a, gb.S21);
        }
      }
      CHECK(lane_end(was));
      CHECK(rc);
    }
    HIPCHECK(hipEventRecord(S.ev_group[h % CS_EVR], sL));
    HIPCHECK(hipEventRecord(S.ev_lane[h % CS_EVR], sL));
    v_final[(size_t)h] = v_event[gr.np - 1] ? 1 : 0;
    LAUNCHCHECK();
    return DHQR_OK;
  };

  auto body = [&]() -> int32_t {
    // order the lane and the comm stream after whatever the caller queued (e.g. the fill)
    HIPCHECK(hipEventRecord(S.ev_start, sW));
    HIPCHECK(hipStreamWaitEvent(sL, S.ev_start, 0));
    HIPCHECK(hipStreamWaitEvent(sC, S.ev_start, 0));
    if (sX) HIPCHECK(hipStreamWaitEvent(sX, S.ev_start, 0));
    if (sK && sK != sC) HIPCHECK(hipStreamWaitEvent(sK, S.ev_start, 0));
    for (int q = 0; q < steps[0].ng; ++q) CHECK(produce(steps[0].g0 + q));
    for (int si = 0; si < NS; ++si) {
      const int glast = steps[si].g0 + steps[si].ng - 1;
      if (groups[glast].last() + 1 >= K) break;  // nothing to the right of this step
      // ---- wide stream: step si -> local blocks beyond the group that follows it (that group gets the step from the lane)
      on(sW, 0);
      const int64_t after_next = groups[glast + 1].last() + 1;
      int64_t lo = pr.local_from(after_next);
      // head: the blocks of group glast + 2 -- what the lane needs first (has_head above).  (Starting the head's Y products
      // behind "V of the step's last panel is final" was measured slower in round 4 and deleted: profiles/r04_ab_head_early.txt.)
      if (has_head(si)) {
        const int64_t hi = pr.local_from(groups[glast + 2].last() + 1);
        HIPCHECK(hipStreamWaitEvent(sW, S.ev_group[glast % CS_EVR], 0));
        CHECK(apply_step(si, lo, hi - lo));
        lo = hi;
      } else {
        HIPCHECK(hipStreamWaitEvent(sW, S.ev_group[glast % CS_EVR], 0));
      }
      HIPCHECK(hipEventRecord(S.ev_head[si % CS_EVR], sW));
      CHECK(apply_step(si, lo, pr.ncl - lo));
      HIPCHECK(hipEventRecord(S.ev_wide[si % CS_EVR], sW));
      // ---- lane + comm: the groups of step si + 1
      if (si + 1 < NS)
        for (int q = 0; q < steps[si + 1].ng; ++q) CHECK(produce(steps[si + 1].g0 + q));
    }
    // join: the caller's stream owns the result
    on(sW, 0);
    HIPCHECK(hipEventRecord(S.ev_end, sL));
    HIPCHECK(hipStreamWaitEvent(sW, S.ev_end, 0));
    HIPCHECK(hipEventRecord(S.ev_end, sC));
    HIPCHECK(hipStreamWaitEvent(sW, S.ev_end, 0));
    if (sX) {
      HIPCHECK(hipEventRecord(S.ev_end, sX));
      HIPCHECK(hipStreamWaitEvent(sW, S.ev_end, 0));
    }
    if (sK && sK != sC) {
      HIPCHECK(hipEventRecord(S.ev_end, sK));
      HIPCHECK(hipStreamWaitEvent(sW, S.ev_end, 0));
    }
    return DHQR_OK;
  };
  int32_t rc = body();
  on(sW, 0);
  c->epoch = saved_epoch;
  if (rc == DHQR_OK) rc = status_read(c, failed);
  return rc;
}

// Workspaces of one rank sized up front for an m x n problem: nothing is (re)allocated while streams run.
static int32_t cs_prepare(const CsProblem &pr) {
  dhqr_ctx *c = pr.c;
  CHECK(cs_state_init(c));
  const int64_t NB = DHQR_NBV, m = pr.m;
  const size_t NN = (size_t)NB * NB;
  for (int s = 0; s < CS_NGB; ++s) CHECK(ensure(c, c->cs->gbuf[s], cs_gbuf_elems(m)));
  CHECK(ensure(c, c->cs->vt, (size_t)panel_elems(m)));
  CHECK(ensure(c, c->vts, (size_t)panel_elems(m)));
  const size_t ncmax = (size_t)std::max<int64_t>(pr.ncl, 2 * NB);
  const size_t ntmax = (ncmax + 127) / 128;
  // split-K partials of k_gemm_tn (128 rows) / k_gemm_tn2 (256 rows; stream-K with R row groups: R x P pieces of 2 ntiles
  // matrices, P ntiles <= G / R + 2 ntiles)
  const size_t w1cap = NN * std::max<size_t>(6144 + 2 * ntmax + 128, 640 + 32 * ntmax);
  for (int s = 0; s < 3; ++s) {
    CHECK(ensure(c, c->ws[s].w1, s == 0 ? w1cap : NN * 2200));
    CHECK(ensure(c, c->ws[s].w1r, s == 2 ? (size_t)4 * NB * 2 * NB : (size_t)4 * NB * ncmax));  // Y_1, Y_2 / [W_1; W_2] of a quad step
    CHECK(ensure(c, c->ws[s].w2, s == 2 ? (size_t)4 * NB * 2 * NB : (size_t)4 * NB * ncmax));   // ([2]: the side stream's narrow products)
  }
  CHECK(ensure(c, c->spart, (size_t)512 * NN));  // Gram partials; 128 slabs of the 256 x 256 cross term of a quad
  CHECK(ensure(c, c->spart2, (size_t)512 * NN));  // the same for the cross terms built on the lane's side stream
  CHECK(ensure(c, c->sfull, NN));
  CHECK(ensure(c, c->rbuf, 2 * panel_rbuf_elems()));
  if (c->cholqr_passes == 3) CHECK(ensure(c, c->tsq, TsqrLocal::elems(m)));  // TSQR-HR for every panel
  CHECK(ensure(c, c->scratch, 4096));
  return DHQR_OK;
}

// Quad steps at P > 1: the schedule (cs_plan) is part of the SPMD program, so the ranks must take the SAME decision, while
// the conditions of the 16-byte path (even m, even lda, 16-byte aligned block) are local: one tiny all-reduce per
// factorisation counts the ranks that do not meet them.
static int32_t cs_agree_quads(CsProblem &pr) {
  dhqr_ctx *c = pr.c;
  if (pr.P == 1 || !pr.cm || !c->quad) {
    pr.quads_ok = 0;
    return DHQR_OK;
  }
  const double bad = (pr.ncl == 0 || (pr.m % 2 == 0 && pr.lda % 2 == 0 && aligned16(pr.A))) ? 0.0 : 1.0;
  CHECK(ensure(c, c->scratch, 4096));
  double h = bad;
  HIPCHECK(hipMemcpyAsync(c->scratch.p, &h, sizeof(double), hipMemcpyHostToDevice, c->stream));
  CHECK(comm_allreduce_sum(pr.cm, c->scratch.p, 1, c->stream));
  HIPCHECK(hipMemcpyAsync(&h, c->scratch.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  pr.quads_ok = (h == 0.0) ? 1 : 0;
  return DHQR_OK;
}

// householder!(A, alpha) / householder!(A::DArray, alpha) (src:113-120) on this rank's block-cyclic columns.
// Collective over pr.cm; synchronous on return (one status read per pass; a second pass only after a rejected panel).
static int32_t cs_factor(const CsProblem &pr_in) {
  CsProblem pr = pr_in;
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV;
  CHECK(cs_prepare(pr));
  if (pr.quads_ok < 0) CHECK(cs_agree_quads(pr));
  CHECK(status_reset(c));
  int64_t ks = 0;
  bool robust = false;
  std::vector<int64_t> fast_idx;
  for (int pass = 0; ks < pr.K; ++pass) {
    if (pass > pr.K + 2) return set_err(DHQR_EINVAL, "internal error: the blocked driver does not make progress");
    int failed = INT_MAX;
    fast_idx.clear();
    CHECK(cs_run(pr, ks, robust, &failed, fast_idx));
    for (int64_t x : fast_idx)
      if (x < failed) {
        c->n_fast++;
        if (c->cholqr_passes == 3) c->n_tsqr++;
      }
    if (failed == INT_MAX) break;
    c->n_resume++;
    // ---- resume: panels < failed are committed; matrix updates with epoch >= failed did not run
    CHECK(status_reset(c));
    // Committed panels of the failed panel's group / step were only applied as far as the lane needed them: apply them
    // to the rest before resuming.  Pair (a, b), b rejected: a reached block b only.  Quad (a, b)(c, d): c rejected ->
    // a, b reached the blocks of c and d only; d rejected -> so did a, b, and c reached block d only.
    std::vector<CsGroup> groups;
    std::vector<CsStep> steps;
    cs_plan(pr, ks, groups, steps);
    std::vector<std::pair<int64_t, int64_t>> redo;  // (committed panel, first panel of the blocks it has not reached)
    for (const CsStep &st : steps) {
      const CsGroup &g1 = groups[(size_t)st.g0], &gl = groups[(size_t)(st.g0 + st.ng - 1)];
      if (failed < g1.a || failed > gl.last()) continue;
      if (st.ng == 1 || failed <= g1.last()) {
        if (failed == g1.a + 1) redo.push_back({g1.a, failed + 1});
      } else {
        for (int64_t y = g1.a; y <= g1.last(); ++y) redo.push_back({y, gl.last() + 1});
        if (failed == gl.last()) redo.push_back({gl.a, failed + 1});
      }
    }
    for (const auto &rd : redo) {
      const int64_t a = rd.first, rows_a = pr.m - a * NB;
      const PanelBuf pb = vt_view(c->cs->vt.p, rows_a);
      if (pr.mine(a)) CHECK(panel_pack_and_t(c, pr.A + a * NB + pr.lcol(a) * pr.lda, rows_a, NB, pr.lda, pr.alpha + a * NB, pb));
      if (pr.cm && pr.P > 1) CHECK(comm_bcast(pr.cm, c->cs->vt.p, panel_elems(rows_a), pr.owner(a), c->stream, nullptr));
      const int64_t lo = pr.local_from(rd.second);
      CHECK(panel_apply(c, pb, rows_a, pr.A + a * NB + lo * pr.lda, pr.ncl - lo, pr.lda, 1));
      if (pr.cm && pr.P > 1) HIPCHECK(hipStreamSynchronize(c->stream));  // vt is reused by the next resume
    }
    ks = failed;
    robust = true;
  }
  return DHQR_OK;
}

// ||A - QR||_F / ||A||_F with A = u01(seed) regenerated on the device: every rank forms its columns of Q*R by
// re-applying the panels in reverse order (one broadcast per panel again).  dW, dA0: m x ncl scratch (ld = m).
static int32_t cs_residual(const CsProblem &pr, uint64_t seed, double *dW, double *dA0, double *hrel) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV, m = pr.m, ncl = pr.ncl;
  CHECK(cs_prepare(pr));
  const int64_t ldw = m;
  if (ncl > 0) {
    dim3 grid((unsigned)std::min<int64_t>((m + 255) / 256, 128), (unsigned)std::min<int64_t>(ncl, 32768));
    hipLaunchKernelGGL(k_form_r0, grid, dim3(256), 0, c->stream, (const double *)pr.A, pr.lda, (const double *)pr.alpha, m,
                       ncl, dW, ldw, CS_CB, pr.P, pr.r);
  }
  const bool was = c->profiling;
  c->profiling = false;
  auto body = [&]() -> int32_t {
    for (int64_t k = pr.K - 1; k >= 0; --k) {
      const int64_t rows = m - k * NB;
      const PanelBuf pb = vt_view(c->cs->vt.p, rows);
      if (pr.mine(k))
        CHECK(panel_pack_and_t(c, pr.A + k * NB + pr.lcol(k) * pr.lda, rows, pr.width(k), pr.lda, nullptr, pb));
      if (pr.cm && pr.P > 1) CHECK(comm_bcast(pr.cm, c->cs->vt.p, panel_elems(rows), pr.owner(k), c->stream, nullptr));
      const int64_t lo = pr.local_from(k);
      if (ncl - lo > 0) CHECK(panel_apply(c, pb, rows, dW + k * NB + lo * ldw, ncl - lo, ldw, 0));
      // LOCAL transport: the root must not repack the buffer before the others have copied it
      if (pr.cm && pr.cm->kind == COMM_LOCAL) HIPCHECK(hipStreamSynchronize(c->stream));
      if (pr.cm && pr.cm->kind == COMM_LOCAL) CHECK(comm_host_barrier(pr.cm));
    }
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  double h[2] = {0.0, 0.0};
  CHECK(ensure(c, c->scratch, 4096));
  if (ncl > 0) {
    const int64_t total = m * ncl;
    const unsigned gridf = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fill_uniform, dim3(gridf), dim3(256), 0, c->stream, dA0, m, ncl, m, seed, m, (int64_t)0, CS_CB, pr.P,
                       pr.r);
    const int nblk = 1024;
    hipLaunchKernelGGL(k_diff_norms, dim3(nblk), dim3(256), 0, c->stream, (const double *)dA0, m, (const double *)dW, ldw, m,
                       ncl, c->scratch.p);
    hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(256), 0, c->stream, (const double *)c->scratch.p, nblk, c->scratch.p + 2048);
  } else {
    HIPCHECK(hipMemsetAsync(c->scratch.p + 2048, 0, 2 * sizeof(double), c->stream));
  }
  LAUNCHCHECK();
  if (pr.cm && pr.P > 1) CHECK(comm_allreduce_sum(pr.cm, c->scratch.p + 2048, 2, c->stream));
  HIPCHECK(hipMemcpyAsync(h, c->scratch.p + 2048, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  *hrel = std::sqrt(h[0] / h[1]);
  return DHQR_OK;
}

// solve_householder!(b, H, alpha) on the column split (src:226-282).  db (length m, the same on every rank) is
// overwritten; x = db[0:n] on every rank.  Q'b: the owner of each block applies it and hands the updated tail
// of b on (the reference walks the owners sequentially with b in shared memory, src:226-230).  Back substitution:
// every rank accumulates the contribution of ITS columns to the rows above; per block one all-reduce sums those
// partial dots (the reference's sum(fetch.(futures)), src:262-266), the owner solves the diagonal block and
// broadcasts x.  du: scratch of m + 128 doubles.
// r6: the column split's solve at P > 1 on the kernels of dhqr_qtb.h, one cyclic block (a PAIR of panels, both on one rank)
// per step -- half the collectives of the per-panel form below and none of its re-packing:
//   pre-pass (no collective, every rank on its own pairs, independent of b): S = V'V and T' of every local panel by the
//     batched Gram / sum / inverse kernels, each pair addressed as the (m - c0) x 256 sub-matrix whose top-left corner
//     is the pair's diagonal;
//   Q'b: the pair's owner runs the panel steps of k_qtb_step on that sub-matrix and b[c0:m] (V read in place, b never
//     leaves the workgroup between update and dots), then broadcasts b[c0:m] -- the reference hands b from owner to
//     owner through shared memory (src:226-230);
//   back substitution, pairs from the right: ONE all-reduce of the 256 partial sums the ranks have accumulated for the
//     pair's rows (the reference's sum(fetch.(futures)), src:262-266), the owner solves its 256 x 256 triangle with the
//     pipelined kernel (two workgroups) and broadcasts the 256 solved entries, every rank subtracts the contribution of
//     ITS columns... only the owner holds the pair's columns: it subtracts R[0:c0, pair] x from its accumulator.
// du: m doubles of scratch (the accumulator).  Synchronises only where the LOCAL transport needs it.
static int32_t cs_solve_pairs(const CsProblem &pr, double *db, double *du) {
  dhqr_ctx *c = pr.c;
  dhqr_comm *cm = pr.cm;
  const int64_t NB = DHQR_NBV, m = pr.m, lda = pr.lda;
  const int64_t npairs = (pr.K + 1) / 2;
  auto pair_c0 = [&](int64_t q) { return q * CS_CB; };
  auto pair_cols = [&](int64_t q) { return std::min<int64_t>(CS_CB, pr.n - q * CS_CB); };
  auto sync_local = [&]() -> int32_t {
    if (cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
    return DHQR_OK;
  };
  // ---- workspaces: T' / S per local panel, Gram partials of the tallest pair, [ypart | w | x | ints]
  std::vector<int64_t> mine;
  for (int64_t q = 0; q < npairs; ++q)
    if (pr.mine(2 * q)) mine.push_back(q);
  const size_t nlp = 2 * mine.size() + 2;
  CHECK(ensure(c, c->sv_T, nlp * QTB_NB2));
  CHECK(ensure(c, c->sv_S, nlp * QTB_NB2));
  auto pair_rps = [&](int64_t mq, int64_t nq) {
    const int npq = (int)((nq + NB - 1) / NB);
    int64_t total = 0;
    for (int k = 0; k < npq; ++k) total += mq - (int64_t)k * NB;
    int64_t rps = ((total / 1024 + 15) / 16) * 16;
    return std::min<int64_t>(std::max<int64_t>(rps, 256), 4096);
  };
  size_t max_units = 4;
  std::vector<int> tables;  // three ints per local pair: the unit table of its Gram pre-pass
  for (int64_t q : mine) {
    const int64_t mq = m - pair_c0(q), nq = pair_cols(q), rps = pair_rps(mq, nq);
    const int npq = (int)((nq + NB - 1) / NB);
    int u[3] = {0, 0, 0};
    for (int k = 0; k < npq; ++k) u[k + 1] = u[k] + (int)((mq - (int64_t)k * NB + rps - 1) / rps);
    if (npq == 1) u[2] = u[1];
    max_units = std::max<size_t>(max_units, (size_t)u[npq]);
    tables.insert(tables.end(), u, u + 3);
  }
  CHECK(ensure(c, c->sv_part, max_units * QTB_NB2));
  const int64_t SSmax = 128, maxsl = std::max<int64_t>(8, std::min<int64_t>(c->ncu, 256));
  const size_t n_ypart = (size_t)std::max<int64_t>(maxsl + 2, (m + 63) / 64) * QTB_NB, n_w = 3 * (size_t)QTB_NB;
  const size_t n_ints = tables.size() + 3 * mine.size() + 2 * (size_t)npairs + 8;
  CHECK(ensure(c, c->sv_small, n_ypart + n_w + (n_ints + 1) / 2 + 16));
  (void)SSmax;
  double *ypart = c->sv_small.p, *wbuf = ypart + n_ypart;
  int *ints = reinterpret_cast<int *>(wbuf + n_w);
  int *tab_dev = ints, *counters = tab_dev + tables.size(), *flags = counters + 3 * mine.size(), *zero = flags + 2 * npairs;
  HIPCHECK(hipMemsetAsync(ints, 0, n_ints * sizeof(int), c->stream));
  if (!tables.empty()) {
    // (the table is a host vector that dies with this call: the copy is waited for below, before anything else can fail)
    HIPCHECK(hipMemcpyAsync(tab_dev, tables.data(), tables.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
  }
  const bool was = c->profiling;
  CHECK(prof_begin(c, CAT_SOLVE));
  c->profiling = false;
  int *err = c->zflags + DHQR_PIPE_ERR_OFFSET;
  auto body = [&]() -> int32_t {
    // ---- pre-pass on this rank's pairs
    for (size_t i = 0; i < mine.size(); ++i) {
      const int64_t q = mine[i], c0 = pair_c0(q), mq = m - c0, nq = pair_cols(q), rps = pair_rps(mq, nq);
      const int npq = (int)((nq + NB - 1) / NB);
      const double *sub = pr.A + c0 + pr.lcol(2 * q) * lda;
      const bool vec = (lda % 2 == 0) && (mq % 2 == 0) && aligned16(sub);
      const int *tab = tab_dev + 3 * i;
      const int nunits = tables[3 * i + npq];
      if (vec)
        hipLaunchKernelGGL((k_gemm_tn_gram_batch<2>), dim3((unsigned)nunits), dim3(256), 0, c->stream, sub, lda, mq, nq, rps, tab,
                           npq, c->sv_part.p, (const int *)zero);
      else
        hipLaunchKernelGGL((k_gemm_tn_gram_batch<1>), dim3((unsigned)nunits), dim3(256), 0, c->stream, sub, lda, mq, nq, rps, tab,
                           npq, c->sv_part.p, (const int *)zero);
      hipLaunchKernelGGL(k_qtb_sum_gram, dim3((unsigned)npq, 16), dim3(256), 0, c->stream, (const double *)c->sv_part.p, tab, nq,
                         c->sv_S.p + 2 * i * QTB_NB2, (const int *)zero);
      hipLaunchKernelGGL(k_build_t_batch, dim3((unsigned)npq), dim3(1024), 0, c->stream, (const double *)(c->sv_S.p + 2 * i * QTB_NB2),
                         nq, c->sv_T.p + 2 * i * QTB_NB2, (const int *)zero);
    }
    LAUNCHCHECK();
    // ---- b <- Q'b, pair by pair (src:215-242)
    size_t li = 0;
    for (int64_t q = 0; q < npairs; ++q) {
      const int64_t c0 = pair_c0(q), mq = m - c0, nq = pair_cols(q);
      const int npq = (int)((nq + NB - 1) / NB);
      if (pr.mine(2 * q)) {
        const double *sub = pr.A + c0 + pr.lcol(2 * q) * lda;
        const bool vec = (lda % 2 == 0) && (mq % 2 == 0) && aligned16(sub) && aligned16(db + c0);
        const int VEC = (c->qtb_vec == 1 || !vec) ? 1 : (c->qtb_vec == 2 ? 2 : (mq >= 16384 ? 2 : 1));
        const int64_t SS = 64 * VEC;
        const int64_t sl = SS * ((mq + SS * maxsl - 1) / (SS * maxsl));
        const int64_t nsl = (mq + sl - 1) / sl;
        const double *Tt = c->sv_T.p + 2 * li * QTB_NB2;
        for (int k = 0; k <= npq; ++k) {
          const int64_t rfirst = (int64_t)(k >= 1 ? k - 1 : 0) * NB;
          const unsigned grid = (unsigned)(nsl - rfirst / sl);
          if (VEC == 2)
            hipLaunchKernelGGL((k_qtb_step<2>), dim3(grid), dim3(256), 0, c->stream, sub, lda, mq, nq, k, npq, sl, db + c0, Tt, Tt,
                               (const int *)zero, wbuf, ypart, counters + 3 * li, err, (int64_t)0, (double *)nullptr);
          else
            hipLaunchKernelGGL((k_qtb_step<1>), dim3(grid), dim3(256), 0, c->stream, sub, lda, mq, nq, k, npq, sl, db + c0, Tt, Tt,
                               (const int *)zero, wbuf, ypart, counters + 3 * li, err, (int64_t)0, (double *)nullptr);
        }
        LAUNCHCHECK();
        ++li;
      }
      CHECK(comm_bcast(cm, db + c0, mq, pr.owner(2 * q), c->stream, nullptr));
      CHECK(sync_local());
    }
    // ---- back substitution, pairs from the right (src:244-282)
    HIPCHECK(hipMemsetAsync(du, 0, (size_t)m * sizeof(double), c->stream));
    for (int64_t q = npairs - 1; q >= 0; --q) {
      const int64_t c0 = pair_c0(q), nq = pair_cols(q);
      CHECK(comm_allreduce_sum(cm, du + c0, nq, c->stream));
      if (pr.mine(2 * q)) {
        const double *sub = pr.A + c0 + pr.lcol(2 * q) * lda;
        hipLaunchKernelGGL(k_axpy_n, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, c->stream, db + c0, (const double *)(du + c0),
                           (int)nq);
        hipLaunchKernelGGL(k_backsub_pipe, dim3((unsigned)((nq + NB - 1) / NB)), dim3(BSP_THREADS), 0, c->stream, sub, lda,
                           (const double *)(pr.alpha + c0), db + c0, nq, flags + 2 * q, err);
        LAUNCHCHECK();
      }
      CHECK(comm_bcast(cm, db + c0, nq, pr.owner(2 * q), c->stream, nullptr));
      CHECK(sync_local());
      if (pr.mine(2 * q) && c0 > 0) {
        const double *cols = pr.A + pr.lcol(2 * q) * lda;  // the pair's columns from row 0: R above the pair
        hipLaunchKernelGGL(k_backsub_update_wide, dim3((unsigned)((c0 + 255) / 256)), dim3(256), 0, c->stream, cols, lda, du, c0,
                           (const double *)(db + c0), (int)nq);
        LAUNCHCHECK();
      }
    }
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  CHECK(prof_end(c));
  return DHQR_OK;
}

static int32_t cs_solve(const CsProblem &pr, double *db, double *du) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV, m = pr.m;
  dhqr_comm *cm = (pr.cm && pr.P > 1) ? pr.cm : nullptr;
  // one rank: the block-cyclic layout is the matrix itself -- the solve of dhqr_qtb.h (r5: Q'b in one persistent launch, the
  // pipelined back substitution; 8192^2 1.2 ms against 11.5 ms for the per-panel form below, which P > 1 keeps)
  if (!cm && pr.P == 1 && c->solve_pipe) {
    CHECK(prof_begin(c, CAT_SOLVE));
    const bool was1 = c->profiling;
    c->profiling = false;
    const int32_t rc1 = solve_pipelined(c, pr.A, m, pr.n, pr.lda, pr.alpha, db, true);
    c->profiling = was1;
    CHECK(rc1);
    CHECK(prof_end(c));
    return DHQR_OK;
  }
  if (cm && c->solve_pipe) return cs_solve_pairs(pr, db, du);
  CHECK(cs_prepare(pr));
  const bool was = c->profiling;
  CHECK(prof_begin(c, CAT_SOLVE));
  c->profiling = false;
  double *ds = du + m;  // 128 partial dots
  auto sync_local = [&]() -> int32_t {
    if (cm && cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
    return DHQR_OK;
  };
  auto body = [&]() -> int32_t {
    for (int64_t k = 0; k < pr.K; ++k) {
      const int64_t c0 = k * NB, rows = m - c0;
      if (pr.mine(k)) {
        const PanelBuf pb = vt_view(c->cs->vt.p, rows);
        CHECK(panel_pack_and_t(c, pr.A + c0 + pr.lcol(k) * pr.lda, rows, pr.width(k), pr.lda, nullptr, pb));
        CHECK(panel_apply(c, pb, rows, db + c0, 1, m, 1));
      }
      if (cm) {
        CHECK(comm_bcast(cm, db + c0, rows, pr.owner(k), c->stream, nullptr));
        CHECK(sync_local());
      }
    }
    HIPCHECK(hipMemsetAsync(du, 0, (size_t)(m + NB) * sizeof(double), c->stream));
    for (int64_t k = pr.K - 1; k >= 0; --k) {
      const int64_t c0 = k * NB, w = pr.width(k);
      HIPCHECK(hipMemcpyAsync(ds, du + c0, (size_t)w * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      if (cm) CHECK(comm_allreduce_sum(cm, ds, w, c->stream));
      if (pr.mine(k)) {
        // global column j of R is read at base + j*lda
        const double *base = pr.A + (pr.lcol(k) - c0) * pr.lda;
        hipLaunchKernelGGL(k_axpy1, dim3(1), dim3(128), 0, c->stream, db + c0, (const double *)ds, (int)w);
        CHECK(dhqr_backsub_block_f64(c, base, pr.lda, pr.alpha, db, c0, c0 + w, 1, 0));
      }
      if (cm) {
        CHECK(comm_bcast(cm, db + c0, w, pr.owner(k), c->stream, nullptr));
        CHECK(sync_local());
      }
      if (pr.mine(k) && c0 > 0) {
        const double *base = pr.A + (pr.lcol(k) - c0) * pr.lda;
        HIPCHECK(hipMemcpyAsync(du + c0, db + c0, (size_t)w * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        CHECK(dhqr_backsub_block_f64(c, base, pr.lda, pr.alpha, du, c0, c0 + w, 0, 1));
        HIPCHECK(hipMemsetAsync(du + c0, 0, (size_t)w * sizeof(double), c->stream));
      }
    }
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

// ---- the reference's DArray layout (src:115-120, test/runtests.jl:71) ----------------------------
// DistributedArrays' default split gives every process ONE contiguous column block (the first n % P processes get
// one column more).  The factorisation runs block-cyclically (contiguous blocks idle the owners of the early
// columns: <= 5.4x on 8 GPUs), so a caller holding the reference's layout converts on the way in and out: one
// broadcast of every rank's block per direction -- O(mn) traffic next to the O(mn^2) factorisation.
static inline void cs_contig_range(int64_t n, int P, int s, int64_t *lo, int64_t *hi) {
  const int64_t q = n / P, rem = n % P;
  *lo = s * q + std::min<int64_t>(s, rem);
  *hi = *lo + q + (s < rem ? 1 : 0);
}
// contiguous block (dBlk, m x w_r, ld ldb) -> block-cyclic pr.A.  dStage: m x max(n/P + 1, 128) doubles.
static int32_t cs_load_contiguous(const CsProblem &pr, const double *dBlk, int64_t ldb, double *dStage) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV, m = pr.m;
  dhqr_comm *cm = (pr.cm && pr.P > 1) ? pr.cm : nullptr;
  for (int s = 0; s < pr.P; ++s) {
    int64_t lo, hi;
    cs_contig_range(pr.n, pr.P, s, &lo, &hi);
    const int64_t wblk = hi - lo;
    if (wblk == 0) continue;
    if (s == pr.r)
      HIPCHECK(hipMemcpy2DAsync(dStage, m * sizeof(double), dBlk, ldb * sizeof(double), m * sizeof(double), wblk,
                                hipMemcpyDeviceToDevice, c->stream));
    if (cm) CHECK(comm_bcast(cm, dStage, m * wblk, s, c->stream, nullptr));
    // the pieces of [lo, hi) this rank owns in the block-cyclic layout, panel by panel
    for (int64_t k = lo / NB; k <= (hi - 1) / NB; ++k) {
      if (!pr.mine(k)) continue;
      const int64_t g0 = std::max<int64_t>(k * NB, lo), g1 = std::min<int64_t>(std::min<int64_t>((k + 1) * NB, hi), pr.n);
      if (g1 <= g0) continue;
      HIPCHECK(hipMemcpy2DAsync(pr.A + (pr.lcol(k) + g0 - k * NB) * pr.lda, pr.lda * sizeof(double), dStage + (g0 - lo) * m,
                                m * sizeof(double), m * sizeof(double), g1 - g0, hipMemcpyDeviceToDevice, c->stream));
    }
    if (cm && cm->kind == COMM_LOCAL) {  // the stage is reused by the next source rank
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
  }
  return DHQR_OK;
}
// block-cyclic pr.A -> this rank's contiguous block (dBlk, m x w_r, ld ldb): one broadcast per cyclic block.
static int32_t cs_store_contiguous(const CsProblem &pr, double *dBlk, int64_t ldb, double *dStage) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV, m = pr.m;
  dhqr_comm *cm = (pr.cm && pr.P > 1) ? pr.cm : nullptr;
  int64_t lo, hi;
  cs_contig_range(pr.n, pr.P, pr.r, &lo, &hi);
  for (int64_t k = 0; k < pr.K; ++k) {
    const int64_t w = pr.width(k), g0 = k * NB;
    if (pr.mine(k))
      HIPCHECK(hipMemcpy2DAsync(dStage, m * sizeof(double), pr.A + pr.lcol(k) * pr.lda, pr.lda * sizeof(double),
                                m * sizeof(double), w, hipMemcpyDeviceToDevice, c->stream));
    if (cm) CHECK(comm_bcast(cm, dStage, m * w, pr.owner(k), c->stream, nullptr));
    const int64_t i0 = std::max<int64_t>(g0, lo), i1 = std::min<int64_t>(g0 + w, hi);
    if (i1 > i0)
      HIPCHECK(hipMemcpy2DAsync(dBlk + (i0 - lo) * ldb, ldb * sizeof(double), dStage + (i0 - g0) * m, m * sizeof(double),
                                m * sizeof(double), i1 - i0, hipMemcpyDeviceToDevice, c->stream));
    if (cm && cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
  }
  return DHQR_OK;
}
