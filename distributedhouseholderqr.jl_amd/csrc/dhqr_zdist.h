// dhqr_zdist.h -- ComplexF64 column split: `householder!` for a matrix whose columns are distributed over P ranks
// (reference: src:122-148 for the owner's reflectors, src:141-143 for their fan-out, src:171-213 for the update of
// the columns a rank holds; the reference is generic over the element type, test/runtests.jl:42-63 runs ComplexF64).
//
// Map: cyclic blocks of DHQR_ZNB = 64 complex columns (one panel): panel k lives on rank k % P at local column
// (k / P) * 64.  Per panel ONE broadcast of [alpha (64 complex) | T | T' | the factored panel, rows x 64 complex]; every
// rank embeds the 64 reflectors itself (k_zpack_emb: the real 2 x 2 embedding of dhqr_complex.h, 2 rows x 128 -- twice
// the bytes, which therefore do not travel) and updates the columns it holds with the Float64 MFMA kernels (panel_apply).  Look-ahead: the owner of panel k + 1 updates that panel's 64 columns
// first, factors it and broadcasts it on the high-priority stream while the caller's stream applies panel k to
// everything beyond (the single-GPU schedule of dhqr_factor_c64_nb with a broadcast between "factored" and "applied").
// Every broadcast of a factorisation is issued on the high-priority stream, in panel order, on every rank.
#pragma once

static inline int64_t zcs_npanels(int64_t n) { return (n + DHQR_ZNB - 1) / DHQR_ZNB; }
// complex columns rank r holds
static inline int64_t zcs_local_cols(int64_t n, int P, int r) {
  const int64_t K = zcs_npanels(n);
  int64_t cols = 0;
  for (int64_t k = r; k < K; k += P) cols += std::min<int64_t>(DHQR_ZNB, n - k * DHQR_ZNB);
  return cols;
}
// local column of the first panel >= k that rank r holds (clamped to the rank's column count)
static inline int64_t zcs_local_from(int64_t k, int64_t n, int P, int r) {
  const int64_t owned_before = k / P + ((k % P) > r ? 1 : 0);
  return std::min<int64_t>(owned_before * DHQR_ZNB, zcs_local_cols(n, P, r));
}

// A: the rank's columns (m x zcs_local_cols complex, interleaved re/im, leading dimension lda complex elements);
// alpha: n complex, replicated (every rank ends with all of it).  cm == nullptr: one rank.
static int32_t zcs_factor(dhqr_ctx *c, dhqr_comm *cm, double *A, int64_t m, int64_t n, int64_t lda, double *alpha) {
  const int P = cm ? cm->nranks : 1, r = cm ? cm->rank : 0;
  const int64_t ZB = DHQR_ZNB, K = zcs_npanels(n), ncl = zcs_local_cols(n, P, r);
  // broadcast unit of a panel: [alpha: 2 ZB doubles | T, T' (panel tail) | the factored panel itself, (m - c0) x 64 complex,
  // compact] -- HALF the bytes of the embedded operand: every rank embeds the reflectors itself (k_zpack_emb) into one
  // of two local operand buffers.  Two units rotate.
  const size_t unit = (size_t)((2 * ZB + panel_tail_elems() + 2 * m * ZB + 15) & ~(int64_t)15);
  const size_t vloc = (size_t)(panel_ldv(2 * m) * DHQR_NBV);
  CHECK(ensure(c, c->vt, 2 * unit + 2 * vloc));
  double *buf[2] = {c->vt.p, c->vt.p + unit};
  double *Vl[2] = {c->vt.p + 2 * unit, c->vt.p + 2 * unit + vloc};
  auto operand = [&](int64_t k) { return tail_view(Vl[k & 1], panel_ldv(2 * (m - k * ZB)), buf[k & 1] + 2 * ZB); };
  int64_t tick[2] = {-1, -1};
  if (!c->zev[0])
    for (int i = 0; i < 5; ++i) HIPCHECK(hipEventCreateWithFlags(&c->zev[i], hipEventDisableTiming));
  hipEvent_t evP[2] = {c->zev[0], c->zev[1]}, evW[2] = {c->zev[2], c->zev[3]}, evS = c->zev[4];
  hipStream_t sW = c->stream, sL = c->hi;
  auto on = [&](hipStream_t st, int wsi) { c->stream = st; c->cur_ws = wsi; };
  {  // the workspaces of both streams, sized up front: nothing is (re)allocated while the two streams run
    const size_t NN = (size_t)DHQR_NBV * DHQR_NBV, ncmax = (size_t)std::max<int64_t>(ncl, 2 * DHQR_NBV);
    for (int wsi = 0; wsi < 2; ++wsi) {
      CHECK(ensure(c, c->ws[wsi].w1, NN * (wsi == 0 ? 2 * ((ncmax + 127) / 128) + 2600 : 520)));
      CHECK(ensure(c, c->ws[wsi].w1r, (size_t)DHQR_NBV * ncmax));
      CHECK(ensure(c, c->ws[wsi].w2, (size_t)DHQR_NBV * ncmax));
    }
    CHECK(ensure(c, c->spart, 256 * NN));
    CHECK(ensure(c, c->sfull, NN));
  }
  auto width = [&](int64_t k) { return std::min<int64_t>(ZB, n - k * ZB); };
  auto mine = [&](int64_t k) { return (int)(k % P) == r; };
  auto has_right = [&](int64_t k) { return n - k * ZB - width(k) > 0; };
  // the owner factors panel k into buf[k & 1] on the lane, everybody takes part in its broadcast there
  auto produce = [&](int64_t k) -> int32_t {
    const int64_t c0 = k * ZB, w = width(k), rows = m - c0;
    double *b = buf[k & 1];
    on(sL, 1);
    // the readers of the broadcast that last LEFT this buffer (panel k - 2, if this rank was its root) must have copied it
    // out before anything -- this rank's next panel or a peer's incoming one -- overwrites it
    if (cm) CHECK(comm_wait_consumed(cm, tick[k & 1], sL));
    const size_t esz = 2 * sizeof(double);
    double *Pc = b + 2 * ZB + panel_tail_elems();  // the panel as it travels: rows x w complex, leading dimension rows
    const bool spread = cm && P > 1;
    if (mine(k)) {
      double *Pk = A + 2 * (c0 + (k / P) * ZB * lda);
      CHECK(zpanel_make(c, Pk, rows, w, lda, alpha + 2 * c0, operand(k), has_right(k)));
      HIPCHECK(hipMemcpyAsync(b, alpha + 2 * c0, (size_t)(2 * w) * sizeof(double), hipMemcpyDeviceToDevice, sL));
      if (spread && has_right(k))
        HIPCHECK(hipMemcpy2DAsync(Pc, rows * esz, Pk, lda * esz, rows * esz, w, hipMemcpyDeviceToDevice, sL));
    }
    if (spread) {
      const int64_t count = 2 * ZB + (has_right(k) ? panel_tail_elems() + 2 * rows * ZB : 0);
      CHECK(comm_bcast(cm, b, count, (int)(k % P), sL, &tick[k & 1]));
      if (!mine(k)) {
        HIPCHECK(hipMemcpyAsync(alpha + 2 * c0, b, (size_t)(2 * w) * sizeof(double), hipMemcpyDeviceToDevice, sL));
        if (has_right(k)) {  // the receiver's own embedding of the 64 reflectors (the owner's is made by zpanel_make)
          const PanelBuf pb = operand(k);
          const int64_t npad = panel_ldv(2 * rows);
          dim3 grid((unsigned)std::min<int64_t>((npad / 2 + 255) / 256, 64), (unsigned)ZB);
          hipLaunchKernelGGL(k_zpack_emb, grid, dim3(256), 0, sL, reinterpret_cast<const double2 *>(Pc), rows, rows, (int)w, pb.V,
                             pb.ldv, npad);
        }
      }
    }
    HIPCHECK(hipEventRecord(evP[k & 1], sL));
    return DHQR_OK;
  };
  auto body = [&]() -> int32_t {
    HIPCHECK(hipEventRecord(evS, sW));
    HIPCHECK(hipStreamWaitEvent(sL, evS, 0));
    bool produced = false;  // panel k already factored and broadcast by the look-ahead of step k - 1
    for (int64_t k = 0; k < K; ++k) {
      const int64_t c0 = k * ZB, rows = m - c0;
      on(sL, 1);
      if (k >= 1) HIPCHECK(hipStreamWaitEvent(sL, evW[(k - 1) & 1], 0));  // update k - 1 done; its buffer is free
      if (!produced) CHECK(produce(k));
      produced = false;
      if (!has_right(k)) break;  // the last panel is applied to nothing
      const PanelBuf pb = operand(k);
      // look-ahead: panels the pipelined kernel takes in one launch (<= 8192 rows); taller ones are one launch per
      // column and stay in line (a lane of single-column launches is slower than none, profiles/r02_ab_c64_blocked_lookahead.txt)
      const bool ahead = c->lookahead && c->zpipe && rows - ZB <= 8192 && k + 1 < K;
      if (ahead) {
        on(sL, 1);
        if (mine(k + 1)) {
          const int64_t lc = ((k + 1) / P) * ZB;
          CHECK(panel_apply(c, pb, 2 * rows, A + 2 * (c0 + lc * lda), width(k + 1), 2 * lda, 1));
        }
        CHECK(produce(k + 1));
        produced = true;
      }
      // caller's stream: panel k -> every column this rank holds beyond it (beyond panel k + 1 after a look-ahead)
      on(sW, 0);
      HIPCHECK(hipStreamWaitEvent(sW, evP[k & 1], 0));
      const int64_t lo = zcs_local_from(ahead ? k + 2 : k + 1, n, P, r);
      if (ncl - lo > 0) CHECK(panel_apply(c, pb, 2 * rows, A + 2 * (c0 + lo * lda), ncl - lo, 2 * lda, 1));
      HIPCHECK(hipEventRecord(evW[k & 1], sW));
    }
    on(sW, 0);
    HIPCHECK(hipEventRecord(evS, sL));
    HIPCHECK(hipStreamWaitEvent(sW, evS, 0));  // join: the caller's stream owns the result
    if (cm)
      for (int i = 0; i < 2; ++i) CHECK(comm_wait_consumed(cm, tick[i], sW));  // c->vt may be reused once this returns
    return DHQR_OK;
  };
  const int32_t rc = body();
  on(sW, 0);
  if (rc != DHQR_OK) {
    // a failure part-way leaves the lane behind the caller's stream: join it (the call is asynchronous on sW, so a caller
    // that synchronises sW must also be past everything the lane still reads or writes -- c->vt, A) and let the readers
    // of the last broadcasts finish before c->vt can be reused
    if (hipEventRecord(evS, sL) == hipSuccess) (void)hipStreamWaitEvent(sW, evS, 0);
    if (cm)
      for (int i = 0; i < 2; ++i) (void)comm_wait_consumed(cm, tick[i], sW);
    (void)hipStreamSynchronize(sL);
    (void)hipStreamSynchronize(sW);
  }
  CHECK(rc);
  LAUNCHCHECK();
  return DHQR_OK;
}

// (acc_hi, acc_lo)[r0:r1] -= R[r0:r1, lo:hi] x[lo:hi] in double-double (hi - lo <= ZBS_NB); A is addressed with GLOBAL column
// indices (the caller passes a base pointer shifted so that column lo of the block is the rank's local column)
__global__ __launch_bounds__(256) void k_zbacksub_update_rows_dd(const double2 *__restrict__ A, int64_t lda,
                                                                 const double2 *__restrict__ x, double2 *__restrict__ acc_hi,
                                                                 double2 *__restrict__ acc_lo, int64_t r0, int64_t r1,
                                                                 int64_t lo, int64_t hi) {
  __shared__ double2 xs[ZBS_NB];
  const int t = threadIdx.x;
  const int nb = (int)(hi - lo);
  if (t < nb) xs[t] = x[lo + t];
  __syncthreads();
  const int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + t;
  if (r >= r1) return;
  dhqr_dd br = {acc_hi[r].x, acc_lo[r].x}, bi = {acc_hi[r].y, acc_lo[r].y};
  for (int c = 0; c < nb; ++c) {  // src:248-250 / src:276-278
    const double2 a = A[r + (lo + c) * lda], xc = xs[c];
    dd_add_prod(br, -a.x, xc.x);
    dd_add_prod(br, a.y, xc.y);
    dd_add_prod(bi, -a.x, xc.y);
    dd_add_prod(bi, -a.y, xc.x);
  }
  dd_renorm(br);
  dd_renorm(bi);
  acc_hi[r] = zmake(br.hi, bi.hi);
  acc_lo[r] = zmake(br.lo, bi.lo);
}
// (bh, bl)[i] += sum over the P slots of (slot_hi, slot_lo)[i], i < w, exactly (double-double adds): the partial dots of
// the ranks, gathered by an all-reduce of a buffer in which every rank filled only its own slot.  slot p: 2 w complex
// (hi parts, then lo parts) at gath + p * 4 w doubles.  One workgroup of 64 threads.
__global__ __launch_bounds__(64) void k_zdd_accumulate(double2 *__restrict__ bh, double2 *__restrict__ bl,
                                                       const double2 *__restrict__ gath, int P, int w) {
  const int i = threadIdx.x;
  if (i >= w) return;
  dhqr_dd sr = {bh[i].x, bl[i].x}, si = {bh[i].y, bl[i].y};
  for (int p = 0; p < P; ++p) {
    const double2 h = gath[(int64_t)p * 2 * w + i], l = gath[(int64_t)p * 2 * w + w + i];
    sr = dd_add(sr, dhqr_dd{h.x, l.x});
    si = dd_add(si, dhqr_dd{h.y, l.y});
  }
  bh[i] = zmake(sr.hi, si.hi);
  bl[i] = zmake(sr.lo, si.lo);
}

// solve_householder!(b, H, alpha) (src:226-282) for ComplexF64 on the cyclic 64-column split.  db (m complex, the same on
// every rank) is overwritten; x = db[0:n] on every rank.  b is carried in DOUBLE-DOUBLE like the single-GPU solve
// (dhqr_complex.h: the reference's acceptance statistic, which test/runtests.jl:80-82 asserts for exactly this
// distributed call, sees the rounding of the O(mn) solve).  Q'b: the owner of panel k applies its 64 reflectors in column
// order and hands the updated tail of b -- high and low parts -- on with two broadcasts (the reference walks the owners
// sequentially with b in shared memory, src:226-230).  Back substitution: every rank accumulates -R[i, j] x[j] over ITS
// columns j into a double-double vector u; per panel one all-reduce GATHERS the ranks' 64 partial dots (every rank fills
// its own slot of a P-slot buffer, so the sum is exact) -- the reference's sum(fetch.(futures)), src:262-266 -- the
// owner adds them in double-double, solves the diagonal block (division by alpha, src:267) and broadcasts x.
// du: 3 m + 192 + 128 P complex of scratch.
static inline int64_t zcs_solve_work(int64_t m, int P) { return 3 * m + 192 + 128 * (int64_t)P; }
static int32_t zcs_solve(dhqr_ctx *c, dhqr_comm *cm_, const double *A_, int64_t m, int64_t n, int64_t lda, const double *alpha_,
                         double *db_, double *du_) {
  const int P = cm_ ? cm_->nranks : 1, r = cm_ ? cm_->rank : 0;
  dhqr_comm *cm = P > 1 ? cm_ : nullptr;
  const int64_t ZB = DHQR_ZNB, K = zcs_npanels(n);
  const double2 *A = reinterpret_cast<const double2 *>(A_), *al = reinterpret_cast<const double2 *>(alpha_);
  double2 *bh = reinterpret_cast<double2 *>(db_);
  double2 *uh = reinterpret_cast<double2 *>(du_), *ul = uh + (m + ZB), *bl = ul + (m + ZB), *gath = bl + (m + ZB);
  hipStream_t st = c->stream;
  auto sync_local = [&]() -> int32_t {  // LOCAL transport: peers read the root's buffer directly
    if (cm && cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(st));
      CHECK(comm_host_barrier(cm));
    }
    return DHQR_OK;
  };
  auto width = [&](int64_t k) { return std::min<int64_t>(ZB, n - k * ZB); };
  auto mine = [&](int64_t k) { return (int)(k % P) == r; };
  CHECK(prof_begin(c, CAT_SOLVE));
  HIPCHECK(hipMemsetAsync(du_, 0, (size_t)zcs_solve_work(m, P) * 2 * sizeof(double), st));  // u = 0, bl = 0
  for (int64_t k = 0; k < K; ++k) {  // b <- Q' b (src:232-242), panel by panel
    const int64_t c0 = k * ZB, w = width(k);
    if (mine(k)) {
      const double2 *Pk = A + (k / P) * ZB * lda;
      for (int64_t jj = 0; jj < w; ++jj) {
        const int64_t j = c0 + jj;
        if (m - j <= 2048)
          hipLaunchKernelGGL((k_zqtb_col_dd<256>), dim3(1), dim3(256), 0, st, Pk + jj * lda, bh, bl, m, j);
        else
          hipLaunchKernelGGL((k_zqtb_col_dd<1024>), dim3(1), dim3(1024), 0, st, Pk + jj * lda, bh, bl, m, j);
      }
    }
    if (cm) {
      CHECK(comm_bcast(cm, db_ + 2 * c0, 2 * (m - c0), (int)(k % P), st, nullptr));
      CHECK(sync_local());
      CHECK(comm_bcast(cm, reinterpret_cast<double *>(bl + c0), 2 * (m - c0), (int)(k % P), st, nullptr));
      CHECK(sync_local());
    }
  }
  for (int64_t k = K - 1; k >= 0; --k) {  // src:256-270
    const int64_t c0 = k * ZB, w = width(k);
    if (cm) {  // gather the ranks' partial dots of rows [c0, c0 + w): slot r = (hi parts | lo parts), zeros elsewhere
      HIPCHECK(hipMemsetAsync(gath, 0, (size_t)P * 2 * w * sizeof(double2), st));
      HIPCHECK(hipMemcpyAsync(gath + (int64_t)r * 2 * w, uh + c0, (size_t)w * sizeof(double2), hipMemcpyDeviceToDevice, st));
      HIPCHECK(hipMemcpyAsync(gath + (int64_t)r * 2 * w + w, ul + c0, (size_t)w * sizeof(double2), hipMemcpyDeviceToDevice, st));
      CHECK(comm_allreduce_sum(cm, reinterpret_cast<double *>(gath), (int64_t)P * 4 * w, st));
    }
    const double2 *base = A + ((k / P) * ZB - c0) * lda;  // global column j of R is read at base + j * lda
    if (mine(k)) {
      if (cm) {
        hipLaunchKernelGGL(k_zdd_accumulate, dim3(1), dim3(64), 0, st, bh + c0, bl + c0, (const double2 *)gath, P, (int)w);
      } else {  // one rank: its own partial dots, straight from u (slot layout: hi parts at uh, lo parts at ul)
        HIPCHECK(hipMemcpyAsync(gath, uh + c0, (size_t)w * sizeof(double2), hipMemcpyDeviceToDevice, st));
        HIPCHECK(hipMemcpyAsync(gath + w, ul + c0, (size_t)w * sizeof(double2), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_zdd_accumulate, dim3(1), dim3(64), 0, st, bh + c0, bl + c0, (const double2 *)gath, 1, (int)w);
      }
      for (int64_t hi = c0 + w; hi > c0; hi -= ZBS_NB) {
        const int64_t lo = std::max<int64_t>(c0, hi - ZBS_NB);
        hipLaunchKernelGGL(k_zbacksub_diag_dd, dim3(1), dim3(64), 0, st, base, lda, al, bh, bl, lo, hi);
        if (lo > c0)
          hipLaunchKernelGGL(k_zbacksub_update_rows_dd, dim3((unsigned)((lo - c0 + 255) / 256)), dim3(256), 0, st, base, lda,
                             (const double2 *)bh, bh, bl, c0, lo, lo, hi);
      }
    }
    if (cm) {
      CHECK(comm_bcast(cm, db_ + 2 * c0, 2 * w, (int)(k % P), st, nullptr));  // x (its low part is zero)
      CHECK(sync_local());
    }
    if (mine(k) && c0 > 0)
      for (int64_t hi = c0 + w; hi > c0; hi -= ZBS_NB) {
        const int64_t lo = std::max<int64_t>(c0, hi - ZBS_NB);
        hipLaunchKernelGGL(k_zbacksub_update_rows_dd, dim3((unsigned)((c0 + 255) / 256)), dim3(256), 0, st, base, lda,
                           (const double2 *)bh, uh, ul, (int64_t)0, c0, lo, hi);
      }
  }
  CHECK(prof_end(c));
  LAUNCHCHECK();
  return DHQR_OK;
}

// ---- the reference's DArray layout for ComplexF64 (src:115-120, test/runtests.jl:71): ONE contiguous column block per
// process (cs_contig_range) <-> the cyclic 64-column blocks above.  One broadcast of every rank's block on the way in, one
// per panel on the way out.  dBlk: m x w_r complex (leading dimension ldb complex); dStage: m x max(n/P + 1, 64) complex.
static int32_t zcs_convert(dhqr_ctx *c, dhqr_comm *cm_, double *A, int64_t m, int64_t n, int64_t lda, double *dBlk, int64_t ldb,
                           double *dStage, bool load) {
  const int P = cm_ ? cm_->nranks : 1, r = cm_ ? cm_->rank : 0;
  dhqr_comm *cm = P > 1 ? cm_ : nullptr;
  const int64_t ZB = DHQR_ZNB, K = zcs_npanels(n);
  const size_t esz = 2 * sizeof(double);
  auto stage_reusable = [&]() -> int32_t {  // LOCAL: peers read the root's stage directly
    if (cm && cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
    return DHQR_OK;
  };
  if (load) {
    for (int s = 0; s < P; ++s) {
      int64_t lo, hi;
      cs_contig_range(n, P, s, &lo, &hi);
      const int64_t wblk = hi - lo;
      if (wblk == 0) continue;
      if (s == r)
        HIPCHECK(hipMemcpy2DAsync(dStage, m * esz, dBlk, ldb * esz, m * esz, wblk, hipMemcpyDeviceToDevice, c->stream));
      if (cm) CHECK(comm_bcast(cm, dStage, 2 * m * wblk, s, c->stream, nullptr));
      for (int64_t k = lo / ZB; k <= (hi - 1) / ZB; ++k) {  // the pieces of [lo, hi) this rank holds, panel by panel
        if ((int)(k % P) != r) continue;
        const int64_t g0 = std::max<int64_t>(k * ZB, lo), g1 = std::min<int64_t>(std::min<int64_t>((k + 1) * ZB, hi), n);
        if (g1 <= g0) continue;
        HIPCHECK(hipMemcpy2DAsync(A + 2 * ((k / P) * ZB + g0 - k * ZB) * lda, lda * esz, dStage + 2 * (g0 - lo) * m, m * esz,
                                  m * esz, g1 - g0, hipMemcpyDeviceToDevice, c->stream));
      }
      CHECK(stage_reusable());
    }
    return DHQR_OK;
  }
  int64_t lo, hi;
  cs_contig_range(n, P, r, &lo, &hi);
  for (int64_t k = 0; k < K; ++k) {
    const int64_t w = std::min<int64_t>(ZB, n - k * ZB), g0 = k * ZB;
    if ((int)(k % P) == r)
      HIPCHECK(hipMemcpy2DAsync(dStage, m * esz, A + 2 * (k / P) * ZB * lda, lda * esz, m * esz, w, hipMemcpyDeviceToDevice, c->stream));
    if (cm) CHECK(comm_bcast(cm, dStage, 2 * m * w, (int)(k % P), c->stream, nullptr));
    const int64_t i0 = std::max<int64_t>(g0, lo), i1 = std::min<int64_t>(g0 + w, hi);
    if (i1 > i0)
      HIPCHECK(hipMemcpy2DAsync(dBlk + 2 * (i0 - lo) * ldb, ldb * esz, dStage + 2 * (i0 - g0) * m, m * esz, m * esz, i1 - i0,
                                hipMemcpyDeviceToDevice, c->stream));
    CHECK(stage_reusable());
  }
  return DHQR_OK;
}
