// dhqr_rowsplit.h -- tall-skinny factorisation with the ROWS split over the ranks (BASELINE configs[4]:
// 262144 x 4096 over 8 GPUs).  Included by dhqr_api.hip.  The reference cannot split rows
// (`@assert rowrange == 1:size(A,1)`, src:33); its algorithm needs one cross-rank reduction per COLUMN (norm,
// src:129, and the partial dots, src:208).  Here the R-first panel (dhqr_recon.h) needs cross-rank sums only per
// 128-column panel, all through the communicator's all-reduce ("RCCL all-reduce of the cross-partition partial
// dots"):
//     G = sum_r P_r' P_r   (128 x 128)        -> R = chol(G), replay of the top block on the rank that owns the
//                                                 panel's diagonal rows, broadcast of (-M^{-1}, alpha)
//     S = sum_r V_r' V_r   (128 x 128)        -> acceptance decision (identical on every rank: same S) and T
//     W = sum_r V_r' C_r   (128 x ncols)      -> trailing update C_r -= V_r (T' W), local
// Row slabs are 128-row aligned (rs_row_range), so a panel's diagonal block lives on exactly one rank -- whichever
// holds those rows.  Nothing waits on the host: panels are verified and committed on the device like in the
// column-split driver; a rejected panel (or a partial last panel) is redone column by column ACROSS the ranks
// (rs_panel_columns: the reference's algorithm with one small all-reduce + one small broadcast per column).
// Factor format = the reference's, distributed by rows: V rows live where the matrix rows live, R in the top n rows.
#pragma once

static inline void rs_row_range(int64_t m, int P, int r, int64_t *row0, int64_t *mloc) {
  const int64_t NB = DHQR_NBV, blocks = (m + NB - 1) / NB, q = blocks / P, rem = blocks % P;
  const int64_t b0 = (int64_t)r * q + std::min<int64_t>(r, rem), nb = q + (r < rem ? 1 : 0);
  *row0 = std::min<int64_t>(m, b0 * NB);
  *mloc = std::min<int64_t>(m, (b0 + nb) * NB) - *row0;
}

struct RsProblem {
  dhqr_ctx *c;
  dhqr_comm *cm;  // nullptr: single rank
  double *A;      // local rows [row0, row0 + mloc) of the m x n matrix, leading dimension lda
  int64_t m, n, lda, row0, mloc;
  double *alpha;  // n, replicated
  int P, r;
  int owner_of_row(int64_t grow) const {
    for (int q = 0; q < P; ++q) {
      int64_t r0, ml;
      rs_row_range(m, P, q, &r0, &ml);
      if (grow >= r0 && grow < r0 + ml) return q;
    }
    return P - 1;
  }
  // local rows taking part in the panel whose diagonal starts at global row c0: [off, off + rows)
  void active(int64_t c0, int64_t *off, int64_t *rows) const {
    const int64_t lo = std::max<int64_t>(0, c0 - row0);
    *off = std::min<int64_t>(lo, mloc);
    *rows = mloc - *off;
  }
};

// ---- column-by-column kernels of the robust path --------------------------------------------------
// partial[blk][k] = sum over this block's active rows of a_j[i] * a_{j+k}[i],  k in [0, nk)
__global__ __launch_bounds__(256) void k_rs_coldots(const double *__restrict__ Pp, int64_t ldp, int64_t rows,
                                                    int64_t lo, int j, int nk, double *__restrict__ partial) {
  __shared__ double red[8];
  const int t = threadIdx.x;
  double aj[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = lo + (int64_t)blockIdx.x * 1024 + t + e * 256;
    aj[e] = (i < rows) ? Pp[i + (int64_t)j * ldp] : 0.0;
  }
  for (int k = 0; k < nk; ++k) {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = lo + (int64_t)blockIdx.x * 1024 + t + e * 256;
      if (i < rows) s = fma(aj[e], Pp[i + (int64_t)(j + k) * ldp], s);
    }
    s = block_sum<256>(s, red);
    if (t == 0) partial[(int64_t)blockIdx.x * nk + k] = s;
  }
}
// owner of the diagonal row: bc = [alpha, f, s_1 .. s_{nk-1}] from the all-reduced dots d (d[0] = ||a_j||^2)
// and its diagonal row; s_k = v' a_{j+k} = f (d_k - alpha a_{j,j+k})   (src:129-131, 208)
__global__ __launch_bounds__(128) void k_rs_colscalars(const double *__restrict__ d, const double *__restrict__ diagrow,
                                                       int64_t ldp, int nk, double *__restrict__ bc,
                                                       double *__restrict__ alpha_j) {
  const int k = threadIdx.x;
  const double ajj = diagrow[0];
  const double s = sqrt(d[0]);
  const double al = s * dhqr_alphafactor(ajj);
  const double f = 1.0 / sqrt(s * (s + fabs(ajj)));
  if (k == 0) {
    bc[0] = al;
    bc[1] = f;
    *alpha_j = al;
  } else if (k < nk) {
    bc[1 + k] = f * (d[k] - al * diagrow[(int64_t)k * ldp]);
  }
}
// everybody: column j <- v, columns j+k (k >= 1) -= v s_k on the active rows (src:132-135, 209)
__global__ __launch_bounds__(256) void k_rs_colupdate(double *__restrict__ Pp, int64_t ldp, int64_t rows, int64_t lo,
                                                      int64_t diag, int j, int nk, const double *__restrict__ bc) {
  __shared__ double sk[RC_N + 2];
  const int t = threadIdx.x;
  for (int k = t; k < nk + 1; k += 256) sk[k] = bc[k];
  __syncthreads();
  const double al = sk[0], f = sk[1];
  for (int e = 0; e < 4; ++e) {
    const int64_t i = lo + (int64_t)blockIdx.x * 1024 + t + e * 256;
    if (i >= rows) continue;
    const double a = Pp[i + (int64_t)j * ldp];
    const double v = (i == diag) ? (a - al) * f : a * f;
    Pp[i + (int64_t)j * ldp] = v;
    for (int k = 1; k < nk; ++k) Pp[i + (int64_t)(j + k) * ldp] = fma(-v, sk[1 + k], Pp[i + (int64_t)(j + k) * ldp]);
  }
}
__global__ void k_rs_set_breakdown(const double *__restrict__ word, int *__restrict__ stat) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *word != 0.0) stat[1] = 1;
}
__global__ void k_rs_get_breakdown(int *__restrict__ stat, double *__restrict__ word) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *word = stat[1] ? 1.0 : 0.0;
    stat[1] = 0;
  }
}

// ---- state of the grouped / look-ahead driver ------------------------------------------------------------------
// Ring slot = operands of one group (a PAIR of panels applied in one K = 256 pass, or a single panel):
//   [ V : ldv x 256 = [V_a | V_b] | T_a | T_a' | T_b | T_b' | S2 = [V_b'V_a | V_b'V_b] | bc_a | bc_b ]
// V_b sits `shift` rows below V_a (zeros above): 128 on the rank that holds the pair's diagonal blocks, 0 on a rank
// whose rows all lie below them.  bc_x = [-M^{-1} (128 x 128) | alpha (128) | breakdown word ...] is what the owner of
// the diagonal rows broadcasts per panel.
#define RS_RING 2
#define RS_BC ((size_t)DHQR_NBV * DHQR_NBV + 256)
struct RsState {
  bool init = false;
  hipEvent_t ev_group[2 * RS_RING], ev_wide[2 * RS_RING], ev_start = nullptr, ev_end = nullptr;
  Buf ring[RS_RING];
  int64_t ticket[RS_RING][2];  // LOCAL transport: the broadcast this rank rooted out of bc_x (its readers must be done)
};
struct RsSlot {
  double *V, *T[2], *Tt[2], *S2, *bc[2];
};
static inline size_t rs_slot_elems(int64_t ldv) {
  return (size_t)ldv * 2 * DHQR_NBV + 6 * (size_t)DHQR_NBV * DHQR_NBV + 2 * RS_BC;
}
static inline RsSlot rs_slot_view(double *base, int64_t ldv) {
  const size_t NN = (size_t)DHQR_NBV * DHQR_NBV;
  RsSlot s;
  s.V = base;
  double *t = base + (size_t)ldv * 2 * DHQR_NBV;
  s.T[0] = t;
  s.Tt[0] = t + NN;
  s.T[1] = t + 2 * NN;
  s.Tt[1] = t + 3 * NN;
  s.S2 = t + 4 * NN;
  s.bc[0] = t + 6 * NN;
  s.bc[1] = t + 6 * NN + RS_BC;
  return s;
}
static int32_t rs_state_init(dhqr_ctx *c) {
  if (!c->rs) c->rs = new RsState();
  RsState &s = *c->rs;
  if (s.init) return DHQR_OK;
  for (int i = 0; i < 2 * RS_RING; ++i) {
    HIPCHECK(hipEventCreateWithFlags(&s.ev_group[i], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&s.ev_wide[i], hipEventDisableTiming));
  }
  HIPCHECK(hipEventCreateWithFlags(&s.ev_start, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&s.ev_end, hipEventDisableTiming));
  for (int i = 0; i < RS_RING; ++i) s.ticket[i][0] = s.ticket[i][1] = -1;
  s.init = true;
  return DHQR_OK;
}
static void rs_state_free(dhqr_ctx *c) {
  if (!c->rs) return;
  RsState &s = *c->rs;
  if (s.init) {
    for (int i = 0; i < 2 * RS_RING; ++i) {
      (void)hipEventDestroy(s.ev_group[i]);
      (void)hipEventDestroy(s.ev_wide[i]);
    }
    (void)hipEventDestroy(s.ev_start);
    (void)hipEventDestroy(s.ev_end);
  }
  for (Buf &b : s.ring)
    if (b.p) (void)hipFree(b.p);
  delete c->rs;
  c->rs = nullptr;
}

struct RsWork {  // device workspaces of one rank (views into ctx buffers)
  double *Vw;    // ldv x 128: V operand of a re-applied panel (residual, solve, resume) = ring slot 0
  int64_t ldv;
  double *G, *S, *Rref, *T, *Tt, *bc, *altmp, *part;
  RsSlot slot[RS_RING];
};

static int32_t rs_prepare(const RsProblem &pr, RsWork *w) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV;
  const size_t NN = (size_t)NB * NB;
  CHECK(rs_state_init(c));
  w->ldv = panel_ldv(std::max<int64_t>(pr.mloc, 1));
  for (int i = 0; i < RS_RING; ++i) CHECK(ensure(c, c->rs->ring[i], rs_slot_elems(w->ldv)));
  CHECK(ensure(c, c->rbuf, 8 * NN + 4096));
  // split-K partials / reduced Y / W of the two streams (nothing is reallocated while they run): [0] wide, [1] lane
  const size_t ncmax = (size_t)std::max<int64_t>(pr.n, 2 * NB), ntmax = (ncmax + 127) / 128;
  for (int s = 0; s < 2; ++s) {
    CHECK(ensure(c, c->ws[s].w1, s == 0 ? NN * std::max<size_t>(6144 + 2 * ntmax + 128, 640 + 32 * ntmax) : NN * 2200));
    CHECK(ensure(c, c->ws[s].w1r, 2 * NB * ncmax));
    CHECK(ensure(c, c->ws[s].w2, 2 * NB * ncmax));
  }
  CHECK(ensure(c, c->spart, (size_t)512 * NN));
  CHECK(ensure(c, c->sfull, NN));
  CHECK(ensure(c, c->scratch, 4096));
  const size_t nblk = (size_t)((pr.mloc + 1023) / 1024 + 1);
  CHECK(ensure(c, c->pbuf, nblk * NB + 1024));
  for (int i = 0; i < RS_RING; ++i) w->slot[i] = rs_slot_view(c->rs->ring[i].p, w->ldv);
  w->Vw = w->slot[0].V;
  double *r = c->rbuf.p;
  w->G = r;
  w->S = r + NN;
  w->Rref = r + 2 * NN;
  w->T = r + 3 * NN;
  w->Tt = r + 4 * NN;
  w->bc = r + 5 * NN;        // [-M^{-1} (NN) | alpha (128) | breakdown word | ...]
  w->altmp = r + 5 * NN + NN;
  w->part = c->pbuf.p;
  return DHQR_OK;
}

// Y (128 x ncols, ld 128, in the current workspace's w1r) = V' C over the local active rows, summed over the ranks of cmx
static int32_t rs_vtc_allreduce(const RsProblem &pr, dhqr_comm *cmx, const double *V, int64_t ldv, const double *C,
                                int64_t ldc, int64_t rows, int64_t ncols) {
  dhqr_ctx *c = pr.c;
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  const int64_t NB = DHQR_NBV, wstride = NB * ncols;
  CHECK(ensure(c, ws.w1r, (size_t)wstride));
  if (rows <= 0) {
    HIPCHECK(hipMemsetAsync(ws.w1r.p, 0, (size_t)wstride * sizeof(double), c->stream));
  } else {
    const int64_t ntiles = (ncols + 127) / 128;
    int64_t nsplit, rps;
    pick_split(rows, ntiles, 512, ntiles <= 2 ? 256 : 64, &nsplit, &rps, 512, ntiles <= 2 ? 64 : 128);
    CHECK(ensure(c, ws.w1, (size_t)nsplit * NB * (size_t)ncols));
    const bool vec = (ldc % 2 == 0) && (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(C) && aligned16(V);
    const dim3 gtn((unsigned)ntiles, (unsigned)nsplit);
    if (vec)
      hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), gtn, dim3(256), 0, c->stream, V, ldv, C, ldc, 1, (int64_t)0, rows, ncols, rps,
                         ws.w1.p, NB, wstride);
    else
      hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), gtn, dim3(256), 0, c->stream, V, ldv, C, ldc, 1, (int64_t)0, rows, ncols, rps,
                         ws.w1.p, NB, wstride);
    hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)((wstride + 63) / 64)), dim3(256), 0, c->stream,
                       (const double *)ws.w1.p, (int)nsplit, wstride, wstride, ws.w1r.p);
  }
  LAUNCHCHECK();
  if (cmx) CHECK(comm_allreduce_sum(cmx, ws.w1r.p, wstride, c->stream));
  return DHQR_OK;
}
// C (rows x ncols) -= V (op(T)' Y): W = Top' Y, then the NN GEMM (predicated when `pred`)
// TopT: the transpose of Top when the caller has it (narrow updates then use k_tw_fused, dhqr_gemm.h), else nullptr.
static int32_t rs_apply_w(const RsProblem &pr, const double *V, int64_t ldv, const double *Top, double *C, int64_t ldc,
                          int64_t rows, int64_t ncols, bool pred, const double *TopT = nullptr) {
  dhqr_ctx *c = pr.c;
  dhqr_ctx::WS &ws = c->ws[c->cur_ws];
  const int64_t NB = DHQR_NBV, ntiles = (ncols + 127) / 128;
  CHECK(ensure(c, ws.w2, (size_t)NB * (size_t)ncols));
  if (TopT && ntiles <= 2)
    hipLaunchKernelGGL((k_tw_fused<false>), dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, c->stream, (const double *)ws.w1r.p, ncols,
                       TopT, (const double *)nullptr, (const double *)nullptr, ws.w2.p, NB);
  else
  hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3((unsigned)ntiles, 1), dim3(256), 0, c->stream, Top, NB,
                     (const double *)ws.w1r.p, NB, 1, (int64_t)0, NB, ncols, NB, ws.w2.p, NB, (int64_t)0);
  if (rows > 0) {
    const bool vec = (ldc % 2 == 0) && (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(C) && aligned16(V);
    const int64_t gx = (rows + 127) / 128;
    const int swz = (gx >= 16 && ntiles >= 16) ? 1 : 0;
    dim3 grid((unsigned)gx, (unsigned)ntiles);
    if (swz) grid = dim3((unsigned)((((gx + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
    launch_nn_sub<128>(c, vec, grid, V, ldv, (const double *)ws.w2.p, NB, C, ldc, rows, ncols, swz, pred);
  }
  LAUNCHCHECK();
  return DHQR_OK;
}
// out (128 x 128) = sum over the ranks of cmx of X' X
static int32_t rs_gram_allreduce(const RsProblem &pr, dhqr_comm *cmx, const double *X, int64_t ldx, int64_t rows, double *out) {
  dhqr_ctx *c = pr.c;
  const size_t NN = (size_t)DHQR_NBV * DHQR_NBV;
  if (rows <= 0) HIPCHECK(hipMemsetAsync(out, 0, NN * sizeof(double), c->stream));
  else CHECK(gram128(c, X, ldx, rows, out));
  LAUNCHCHECK();
  if (cmx) CHECK(comm_allreduce_sum(cmx, out, (int64_t)NN, c->stream));
  return DHQR_OK;
}
// out2 (128 x 256, ld 128) = sum over the ranks of Vb' [Va | Vb]: the pair's cross term V_b'V_a and S_b = V_b'V_b from ONE
// GEMM and ONE all-reduce.  VaVb = the pair operand at V_b's first row (ld ldv, V_b 128 columns to the right of V_a).
static int32_t rs_gram_cross_allreduce(const RsProblem &pr, dhqr_comm *cmx, const double *VaVb, int64_t ldv, int64_t rows,
                                       double *out2) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV;
  const size_t NN = (size_t)NB * NB;
  if (rows <= 0) {
    HIPCHECK(hipMemsetAsync(out2, 0, 2 * NN * sizeof(double), c->stream));
  } else {
    int64_t nsplit, rps;
    pick_split(rows, 2, 512, 256, &nsplit, &rps, 512, 64);
    CHECK(ensure(c, c->spart, (size_t)nsplit * 2 * NN));
    const double *Vb = VaVb + NB * ldv;
    const bool vec = (ldv % 2 == 0) && (rows % 2 == 0) && aligned16(VaVb);
    if (vec)
      hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3(2, (unsigned)nsplit), dim3(256), 0, c->stream, Vb, ldv, VaVb, ldv, 1,
                         (int64_t)0, rows, 2 * NB, rps, c->spart.p, NB, (int64_t)(2 * NN));
    else
      hipLaunchKernelGGL((k_gemm_tn<1, 1, 128>), dim3(2, (unsigned)nsplit), dim3(256), 0, c->stream, Vb, ldv, VaVb, ldv, 1,
                         (int64_t)0, rows, 2 * NB, rps, c->spart.p, NB, (int64_t)(2 * NN));
    hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)(2 * NN / 64)), dim3(256), 0, c->stream, (const double *)c->spart.p,
                       (int)nsplit, (int64_t)(2 * NN), (int64_t)(2 * NN), out2);
  }
  LAUNCHCHECK();
  if (cmx) CHECK(comm_allreduce_sum(cmx, out2, (int64_t)(2 * NN), c->stream));
  return DHQR_OK;
}
// Vdst <- the V operand of an already factored panel (diag owner: R part zeroed; others: plain copy of their rows)
static int32_t rs_pack(const RsProblem &pr, double *Vdst, int64_t ldv, int64_t c0, int64_t wcols, int64_t off, int64_t rows,
                       bool diag_owner) {
  dhqr_ctx *c = pr.c;
  if (rows <= 0) return DHQR_OK;
  const double *P = pr.A + off + c0 * pr.lda;
  const int64_t npad = panel_ldv(rows);
  dim3 grid((unsigned)std::min<int64_t>((npad + 255) / 256, 64), DHQR_NBV);
  if (diag_owner) {
    hipLaunchKernelGGL(k_pack_v, grid, dim3(256), 0, c->stream, P, pr.lda, rows, wcols, Vdst, ldv, npad);
  } else {
    // all rows are below the diagonal: pack with a "diagonal" far above (rows >= p always) = copy + zero padding columns
    hipLaunchKernelGGL(k_pack_rows, grid, dim3(256), 0, c->stream, P, pr.lda, rows, wcols, Vdst, ldv, npad);
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

// Robust panel: the reference's algorithm column by column across the ranks (any width <= 128).  Leaves the operands
// of the trailing update in Vdst (ld w.ldv), T, Tt.
static int32_t rs_panel_columns(const RsProblem &pr, const RsWork &w, dhqr_comm *cmx, int64_t c0, int64_t wcols, double *Vdst,
                                double *T, double *Tt) {
  dhqr_ctx *c = pr.c;
  dhqr_comm *cm = (cmx && cmx->nranks > 1) ? cmx : nullptr;
  int64_t off, rows;
  pr.active(c0, &off, &rows);
  const int downer = pr.owner_of_row(c0);
  const bool diag_owner = downer == pr.r;
  double *Pp = pr.A + off + c0 * pr.lda;  // local active rows of the panel; on the diag owner row 0 is global row c0
  double *d = w.altmp + 256, *bcs = w.altmp + 512;  // dots (<= 128), scalars (<= 130)
  for (int64_t j = 0; j < wcols; ++j) {
    const int nk = (int)(wcols - j);
    // active rows of column j: global row >= c0 + j
    const int64_t lo = diag_owner ? j : 0;
    const int64_t nrow = std::max<int64_t>(rows - lo, 0);
    const unsigned nblk = (unsigned)std::max<int64_t>((nrow + 1023) / 1024, 1);
    if (nrow > 0) {
      hipLaunchKernelGGL(k_rs_coldots, dim3(nblk), dim3(256), 0, c->stream, (const double *)Pp, pr.lda, rows, lo, (int)j, nk,
                         w.part);
      hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)((nk + 63) / 64)), dim3(256), 0, c->stream, (const double *)w.part,
                         (int)nblk, (int64_t)nk, (int64_t)nk, d);
    } else {
      HIPCHECK(hipMemsetAsync(d, 0, (size_t)nk * sizeof(double), c->stream));
    }
    if (cm) CHECK(comm_allreduce_sum(cm, d, nk, c->stream));
    if (diag_owner)
      hipLaunchKernelGGL(k_rs_colscalars, dim3(1), dim3(128), 0, c->stream, (const double *)d,
                         (const double *)(Pp + j + j * pr.lda), pr.lda, nk, bcs, pr.alpha + c0 + j);
    if (cm) CHECK(comm_bcast(cm, bcs, nk + 1, downer, c->stream, nullptr));
    if (!diag_owner)
      hipLaunchKernelGGL(k_commit_alpha, dim3(1), dim3(DHQR_NBV), 0, c->stream, (const double *)bcs, 1, pr.alpha + c0 + j,
                         (double *)nullptr, (const int *)nullptr, 0);
    if (nrow > 0)
      hipLaunchKernelGGL(k_rs_colupdate, dim3(nblk), dim3(256), 0, c->stream, Pp, pr.lda, rows, lo, diag_owner ? j : (int64_t)-1,
                         (int)j, nk, (const double *)bcs);
    LAUNCHCHECK();
    if (cm && cm->kind == COMM_LOCAL) {  // the tiny broadcast source is reused by the next column
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
  }
  // operands of the trailing update: V packed, S all-reduced, T
  if (rows > 0) CHECK(rs_pack(pr, Vdst, w.ldv, c0, wcols, off, rows, diag_owner));
  CHECK(rs_gram_allreduce(pr, cmx, Vdst, w.ldv, rows, w.S));
  launch_build_t(c, w.S, (int)wcols, T, Tt);
  LAUNCHCHECK();
  return DHQR_OK;
}

// TSQR-HR of the row-split panel (dhqr_tsqr.h): local tree up, gather of the P local R factors through one all-reduce
// of a zero-padded buffer, the same tree over them on every rank (up: R_t, down: this rank's block C_r), local tree
// down from C_r.  Rt <- the tree's R (128 x 128, identical on every rank); *Q / *ldq <- the local rows of the explicit
// orthonormal factor (inside c->tsq).
static int32_t rs_tsqr_hr(const RsProblem &pr, dhqr_comm *cmx, const double *P, int64_t rows, double *Rt, const double **Q,
                          int64_t *ldq) {
  dhqr_ctx *c = pr.c;
  const size_t NN = TSQR_NN;
  const bool multi = cmx && cmx->nranks > 1;
  const size_t Pn = multi ? (size_t)pr.P : 0;
  const size_t top = Pn ? (5 * Pn + 4) * NN + (size_t)TsqrLevels::node_blocks((int64_t)Pn) * 2 * NN : 0;
  CHECK(ensure(c, c->tsq, top + TsqrLocal::elems(rows)));
  TsqrLocal t;
  t.place(c->tsq.p + top, rows);
  *Q = t.Yl;
  *ldq = t.ldy;
  if (!multi) {
    CHECK(tsqr_local_up(c, t, P, pr.lda, Rt, true));
    return tsqr_local_down(c, t, nullptr);
  }
  double *gather = c->tsq.p, *ping = gather + Pn * NN, *pong = ping + (Pn + 1) * NN, *cping = pong + (Pn + 1) * NN,
         *cpong = cping + (Pn + 1) * NN, *Ytop = cpong + (Pn + 1) * NN;
  HIPCHECK(hipMemsetAsync(gather, 0, Pn * NN * sizeof(double), c->stream));
  CHECK(tsqr_local_up(c, t, P, pr.lda, gather + (size_t)pr.r * NN, true));
  CHECK(comm_allreduce_sum(cmx, gather, (int64_t)(Pn * NN), c->stream));
  CHECK(tsqr_pairs_up(c, gather, pr.P, Rt, Ytop, ping, pong));
  const double *Cblocks = nullptr;
  CHECK(tsqr_pairs_down(c, pr.P, Ytop, nullptr, cping, cpong, &Cblocks));
  return tsqr_local_down(c, t, Cblocks + (size_t)pr.r * NN);
}

// R-first factorisation of the full-width panel k, enqueued on c->stream over the channel cmx; nothing is written to the
// matrix, alpha or T unless the panel is accepted on the device (k_build_t: the same decision on every rank from the
// same all-reduced S).  Vdst (ld w.ldv) <- this rank's rows of V, T / Tt <- the compact-WY factor, bcbuf = the panel's
// broadcast block.  cross != nullptr (second panel of a pair): cross = the pair operand at this panel's first local
// row, S2 <- [V_b'V_a | V_b'V_b] from one GEMM + one all-reduce.  tsqr: R from the TSQR tree instead of Gram/Cholesky.
static int32_t rs_panel_fast(const RsProblem &pr, const RsWork &w, dhqr_comm *cmx, int64_t k, bool tsqr, double *Vdst, double *T,
                             double *Tt, double *bcbuf, int64_t *ticket, const double *cross, double *S2) {
  dhqr_ctx *c = pr.c;
  dhqr_comm *cm = (cmx && cmx->nranks > 1) ? cmx : nullptr;
  const int64_t NB = DHQR_NBV, c0 = k * NB;
  const size_t NN = (size_t)NB * NB;
  int64_t off, rows;
  pr.active(c0, &off, &rows);
  const int downer = pr.owner_of_row(c0);
  const bool diag_owner = downer == pr.r;
  double *P = pr.A + off + c0 * pr.lda;
  const double *X = P;  // what the reflectors are reconstructed from: the panel rows, or the rows of its Q factor
  int64_t ldx = pr.lda;
  const double *alpha_commit = bcbuf + NN;
  // LOCAL transport: the readers of the broadcast this rank last rooted out of bcbuf must be done before it changes
  if (cm && ticket) CHECK(comm_wait_consumed(cm, *ticket, c->stream));
  if (tsqr) {
    CHECK(rs_tsqr_hr(pr, cmx, P, rows, w.G, &X, &ldx));                       // w.G = R_t, X = local rows of Q
    if (diag_owner) {
      hipLaunchKernelGGL(k_tsqr_identity, dim3((unsigned)(NN / 256)), dim3(256), 0, c->stream, w.S);
      launch_recon_top(c, X, ldx, w.S, bcbuf + NN, w.Rref, bcbuf);            // R(Q) = I: alpha(Q) = +-1, -M^{-1}
      hipLaunchKernelGGL(k_tsqr_sign_cols, dim3((unsigned)(NN / 256)), dim3(256), 0, c->stream, bcbuf, (const double *)w.G);
      hipLaunchKernelGGL(k_rs_get_breakdown, dim3(1), dim3(64), 0, c->stream, c->dstat, bcbuf + NN + NB);
    }
  } else {
    CHECK(rs_gram_allreduce(pr, cmx, P, pr.lda, rows, w.G));                  // G = sum P_r' P_r
    if (diag_owner) {
      hipLaunchKernelGGL((k_panel_top<false>), dim3(1), dim3(1024), 0, c->stream, (const double *)w.G, (const double *)P, pr.lda,
                         bcbuf + NN, w.Rref, bcbuf, c->dstat + 1);            // alpha -> bc tail, -M^{-1} -> bc
      hipLaunchKernelGGL(k_rs_get_breakdown, dim3(1), dim3(64), 0, c->stream, c->dstat, bcbuf + NN + NB);
    }
  }
  if (cm) CHECK(comm_bcast(cm, bcbuf, (int64_t)NN + NB + 8, downer, c->stream, ticket));
  hipLaunchKernelGGL(k_rs_set_breakdown, dim3(1), dim3(64), 0, c->stream, (const double *)(bcbuf + NN + NB), c->dstat);
  if (tsqr) {  // R = D R_t, alpha = diag(R) with D = alpha(Q) from the broadcast (every rank: same values)
    hipLaunchKernelGGL(k_tsqr_final_r, dim3(NN / 256), dim3(256), 0, c->stream, (const double *)w.G,
                       (const double *)(bcbuf + NN), w.Rref, w.altmp + 1024);
    alpha_commit = w.altmp + 1024;
  }
  if (rows > 0) CHECK(mul128(c, X, ldx, rows, bcbuf, Vdst, w.ldv));           // V = X M^{-1}
  if (diag_owner)
    hipLaunchKernelGGL(k_recon_fix, dim3(NN / 256), dim3(256), 0, c->stream, Vdst, w.ldv, (const double *)(bcbuf + NN),
                       (const double *)bcbuf);
  const double *S = w.S;
  if (cross) {
    CHECK(rs_gram_cross_allreduce(pr, cmx, cross, w.ldv, rows, S2));          // [V_b'V_a | V_b'V_b]
    S = S2 + NN;
  } else {
    CHECK(rs_gram_allreduce(pr, cmx, Vdst, w.ldv, rows, w.S));                // S = sum V_r' V_r: same on every rank
  }
  hipLaunchKernelGGL(k_build_t, dim3(1), dim3(1024), 0, c->stream, S, (int)NB, T, Tt, c->recon_tol, c->dstat, (int)k,
                     (double *)nullptr, (double *)nullptr);                                      // same decision on every rank
  if (rows > 0) {
    dim3 grid((unsigned)std::min<int64_t>((rows + 255) / 256, 64), DHQR_NBV);
    if (diag_owner) {  // reflectors, R and alpha in one launch
      hipLaunchKernelGGL(k_commit_panel, grid, dim3(256), 0, c->stream, P, pr.lda, rows, (const double *)Vdst, w.ldv,
                         (const double *)w.Rref, alpha_commit, pr.alpha + c0, (double *)nullptr, (const int *)c->dstat, (int)k);
    } else {
      hipLaunchKernelGGL(k_unpack_rows, grid, dim3(256), 0, c->stream, P, pr.lda, rows, NB, (const double *)Vdst, w.ldv,
                         (const int *)c->dstat, (int)k);
    }
  }
  if (!(diag_owner && rows > 0))
    hipLaunchKernelGGL(k_commit_alpha, dim3(1), dim3(DHQR_NBV), 0, c->stream, alpha_commit, (int)NB, pr.alpha + c0,
                       (double *)nullptr, (const int *)c->dstat, (int)k);
  LAUNCHCHECK();
  return DHQR_OK;
}

struct RsGroup {
  int64_t a;    // first panel
  int np;       // 1 or 2 panels
  bool fast;    // every panel of the group goes through the asynchronous, device-verified path
  int64_t last() const { return a + np - 1; }
};

// Group g (operands in `sl`) applied to this rank's active rows of the columns [col0, col0 + ncols) on c->stream, the
// partial dots summed over the channel cmx.  A rank without active rows still joins the all-reduce (with zeros).
static int32_t rs_group_apply(const RsProblem &pr, const RsWork &w, const RsSlot &sl, const RsGroup &gr, int64_t col0,
                              int64_t ncols, dhqr_comm *cmx) {
  dhqr_ctx *c = pr.c;
  if (ncols <= 0) return DHQR_OK;
  const int64_t NB = DHQR_NBV;
  dhqr_comm *cm = cmx;  // also at one rank: the all-reduces of V'C are then counted as issued (dhqr_comm_counters) and move nothing
  int64_t off, rows;
  pr.active(gr.a * NB, &off, &rows);
  double *C = pr.A + off + col0 * pr.lda;
  c->epoch = gr.fast ? (int)gr.last() : -1;
  if (gr.np == 2) {
    int64_t offb, rowsb;
    pr.active((gr.a + 1) * NB, &offb, &rowsb);
    if (rows <= 0) {
      dhqr_ctx::WS &ws = c->ws[c->cur_ws];
      if (cm) {
        HIPCHECK(hipMemsetAsync(ws.w1r.p, 0, (size_t)(2 * NB * ncols) * sizeof(double), c->stream));
        CHECK(comm_allreduce_sum(cm, ws.w1r.p, 2 * NB * ncols, c->stream));
      }
      return DHQR_OK;
    }
    return pair_apply(c, sl.V, w.ldv, rows, sl.T[0], sl.T[1], sl.S2, C, ncols, pr.lda, cm, rowsb);
  }
  CHECK(prof_begin(c, CAT_VTA));
  CHECK(rs_vtc_allreduce(pr, cmx, sl.V, w.ldv, C, pr.lda, rows, ncols));
  CHECK(prof_end(c));
  CHECK(prof_begin(c, CAT_AVW));
  CHECK(rs_apply_w(pr, sl.V, w.ldv, sl.T[0], C, pr.lda, rows, ncols, true, sl.Tt[0]));
  CHECK(prof_end(c));
  if (c->profiling) {
    c->st.flops_gemm_vta += 2.0 * NB * (double)rows * (double)ncols;
    c->st.flops_gemm_avw += 2.0 * NB * (double)rows * (double)ncols;
  }
  return DHQR_OK;
}

// One asynchronous pass over the panels [kstart, K).  level: how the FIRST panel of the pass is factored -- 0 like the
// others, 1 R from the TSQR tree (a panel the Gram/Cholesky path was rejected on), 2 column by column.
//
// Panels are grouped like in the column-split driver (dhqr_dist.h): a group is a PAIR of full-width panels applied in
// one K = 256 pass, or a single panel.  Two streams, each with its own communicator channel (the collectives of one
// channel are ordered; the lane's small latency-bound all-reduces must not queue behind the wide stream's):
//   lane (high priority, cm->lane)  group g+1: the previous group applied to ITS columns only, the panels factored
//                                   (Gram all-reduce -> top block on the diagonal owner -> broadcast -> V -> S
//                                   all-reduce -> T), panel a applied to panel b's columns, the pair's cross term;
//   wide (caller's stream, cm)      group g applied to the columns beyond group g+1 (one all-reduce of the 256 x ncols
//                                   partial dots per pair), overlapping the lane's chain of small collectives.
// c->lookahead == false: both roles on the caller's stream and one channel, in the same order.
// pair_b <- the second panels of the pairs formed (what rs_factor needs to resume after a rejected panel).
static int32_t rs_run(const RsProblem &pr, const RsWork &w, int64_t kstart, int level, int *failed, int64_t *nfast,
                      std::vector<int64_t> &pair_b) {
  dhqr_ctx *c = pr.c;
  RsState &S = *c->rs;
  const int64_t NB = DHQR_NBV, n = pr.n, K = (n + NB - 1) / NB;
  auto is_fast = [&](int64_t k) {
    const int64_t c0 = k * NB;
    return c->panel_impl == 3 && std::min<int64_t>(NB, n - c0) == NB && pr.m - c0 >= 2 * NB && !(level == 2 && k == kstart);
  };
  auto is_tsqr = [&](int64_t k) { return c->cholqr_passes == 3 || (level == 1 && k == kstart); };
  std::vector<RsGroup> groups;
  for (int64_t k = kstart; k < K;) {
    RsGroup g;
    g.a = k;
    g.fast = is_fast(k);
    g.np = (c->pair && g.fast && !is_tsqr(k) && k + 1 < K && is_fast(k + 1) && !is_tsqr(k + 1)) ? 2 : 1;
    if (g.np == 2) pair_b.push_back(k + 1);
    groups.push_back(g);
    k += g.np;
  }
  const int G = (int)groups.size();
  const bool la = c->lookahead && G >= 2;
  dhqr_comm *cmW = pr.cm, *cmL = (la && pr.cm && pr.cm->lane) ? pr.cm->lane : pr.cm;
  hipStream_t sW = c->stream, sL = la ? c->hi : c->stream;
  auto on = [&](hipStream_t s, int wsi) { c->stream = s; c->cur_ws = wsi; };
  const int saved_epoch = c->epoch;
  bool stop = false;

  // lane: bring group h's columns up to date and factor its panels
  auto produce = [&](int h) -> int32_t {
    const RsGroup &gr = groups[h];
    const RsSlot &sl = w.slot[h % RS_RING];
    on(sL, la ? 1 : 0);
    // the slot's previous readers (wide update of group h-2) are done, and the columns of group h carry every group
    // before h-1, once the wide update of group h-2 has finished
    if (la && h >= 2) HIPCHECK(hipStreamWaitEvent(sL, S.ev_wide[(h - 2) % (2 * RS_RING)], 0));
    bool was = c->profiling;
    CHECK(prof_begin(c, CAT_PANEL));
    c->profiling = false;
    auto body = [&]() -> int32_t {
      int64_t ncols_g = 0;
      for (int idx = 0; idx < gr.np; ++idx) ncols_g += std::min<int64_t>(NB, n - (gr.a + idx) * NB);
      if (la && h >= 1) CHECK(rs_group_apply(pr, w, w.slot[(h - 1) % RS_RING], groups[h - 1], gr.a * NB, ncols_g, cmL));
      int64_t offa, rowsa;
      pr.active(gr.a * NB, &offa, &rowsa);
      for (int idx = 0; idx < gr.np; ++idx) {
        const int64_t k = gr.a + idx, c0 = k * NB, wcols = std::min<int64_t>(NB, n - c0);
        int64_t off, rows;
        pr.active(c0, &off, &rows);
        const int64_t shift = off - offa;  // 0 or 128 (<= rows of panel a)
        double *Vdst = sl.V + (size_t)idx * NB * w.ldv + shift;
        c->epoch = -1;
        if (gr.fast) {
          if (idx == 1) {  // panel a -> the columns of panel b, then the rows of V_b above its first row
            RsGroup ga = gr;
            ga.np = 1;
            CHECK(rs_group_apply(pr, w, sl, ga, c0, wcols, cmL));
            c->epoch = -1;
            if (shift > 0)
              hipLaunchKernelGGL(k_zero_rows, dim3(DHQR_NBV), dim3(128), 0, c->stream, sl.V + (size_t)NB * w.ldv, w.ldv, (int)shift);
          }
          CHECK(rs_panel_fast(pr, w, cmL, k, is_tsqr(k), Vdst, sl.T[idx], sl.Tt[idx], sl.bc[idx], &S.ticket[h % RS_RING][idx],
                              idx == 1 ? sl.V + shift : nullptr, sl.S2));
          (*nfast)++;
        } else {
          // partial / short panels and the panel a resumed pass starts with: column by column.  Inside a pass they may
          // only run if nothing was rejected before (one status read; every rank holds the same status).
          if (k != kstart) {
            int f = 0;
            CHECK(status_read(c, &f));
            if (f != INT_MAX) {
              stop = true;
              return DHQR_OK;
            }
          }
          CHECK(rs_panel_columns(pr, w, cmL, c0, wcols, Vdst, sl.T[idx], sl.Tt[idx]));
        }
      }
      return DHQR_OK;
    };
    const int32_t rc = body();
    c->profiling = was;
    CHECK(rc);
    CHECK(prof_end(c));
    if (la) HIPCHECK(hipEventRecord(S.ev_group[h % (2 * RS_RING)], sL));
    LAUNCHCHECK();
    return DHQR_OK;
  };

  auto body = [&]() -> int32_t {
    if (la) {  // order the lane after whatever the caller queued (e.g. the fill)
      HIPCHECK(hipEventRecord(S.ev_start, sW));
      HIPCHECK(hipStreamWaitEvent(sL, S.ev_start, 0));
    }
    CHECK(produce(0));
    for (int g = 0; g < G && !stop; ++g) {
      const RsGroup &gr = groups[g];
      if (gr.last() + 1 >= K) break;  // nothing to the right of this group
      on(sW, 0);
      if (la) {
        // wide: group g -> the columns beyond group g+1 (group g+1's own columns are the lane's: produce(g+1))
        HIPCHECK(hipStreamWaitEvent(sW, S.ev_group[g % (2 * RS_RING)], 0));
        const int64_t col0 = std::min<int64_t>(n, (groups[g + 1].last() + 1) * NB);
        CHECK(rs_group_apply(pr, w, w.slot[g % RS_RING], gr, col0, n - col0, cmW));
        HIPCHECK(hipEventRecord(S.ev_wide[g % (2 * RS_RING)], sW));
      } else {
        // one stream: the group is applied to everything to its right in one pass, then the next group is factored
        const int64_t col0 = (gr.last() + 1) * NB;
        CHECK(rs_group_apply(pr, w, w.slot[g % RS_RING], gr, col0, n - col0, cmW));
      }
      CHECK(produce(g + 1));
    }
    on(sW, 0);
    if (la) {  // join: the caller's stream owns the result
      HIPCHECK(hipEventRecord(S.ev_end, sL));
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

// Panel k (already factored and committed) applied again to the columns [col0, n): the resume after the SECOND panel of
// a pair was rejected -- the pair's first panel had reached that panel's columns only.
static int32_t rs_reapply_panel(const RsProblem &pr, const RsWork &w, int64_t k, int64_t col0) {
  dhqr_ctx *c = pr.c;
  const int64_t NB = DHQR_NBV, c0 = k * NB, ncols = pr.n - col0;
  if (ncols <= 0) return DHQR_OK;
  int64_t off, rows;
  pr.active(c0, &off, &rows);
  const bool diag_owner = pr.owner_of_row(c0) == pr.r;
  const int saved_epoch = c->epoch;
  c->epoch = -1;
  auto body = [&]() -> int32_t {
    if (rows > 0) CHECK(rs_pack(pr, w.Vw, w.ldv, c0, NB, off, rows, diag_owner));
    CHECK(rs_gram_allreduce(pr, pr.cm, w.Vw, w.ldv, rows, w.S));
    launch_build_t(c, w.S, (int)NB, w.T, w.Tt);
    double *C = pr.A + off + col0 * pr.lda;
    CHECK(rs_vtc_allreduce(pr, pr.cm, w.Vw, w.ldv, C, pr.lda, rows, ncols));
    return rs_apply_w(pr, w.Vw, w.ldv, w.T, C, pr.lda, rows, ncols, false);
  };
  const int32_t rc = body();
  c->epoch = saved_epoch;
  return rc;
}

static int32_t rs_factor(const RsProblem &pr) {
  dhqr_ctx *c = pr.c;
  RsWork w;
  CHECK(rs_prepare(pr, &w));
  CHECK(status_reset(c));
  const int64_t K = (pr.n + DHQR_NBV - 1) / DHQR_NBV;
  int64_t ks = 0;
  int level = 0;  // ladder for a rejected panel: Gram/Cholesky (0) -> TSQR tree (1) -> column by column (2)
  std::vector<int64_t> pair_b;
  for (int pass = 0; ks < K; ++pass) {
    if (pass > 2 * K + 2) return set_err(DHQR_EINVAL, "internal error: the row-split driver does not make progress");
    int failed = INT_MAX;
    int64_t nfast = 0;
    pair_b.clear();
    CHECK(rs_run(pr, w, ks, level, &failed, &nfast, pair_b));
    const int64_t accepted = (failed == INT_MAX) ? nfast : std::max<int64_t>(0, std::min<int64_t>(nfast, failed - ks));
    c->n_fast += accepted;
    if (c->cholqr_passes == 3) c->n_tsqr += (int)accepted;
    else if (level == 1 && accepted > 0) c->n_tsqr++;
    if (failed == INT_MAX) break;
    CHECK(status_reset(c));
    // the rejected panel was the second of a pair: its (committed) first panel has only reached the rejected panel's columns
    if (std::find(pair_b.begin(), pair_b.end(), (int64_t)failed) != pair_b.end())
      CHECK(rs_reapply_panel(pr, w, failed - 1, ((int64_t)failed + 1) * DHQR_NBV));
    if (failed == ks && level >= 1) {
      level = 2;  // the tree's R did not pass either: the reference's column-by-column algorithm
    } else {
      const bool rung = c->tsqr_rung == 1 || (c->tsqr_rung < 0 && pr.cm && pr.P > 1);
      level = (c->cholqr_passes == 3 || !rung) ? 2 : 1;
    }
    if (level == 2) c->n_fallback++;
    ks = failed;
  }
  return DHQR_OK;
}

// ||A - QR||_F / ||A||_F: every rank forms ITS ROWS of Q*R by re-applying the panels in reverse order to [R; 0]
// (one all-reduce of V'B per panel), A regenerated from `seed`.  dB, dA0: mloc x n scratch (ld = max(mloc,1)).
static int32_t rs_residual(const RsProblem &pr, uint64_t seed, double *dB, double *dA0, double *hrel) {
  dhqr_ctx *c = pr.c;
  RsWork w;
  CHECK(rs_prepare(pr, &w));
  const int64_t NB = DHQR_NBV, n = pr.n, K = (n + NB - 1) / NB, ldb = std::max<int64_t>(pr.mloc, 1);
  const bool was = c->profiling;
  c->profiling = false;
  c->epoch = -1;
  auto body = [&]() -> int32_t {
    if (pr.mloc > 0) {
      // [R; 0] rows of this rank: global row g = row0 + i carries R[g, :] for g < n
      dim3 grid((unsigned)std::min<int64_t>((pr.mloc + 255) / 256, 128), (unsigned)std::min<int64_t>(n, 32768));
      hipLaunchKernelGGL(k_form_r0_rows, grid, dim3(256), 0, c->stream, (const double *)pr.A, pr.lda, (const double *)pr.alpha,
                         pr.mloc, n, pr.row0, dB, ldb);
    }
    for (int64_t k = K - 1; k >= 0; --k) {
      const int64_t c0 = k * NB, wcols = std::min<int64_t>(NB, n - c0);
      int64_t off, rows;
      pr.active(c0, &off, &rows);
      const bool diag_owner = pr.owner_of_row(c0) == pr.r;
      if (rows > 0) CHECK(rs_pack(pr, w.Vw, w.ldv, c0, wcols, off, rows, diag_owner));
      CHECK(rs_gram_allreduce(pr, pr.cm, w.Vw, w.ldv, rows, w.S));
      launch_build_t(c, w.S, (int)wcols, w.T, w.Tt);
      const int64_t ncols = n - c0;
      double *C = dB + off + c0 * ldb;
      CHECK(rs_vtc_allreduce(pr, pr.cm, w.Vw, w.ldv, C, ldb, rows, ncols));
      CHECK(rs_apply_w(pr, w.Vw, w.ldv, w.Tt, C, ldb, rows, ncols, false));  // Q (not Q'): op(T) = T
    }
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  double h[2] = {0.0, 0.0};
  if (pr.mloc > 0) {
    const int64_t total = pr.mloc * n;
    const unsigned gridf = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fill_uniform, dim3(gridf), dim3(256), 0, c->stream, dA0, pr.mloc, n, ldb, seed, pr.m, pr.row0, NB, 1, 0);
    const int nblk = 1024;
    hipLaunchKernelGGL(k_diff_norms, dim3(nblk), dim3(256), 0, c->stream, (const double *)dA0, ldb, (const double *)dB, ldb,
                       pr.mloc, n, c->scratch.p);
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

// r6: b <- Q'b (src:215-242) of the row split at P > 1 on the kernels of dhqr_qtb.h.  The per-panel form below re-packs V,
// all-reduces a 128 x 128 Gram matrix, builds T and pushes ONE right-hand side through the 128-column MFMA tiles: two
// collectives and ~8 launches per panel.  Here:
//   pre-pass (independent of b)  every rank's batched Gram products of ALL panels on its own rows, V read in place -- the
//     panels whose top block it holds with the triangle of R masked (rows addressed globally through a shifted base
//     pointer), the panels that start above its rows unmasked over its whole row range -- then ONE all-reduce of the np
//     Gram matrices and the batched T' = (I + striu(S))^{-T};
//   per panel step               k_qtb_step on the local rows (update by panel k - 1, partial dots of panel k; `goff` = the
//     rank's first global row), ONE all-reduce of the 128 dots -- the north star's "all-reduce of the cross-partition
//     partial dots" -- and w_k = T_k' y_k by k_qtb_tw on every rank.
static int32_t rs_qtb_pipelined(const RsProblem &pr, double *db) {
  dhqr_ctx *c = pr.c;
  dhqr_comm *cm = pr.cm;
  const int64_t NB = DHQR_NBV, n = pr.n, R0 = pr.row0, ml = pr.mloc, lda = pr.lda;
  const int np = (int)((n + NB - 1) / NB);
  // panels: k < kA start above this rank's rows (unmasked, all local rows); kA <= k < kB have their top block here; the rest
  // lie below
  const int kA = (int)std::min<int64_t>(np, (R0 + NB - 1) / NB), kB = (int)std::min<int64_t>(np, (R0 + ml + NB - 1) / NB);
  int64_t total = 0;
  for (int k = 0; k < kB; ++k) total += (k < kA) ? ml : R0 + ml - (int64_t)k * NB;
  int64_t rps = ((total / 1024 + 15) / 16) * 16;
  rps = std::min<int64_t>(std::max<int64_t>(rps, 256), 4096);
  std::vector<int> tab((size_t)3 * (np + 1), 0);  // merged | unmasked launch | masked launch
  int *merged = tab.data(), *t1 = merged + (np + 1), *t2 = t1 + (np + 1);
  for (int k = 0; k < np; ++k) {
    const int64_t rows = (ml <= 0 || k >= kB) ? 0 : ((k < kA) ? ml : R0 + ml - (int64_t)k * NB);
    merged[k + 1] = merged[k] + (int)((rows + rps - 1) / rps);
  }
  const int U2 = merged[kA], U = merged[np];  // units of the unmasked launch (they come first), of both
  for (int k = 0; k <= np; ++k) {
    t1[k] = std::min(merged[k], U2);
    t2[k] = std::max(merged[k] - U2, 0);
  }
  const bool vecA = (lda % 2 == 0) && (ml % 2 == 0) && (R0 % 2 == 0) && aligned16(pr.A);
  const int VEC = (c->qtb_vec == 1 || !(vecA && aligned16(db))) ? 1 : (c->qtb_vec == 2 ? 2 : (ml >= 16384 ? 2 : 1));
  const int64_t SS = 64 * VEC, maxsl = std::max<int64_t>(8, std::min<int64_t>(c->ncu, 256));
  const int64_t sl = SS * std::max<int64_t>(1, (ml + SS * maxsl - 1) / (SS * maxsl)), nsl = (ml + sl - 1) / sl;
  CHECK(ensure(c, c->sv_T, (size_t)np * QTB_NB2));
  CHECK(ensure(c, c->sv_S, (size_t)np * QTB_NB2));
  CHECK(ensure(c, c->sv_part, (size_t)std::max(U, 1) * QTB_NB2));
  const size_t n_ypart = (size_t)std::max<int64_t>(nsl + 2, 8) * QTB_NB, n_w = (size_t)(np + 1) * QTB_NB, n_y = (size_t)np * QTB_NB;
  const size_t n_ints = tab.size() + (size_t)(np + 1) + 4;
  CHECK(ensure(c, c->sv_small, n_ypart + n_w + n_y + (n_ints + 1) / 2 + 16));
  double *ypart = c->sv_small.p, *wbuf = ypart + n_ypart, *ydist = wbuf + n_w;
  int *ints = reinterpret_cast<int *>(ydist + n_y);
  int *tab_dev = ints, *counters = tab_dev + tab.size(), *zero = counters + (np + 1);
  HIPCHECK(hipMemsetAsync(ydist, 0, (n_y + (n_ints + 1) / 2 + 8) * sizeof(double), c->stream));  // the dots of absent ranks, counters, flags
  HIPCHECK(hipMemcpyAsync(tab_dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));  // (the table is a host vector)
  const int *dm = tab_dev, *d1 = tab_dev + (np + 1), *d2 = d1 + (np + 1);
  int *err = c->zflags + DHQR_PIPE_ERR_OFFSET;
  auto sync_local = [&]() -> int32_t {
    if (cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
    return DHQR_OK;
  };
  // ---- pre-pass
  if (U2 > 0) {  // panels that start above this rank's rows
    if (vecA)
      hipLaunchKernelGGL((k_gemm_tn_gram_batch<2, false>), dim3((unsigned)U2), dim3(256), 0, c->stream, (const double *)pr.A, lda, ml, n,
                         rps, d1, np, c->sv_part.p, (const int *)zero);
    else
      hipLaunchKernelGGL((k_gemm_tn_gram_batch<1, false>), dim3((unsigned)U2), dim3(256), 0, c->stream, (const double *)pr.A, lda, ml, n,
                         rps, d1, np, c->sv_part.p, (const int *)zero);
  }
  if (U - U2 > 0) {  // panels whose top block lives here: global row numbers through the shifted base
    const double *Ag = pr.A - R0;
    if (vecA)
      hipLaunchKernelGGL((k_gemm_tn_gram_batch<2, true>), dim3((unsigned)(U - U2)), dim3(256), 0, c->stream, Ag, lda, R0 + ml, n, rps,
                         d2, np, c->sv_part.p + (size_t)U2 * QTB_NB2, (const int *)zero);
    else
      hipLaunchKernelGGL((k_gemm_tn_gram_batch<1, true>), dim3((unsigned)(U - U2)), dim3(256), 0, c->stream, Ag, lda, R0 + ml, n, rps,
                         d2, np, c->sv_part.p + (size_t)U2 * QTB_NB2, (const int *)zero);
  }
  hipLaunchKernelGGL(k_qtb_sum_gram, dim3((unsigned)np, 16), dim3(256), 0, c->stream, (const double *)c->sv_part.p, dm, n, c->sv_S.p,
                     (const int *)zero);
  LAUNCHCHECK();
  CHECK(comm_allreduce_sum(cm, c->sv_S.p, (int64_t)np * QTB_NB2, c->stream));
  CHECK(sync_local());
  hipLaunchKernelGGL(k_build_t_batch, dim3((unsigned)np), dim3(1024), 0, c->stream, (const double *)c->sv_S.p, n, c->sv_T.p,
                     (const int *)zero);
  // ---- panel steps
  for (int k = 0; k <= np; ++k) {
    const int64_t rglob = (int64_t)(k >= 1 ? k - 1 : 0) * NB, rfl = std::max<int64_t>(0, rglob - R0);
    if (ml > 0 && rfl < ml) {
      const unsigned grid = (unsigned)(nsl - rfl / sl);
      if (VEC == 2)
        hipLaunchKernelGGL((k_qtb_step<2>), dim3(grid), dim3(256), 0, c->stream, (const double *)pr.A, lda, ml, n, k, np, sl, db,
                           (const double *)c->sv_T.p, (const double *)c->sv_T.p, (const int *)zero, wbuf, ypart, counters, err, R0, ydist);
      else
        hipLaunchKernelGGL((k_qtb_step<1>), dim3(grid), dim3(256), 0, c->stream, (const double *)pr.A, lda, ml, n, k, np, sl, db,
                           (const double *)c->sv_T.p, (const double *)c->sv_T.p, (const int *)zero, wbuf, ypart, counters, err, R0, ydist);
      LAUNCHCHECK();
    }
    if (k < np) {
      CHECK(comm_allreduce_sum(cm, ydist + (size_t)k * QTB_NB, QTB_NB, c->stream));
      CHECK(sync_local());
      hipLaunchKernelGGL(k_qtb_tw, dim3(1), dim3(256), 0, c->stream, (const double *)(c->sv_T.p + (size_t)k * QTB_NB2),
                         (const double *)(ydist + (size_t)k * QTB_NB), wbuf + (size_t)k * QTB_NB);
      LAUNCHCHECK();
    }
  }
  return DHQR_OK;
}

// `H \ b` (src:317-321) for the row split: db = this rank's rows of b (mloc, overwritten); dx (n) <- x on every rank.
// Q'b (src:215-242): per panel the partial dots V_r' b_r are all-reduced (a 128-vector), then b_r -= V_r (T' w)
// locally.  The back substitution (src:244-254) needs R = the top n rows: each 128-row block is solved by the rank
// that owns those rows and broadcast; the ranks owning rows above subtract their part.
static int32_t rs_solve(const RsProblem &pr, double *db, double *dx) {
  dhqr_ctx *c = pr.c;
  dhqr_comm *cm = (pr.cm && pr.P > 1) ? pr.cm : nullptr;
  // one rank: its rows are the matrix -- the solve of dhqr_qtb.h (see cs_solve), x copied out of b[0:n]
  if (!cm && pr.P == 1 && c->solve_pipe && pr.mloc == pr.m) {
    CHECK(prof_begin(c, CAT_SOLVE));
    const bool was1 = c->profiling;
    c->profiling = false;
    int32_t rc1 = solve_pipelined(c, pr.A, pr.m, pr.n, pr.lda, pr.alpha, db, true);
    if (rc1 == DHQR_OK && dx != db)
      rc1 = hipMemcpyAsync(dx, db, (size_t)pr.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? DHQR_OK : DHQR_EHIP;
    c->profiling = was1;
    CHECK(rc1);
    CHECK(prof_end(c));
    return DHQR_OK;
  }
  RsWork w;
  CHECK(rs_prepare(pr, &w));
  const int64_t NB = DHQR_NBV, n = pr.n, K = (n + NB - 1) / NB, ldb = std::max<int64_t>(pr.mloc, 1);
  const bool was = c->profiling;
  CHECK(prof_begin(c, CAT_SOLVE));
  c->profiling = false;
  c->epoch = -1;
  auto sync_local = [&]() -> int32_t {
    if (cm && cm->kind == COMM_LOCAL) {
      HIPCHECK(hipStreamSynchronize(c->stream));
      CHECK(comm_host_barrier(cm));
    }
    return DHQR_OK;
  };
  auto body = [&]() -> int32_t {
    if (cm && c->solve_pipe) {
      CHECK(rs_qtb_pipelined(pr, db));  // r6: Q'b on the kernels of dhqr_qtb.h
    } else {
    for (int64_t k = 0; k < K; ++k) {
      const int64_t c0 = k * NB, wcols = std::min<int64_t>(NB, n - c0);
      int64_t off, rows;
      pr.active(c0, &off, &rows);
      const bool diag_owner = pr.owner_of_row(c0) == pr.r;
      if (rows > 0) CHECK(rs_pack(pr, w.Vw, w.ldv, c0, wcols, off, rows, diag_owner));
      CHECK(rs_gram_allreduce(pr, pr.cm, w.Vw, w.ldv, rows, w.S));
      launch_build_t(c, w.S, (int)wcols, w.T, w.Tt);
      CHECK(rs_vtc_allreduce(pr, pr.cm, w.Vw, w.ldv, db + off, ldb, rows, 1));
      CHECK(rs_apply_w(pr, w.Vw, w.ldv, w.T, db + off, ldb, rows, 1, false));
    }
    }
    // back substitution, block by block from the bottom of R: x_blk solved by the owner of rows [c0, c0 + w)
    HIPCHECK(hipMemsetAsync(dx, 0, (size_t)n * sizeof(double), c->stream));
    for (int64_t k = K - 1; k >= 0; --k) {
      const int64_t c0 = k * NB, wcols = std::min<int64_t>(NB, n - c0);
      const int downer = pr.owner_of_row(c0);
      if (downer == pr.r) {
        // rows [c0, c0 + w) are local rows [c0 - row0, ...): R block rows sit in pr.A at those local rows;
        // dhqr_backsub_block addresses column j of R at base + j*lda and rows by GLOBAL index i: shift the base
        const double *base = pr.A - pr.row0;  // so that (global row i, column j) is base[i + j*lda]
        CHECK(dhqr_backsub_block_f64(c, base, pr.lda, pr.alpha, db - pr.row0, c0, c0 + wcols, 1, 0));
        HIPCHECK(hipMemcpyAsync(dx + c0, db + (c0 - pr.row0), (size_t)wcols * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      }
      if (cm) {
        CHECK(comm_bcast(cm, dx + c0, wcols, downer, c->stream, nullptr));
        CHECK(sync_local());
      }
      // every rank that owns R rows above the block subtracts R[rows, blk] x_blk from its part of b
      const int64_t lo = pr.row0, hi = std::min<int64_t>(pr.row0 + pr.mloc, c0);
      if (hi > lo)
        hipLaunchKernelGGL(k_rs_backsub_update, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, c->stream,
                           (const double *)pr.A, pr.lda, db, hi - lo, c0, (int)wcols, (const double *)(dx + c0));
      LAUNCHCHECK();
    }
    return DHQR_OK;
  };
  const int32_t rc = body();
  c->profiling = was;
  CHECK(rc);
  CHECK(prof_end(c));
  return DHQR_OK;
}
