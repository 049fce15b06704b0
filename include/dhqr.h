/*
 * dhqr.h -- C ABI of libdhqr.so: MI355X (gfx950) Householder QR hot path, drop-in for the
 * Julia functions of jwscook/DistributedHouseholderQR.jl (src/DistributedHouseholderQR.jl).
 *
 * The reference has no FFI layer; its boundary is the Julia function API.  Each entry point
 * below names the reference function (file:line) it replaces.  A Julia `ccall` wrapper with the
 * reference's names (`qr!`, `\`, `householder!`, `solve_householder!`, `partialdot`) is in
 * distributedhouseholderqr.jl_amd/julia/src/DistributedHouseholderQR.jl; the same ABI is bound from
 * Python (ctypes) by distributedhouseholderqr.jl_amd/_lib.py.  INTEGRATION.md shows both stubs.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns int32 status (0 = DHQR_OK, <0 error),
 *     message via dhqr_last_error() (thread-local); nothing throws across the boundary.
 *   - sizes are int64 (Julia Int); matrices are column-major Float64 with leading dimension ld*.
 *   - "d"-prefixed pointers are DEVICE pointers owned by the caller (hipMalloc, torch tensor
 *     .data_ptr(), AMDGPU.jl ROCArray pointer ...); "h"-prefixed pointers are HOST pointers,
 *     borrowed for the duration of the call.
 *   - device-resident entry points enqueue work on the context's stream and return without
 *     synchronising unless they hand a host scalar back (documented per function).
 *   - one dhqr_ctx per (host thread, GPU); a ctx is not re-entrant.
 *   - a matrix without columns (n == 0, m >= 0) is a no-op for the factor / solve entry points, like the
 *     reference's empty loops (src:127, src:217); m < n, negative sizes and ld < m are DHQR_EINVAL.
 *   - factor format (identical to the reference, src:296-309): after factorisation
 *       A[j:m, j] = v_j  (diagonal INCLUDED, ||v_j||^2 = 2, H_j = I - v_j v_j'),
 *       A[i, j]  = R[i,j] for i < j,   alpha[j] = R[j,j].
 */
#ifndef DHQR_H
#define DHQR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHQR_VERSION 500 /* 0.5.0: round 6 (dhqr_set_small_route added; nothing else changed) */

#define DHQR_OK 0
#define DHQR_EINVAL (-1)   /* bad argument (null pointer, m < n, ld < m, unsupported nb ...) */
#define DHQR_EHIP (-2)     /* a HIP runtime call failed; text in dhqr_last_error() */
#define DHQR_ENOMEM (-3)   /* workspace allocation failed */
#define DHQR_ENODEVICE (-4) /* no gfx950 device visible */
#define DHQR_ECOMM (-5)    /* rank-to-rank transport failed (RCCL error, peer rank failed, callback error) */

/* Panel (block-reflector) width of the blocked path.  BASELINE.json configs 3/4 fix it at 128. */
#define DHQR_NB 128
/* Cyclic block of the 1-D block-cyclic column split (dhqr_cs_*, dhqr_mg_*): TWO panels.  Global column j lives on rank
 * (j / DHQR_CS_BLOCK) % P at local column ((j / DHQR_CS_BLOCK) / P) * DHQR_CS_BLOCK + j % DHQR_CS_BLOCK. */
#define DHQR_CS_BLOCK 256

typedef struct dhqr_ctx dhqr_ctx; /* opaque: device id, stream, workspaces, event pools */

/* Per-phase device timings (ms, hipEvent on the ctx stream) and launch counts accumulated since
 * the last dhqr_reset_stats(); only filled while profiling is enabled (dhqr_set_profiling).
 * Replaces the reference's inline @elapsed accumulators t1a/t1b/t2 (src:126-146, src:291). */
typedef struct dhqr_stats {
  double ms_panel;      /* reflector construction + in-panel rank-1 updates   (src:122-148)  */
  double ms_tbuild;     /* pack V, V'V, T recurrence                           (new)          */
  double ms_gemm_vta;   /* W = V' * A     trailing GEMM 1 (MFMA)               (src:208)      */
  double ms_gemm_tw;    /* W = T' * W     small GEMM (MFMA)                    (new)          */
  double ms_gemm_avw;   /* A -= V * W     trailing GEMM 2 (MFMA)               (src:209)      */
  double ms_rank1;      /* unblocked fused reflector-apply kernel (nb = 0)     (src:198-213)  */
  double ms_solve;      /* Q'b + back substitution                             (src:215-294)  */
  int64_t n_panel, n_tbuild, n_gemm_vta, n_gemm_tw, n_gemm_avw, n_rank1, n_solve; /* launches */
  double flops_gemm_vta, flops_gemm_avw; /* algorithmic flops issued by the two trailing GEMMs */
  double bytes_rank1;                    /* HBM bytes of the nb = 0 launches AS IMPLEMENTED: 16 B per element of every column a
                                          * pass loads and stores once; a pass applies K reflectors (K = 5 where a column
                                          * fits 8192 rows), i.e. 16/K B per trailing element and reflector */
  double bytes_panel;                    /* algorithmic bytes of the panel-lane launches (16 B per element touched) */
} dhqr_stats;

/* ------------------------------------------------------------------ library / context */
int32_t dhqr_version(void);
const char *dhqr_last_error(void);
int32_t dhqr_device_count(int32_t *count);

/* Create a context on HIP device `device` (fails with DHQR_ENODEVICE when no GPU is visible:
 * there is NO CPU fallback).  Owns a stream and lazily grown workspaces. */
int32_t dhqr_create(dhqr_ctx **ctx, int32_t device);
int32_t dhqr_destroy(dhqr_ctx *ctx);
/* Run on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL means the
 * device's default (null) stream -- that IS torch's default stream.  dhqr_use_own_stream switches
 * back to the ctx-owned non-blocking stream (the state after dhqr_create). */
int32_t dhqr_set_stream(dhqr_ctx *ctx, void *hip_stream);
int32_t dhqr_use_own_stream(dhqr_ctx *ctx);
/* Waits for the ctx stream AND reports what only the device knows: the library's inter-workgroup pipelines (the ComplexF64
 * panel pipeline, the lead pipeline of the unblocked path, the flag-pipelined back substitution and the persistent Q'b
 * kernel of dhqr_solve_f64) bound their waits; a wait that expires lets its kernel finish with wrong numbers and records
 * the fact in an error word.  RESULTS OF AN ASYNCHRONOUS ENTRY POINT (dhqr_factor_f64 with nb == 0, dhqr_factor_c64*,
 * dhqr_solve_f64 / _c64, dhqr_cs_factor_c64) ARE VALID ONLY AFTER dhqr_synchronize RETURNED DHQR_OK.  The synchronous
 * entry points and every blocked driver (one status read per pass) check the word themselves. */
int32_t dhqr_synchronize(dhqr_ctx *ctx);
/* Release what the context keeps between calls for speed: the device copy and staging buffers of the host-in / host-out
 * entry points, the solve's workspaces and kept T factors, the blocked driver's group buffers.  Synchronises; the next call
 * that needs them allocates again. */
int32_t dhqr_trim(dhqr_ctx *ctx);
int32_t dhqr_set_profiling(dhqr_ctx *ctx, int32_t on);
int32_t dhqr_reset_stats(dhqr_ctx *ctx);
int32_t dhqr_get_stats(dhqr_ctx *ctx, dhqr_stats *out); /* synchronises the ctx stream */

/* Panels factored since the last dhqr_reset_stats() by the R-first fast path (csrc/dhqr_recon.h)
 * and panels whose verification failed and were redone by the column-by-column path. */
int32_t dhqr_get_panel_counters(dhqr_ctx *ctx, int64_t *n_fast, int64_t *n_fallback);
/* Where the R-first panel path takes its factors from (all drivers): 1 = Gram matrix + Cholesky, reflectors from the
 * panel itself (default), 2 = CholeskyQR2, 3 = TSQR-HR (csrc/dhqr_tsqr.h: 256-row leaves, pairwise reduction of the
 * 128 x 128 R factors -- across the ranks in the row-split driver --, explicit orthonormal Q back down the tree,
 * reflectors from Q: accuracy independent of the panel's condition number).  A panel rejected by the on-device
 * verification climbs the ladder 1 -> 2 (-> 3, see dhqr_set_tsqr_rung; row split: 1 -> 3) before it is redone column
 * by column.  Environment: DHQR_CHOLQR_PASSES=2, DHQR_TSQR=1, DHQR_TSQR_RUNG=0/1.
 * dhqr_get_tsqr_count: accepted panels that went through the tree since the last dhqr_reset_stats(). */
int32_t dhqr_set_r_source(dhqr_ctx *ctx, int32_t source);
/* The TSQR-HR rung of the ladder: on = 1 always, on = 0 never.  Default (neither called nor DHQR_TSQR_RUNG set): only in
 * the row-split driver on more than one rank, where the column-by-column rung costs two collectives per column; on a
 * single GPU the column kernels redo a panel faster than the tree. */
int32_t dhqr_set_tsqr_rung(dhqr_ctx *ctx, int32_t on);
/* The tree alone: R (128 x 128, column-major, upper triangular, zeros below) of a device-resident rows x 128 panel
 * (leading dimension ldp), unique up to the sign of each row (the replay of csrc/dhqr_recon.h fixes the reference's
 * signs R_jj = alpha_j when the tree feeds a factorisation).  Async on the ctx stream. */
int32_t dhqr_tsqr_r_f64(dhqr_ctx *ctx, const double *dP, int64_t rows, int64_t ldp, double *dR);
int32_t dhqr_get_tsqr_count(dhqr_ctx *ctx, int64_t *n_tsqr);
/* Small matrices -- the reference's own test shapes start at 110 x 100, test/runtests.jl:42 -- take ONE single-workgroup
 * launch per qr! (m <= 128 and n <= 128, m <= 224 and n <= 224, or m <= 256 and n <= 192: the matrix lives in the
 * registers of one compute unit, the reference's column-by-column algorithm src:122-148,198-213 as written) and per `\`
 * (m <= 256), whatever `nb` says; the host-array entry points dhqr_qr_f64 / dhqr_ldiv_f64 then run the kernel directly
 * on a pinned staging buffer (csrc/dhqr_small.h).  on = 0: the general drivers for every shape (also DHQR_SMALL=0).
 * Above 128 rows the kernel's waves hand columns and reflectors to each other through LDS flags instead of a barrier per
 * column; those waits are bounded, and a kernel that gave up on one answers NaN in every alpha (never a hung device):
 * dhqr_qr_f64 then factors once more with the barrier form by itself; a caller of the device-resident, asynchronous
 * dhqr_factor_f64 sees the NaN (DHQR_TUNE small_flags=0 selects the barrier form from the start). */
int32_t dhqr_set_small_route(dhqr_ctx *ctx, int32_t on);
/* Solves this context REPEATED with one launch per panel step because a wait of the persistent Q'b kernel (all of whose
 * workgroups must be resident at once) expired -- another process or stream held compute units.  The repetition happens
 * inside the first synchronising entry point after the solve (dhqr_synchronize, dhqr_ldiv_f64, ...), from a copy of b
 * taken before the persistent launch; the caller sees a correct x and DHQR_OK. */
int32_t dhqr_get_solve_retries(dhqr_ctx *ctx, int64_t *n_retries);

/* ------------------------------------------------------------------ synthetic inputs
 * Replaces rand(T,m,n) / rand(T,m) of test/runtests.jl:45-46 with the portable counter-based
 * generator shared with oracle/ :  value(gi, gj) = u01(seed, gi + gj*global_m).
 * The local block holds `rows` x `cols`; local row il is global row row0+il; local column jl is
 * global column ((jl / colblock) * nranks + rank) * colblock + jl % colblock  (block-cyclic 1-D
 * column layout; nranks = 1, rank = 0 gives the identity map). Async. */
int32_t dhqr_fill_uniform_f64(dhqr_ctx *ctx, double *dA, int64_t rows, int64_t cols, int64_t lda,
                              uint64_t seed, int64_t global_m, int64_t row0, int64_t colblock,
                              int32_t nranks, int32_t rank);

/* ------------------------------------------------------------------ factorisation
 * dhqr_factor_f64: device-resident replacement of householder!(A, alpha) (src:113, src:122-148,
 * src:198-213) for one GPU.  In place on dA (m x n, m >= n), writes dalpha[0:n].
 *   nb == 0      : unblocked path (BASELINE configs[1]) -- the reference's operations in the reference's order
 *                  (src:129-143, 208-209: each dot product over the column as updated so far); a launch applies K
 *                  consecutive reflectors to every trailing column it loads once (K = 5 ... 8 for columns of up to
 *                  8192 rows, 5 up to 32768 rows, one reflector per launch above) and builds the next K reflectors
 *                  in its lead workgroup(s).
 *   nb == DHQR_NB: blocked path -- panel factorisation + compact-WY trailing update
 *                  A -= V * (T' * (V' * A)) on FP64 MFMA (BASELINE configs[2]).
 * nb == 0 is asynchronous on the ctx stream.  nb == DHQR_NB enqueues the whole factorisation without host
 * synchronisation and then waits ONCE for the device-side panel verification word (a rejected panel is redone by the
 * robust ladder before returning): synchronous on return. */
int32_t dhqr_factor_f64(dhqr_ctx *ctx, double *dA, int64_t m, int64_t n, int64_t lda,
                        double *dalpha, int32_t nb);

/* dhqr_qr_f64: host-in / host-out drop-in for qr!(A::Matrix{Float64}) (src:311-315): uploads hA,
 * factors, downloads hA and halpha.  Synchronous.  Column blocks travel back while later panels are still being
 * factored, so ON AN ERROR RETURN hA IS UNDEFINED (a mix of original and factored columns) and halpha is not written.
 * The device copy of the matrix (and the pinned staging buffers) stay in the context between calls -- 8 GiB at 32768^2;
 * dhqr_trim releases them. */
int32_t dhqr_qr_f64(dhqr_ctx *ctx, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha,
                    int32_t nb);

/* ------------------------------------------------------------------ solve
 * dhqr_solve_f64: device-resident replacement of solve_householder!(b, H, alpha) (src:284-294):
 * db (length m) <- Q' db (src:215-242), then back substitution with strict-upper dA and the
 * diagonal dalpha (src:244-282); the solution is db[0:n].  Mutates db like the reference.  Asynchronous (see
 * dhqr_synchronize).  csrc/dhqr_qtb.h: per 128-column panel the compact-WY form I - V T' V' with V read in place, GEMV-class
 * kernels (every element of V and of R crosses the memory system once), one persistent launch for Q'b and one
 * flag-pipelined launch for the back substitution.  T' of every panel comes from a batched pre-pass over the factor --
 * or, when this context's last blocked dhqr_factor_f64 was of this very matrix (same dA, m, n, lda) and dalpha still
 * holds that factorisation's values (compared on the device), from what the factorisation kept (DHQR_KEEP_T). */
int32_t dhqr_solve_f64(dhqr_ctx *ctx, const double *dA, int64_t m, int64_t n, int64_t lda,
                       const double *dalpha, double *db);

/* One block step of the back substitution (src:244-282) for rows/columns [lo, hi):
 *   do_diag  : solve the diagonal block in place in db[lo:hi] (divide by dalpha[lo:hi]);
 *   do_update: db[0:lo] -= R[0:lo, lo:hi] * db[lo:hi].
 * dAcols is addressed by GLOBAL column index: column j of R is read at dAcols + j*lda (a
 * column-split caller passes its local block shifted accordingly).  dhqr_solve_f64 is this call
 * looped over blocks; the distributed solve interleaves it with the all-reduce of the partial
 * dots (replacing sum(fetch.(futures)), src:262-266).  Async. */
int32_t dhqr_backsub_block_f64(dhqr_ctx *ctx, const double *dAcols, int64_t lda, const double *dalpha,
                               double *db, int64_t lo, int64_t hi, int32_t do_diag, int32_t do_update);

/* dhqr_ldiv_f64: host-in / host-out drop-in for `H \ b` (src:317-321): does NOT modify hb (the
 * reference copies b into a SharedArray first); writes hx[0:n].  Synchronous. */
int32_t dhqr_ldiv_f64(dhqr_ctx *ctx, const double *hA, int64_t m, int64_t n, int64_t lda,
                      const double *halpha, const double *hb, double *hx);

/* KAT hook mirroring partialdot(a, b, lo:hi, Float64) (src:42-49; test/partialdot.jl:18):
 * sum_{i=lo}^{hi-1} da[i]*db[i] (0-based, hi exclusive) reduced on the device with the same
 * wavefront-shuffle + LDS tree the factor kernels use.  Synchronous (returns a host scalar). */
int32_t dhqr_partialdot_f64(dhqr_ctx *ctx, const double *da, const double *db, int64_t lo,
                            int64_t hi, double *hout);
/* Same with HOST vectors (uploads ha[lo:hi], hb[lo:hi]); what the Julia module's partialdot binds. */
int32_t dhqr_partialdot_host_f64(dhqr_ctx *ctx, const double *ha, const double *hb, int64_t lo,
                                 int64_t hi, double *hout);

/* ------------------------------------------------------------------ ComplexF64 methods
 * The reference's qr!/`\` are generic over the element type and its tests run every shape for
 * ComplexF64 too (test/runtests.jl:43); these are the ComplexF64 counterparts of the Float64 entry
 * points above (unblocked path).  A complex element is an interleaved
 * (re, im) pair of doubles == Julia's ComplexF64, so a `Ptr{ComplexF64}` is passed as `double *`;
 * m, n, lda and the lo/hi ranges count ELEMENTS.  Device pointers must be 16-byte aligned.
 * Factor format as for Float64 with H_j = I - v_j v_j^H and complex alpha (the reference does not
 * phase-normalise R: alpha_j = -exp(i arg a_jj) ||a_j||, src:9,130; a zero pivot gives -||a_j||).
 *
 * dhqr_factor_c64   householder!(A, alpha) for ComplexF64 (src:113, 122-148, 171-213). Async.
 * dhqr_qr_c64       host-in / host-out qr!(A) (src:311-315). Synchronous.
 * dhqr_solve_c64    solve_householder!(b, H, alpha) (src:284-294 with the conj-dot of src:51-59):
 *                   db (m elements) is overwritten, x = db[0:n]. Async.
 * dhqr_ldiv_c64     host-in / host-out `H \ b` (src:317-321); hb is not modified. Synchronous.
 * dhqr_partialdot_c64 / _host_c64
 *                   partialdot(a, b, lo:hi, ComplexF64) = sum conj(a[i]) b[i] (src:51-59) -- the
 *                   function the reference's only known-answer test exercises
 *                   (test/partialdot.jl:12-20); hout[0] = re, hout[1] = im. Synchronous.
 *
 * dhqr_factor_c64_nb / dhqr_qr_c64_nb: the same with a panel width: nb = 0 unblocked, nb = DHQR_ZNB (64 complex columns)
 *                   BLOCKED: the panel's block reflector I - V T V^H is applied to the trailing matrix by the Float64
 *                   MFMA kernels through the real 2 x 2 embedding of the 64 complex reflectors (= 128 real columns);
 *                   identical factor format and values (to rounding). */
#define DHQR_ZNB 64
int32_t dhqr_factor_c64_nb(dhqr_ctx *ctx, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha, int32_t nb);
int32_t dhqr_qr_c64_nb(dhqr_ctx *ctx, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha, int32_t nb);
int32_t dhqr_factor_c64(dhqr_ctx *ctx, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha);
int32_t dhqr_qr_c64(dhqr_ctx *ctx, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha);
int32_t dhqr_solve_c64(dhqr_ctx *ctx, const double *dA, int64_t m, int64_t n, int64_t lda,
                       const double *dalpha, double *db);
int32_t dhqr_ldiv_c64(dhqr_ctx *ctx, const double *hA, int64_t m, int64_t n, int64_t lda,
                      const double *halpha, const double *hb, double *hx);
int32_t dhqr_partialdot_c64(dhqr_ctx *ctx, const double *da, const double *db, int64_t lo, int64_t hi,
                            double *hout);
int32_t dhqr_partialdot_host_c64(dhqr_ctx *ctx, const double *ha, const double *hb, int64_t lo,
                                 int64_t hi, double *hout);

/* ------------------------------------------------------------------ Q application / metric
 * dB (m x nrhs) <- Q' dB (trans = 1) or Q dB (trans = 0), Q = H_1 ... H_n from a factored dA.
 * Blocked compact-WY on MFMA (T is rebuilt per panel from V). No reference analogue beyond
 * src:215-242 (single vector); needed for the north-star metric ||A - QR|| / ||A||. Async. */
int32_t dhqr_apply_q_f64(dhqr_ctx *ctx, const double *dA, int64_t m, int64_t n, int64_t lda,
                         double *dB, int64_t nrhs, int64_t ldb, int32_t trans);

/* rel = ||dAorig - Q R||_F / ||dAorig||_F with Q,R taken from (dAfact, dalpha). dwork is an
 * m x n scratch matrix (leading dimension m) that receives Q*R.  Synchronous (host scalar). */
int32_t dhqr_residual_f64(dhqr_ctx *ctx, const double *dAfact, int64_t m, int64_t n, int64_t lda,
                          const double *dalpha, const double *dAorig, int64_t ldo, double *dwork,
                          double *hrel);

/* Building blocks of the residual for a column-split matrix (same maps as dhqr_fill_uniform_f64):
 * dW[:, jl] = [R; 0] column of local column jl (global index via the block-cyclic map; dalpha is
 * the GLOBAL alpha vector).  Async. */
int32_t dhqr_form_r0_f64(dhqr_ctx *ctx, const double *dA, int64_t m, int64_t cols, int64_t lda,
                         const double *dalpha, double *dW, int64_t ldw, int64_t colblock,
                         int32_t nranks, int32_t rank);
/* hout2[0] = sum (X-Y)^2, hout2[1] = sum X^2 over an m x n block. Synchronous (host scalars). */
int32_t dhqr_diff_norms_f64(dhqr_ctx *ctx, const double *dX, int64_t ldx, const double *dY, int64_t ldy,
                            int64_t m, int64_t n, double *hout2);

/* ------------------------------------------------------------------ panel level (multi-GPU)
 * The 1-D column-split driver (one process per GPU, torch.distributed/RCCL broadcast of the panel,
 * replacing the per-column @spawnat fan-out of src:141-143) is built from these two calls.
 *
 * Packed panel buffer ("VT"), doubles:  [ V : ldv x DHQR_NB | T : NB x NB | T' : NB x NB | alpha : NB | status : 16 ]
 *   ldv = dhqr_panel_ldv(rows); V is the panel's reflectors with the R part above the diagonal
 *   zeroed and columns >= ncols zero; T is the upper-triangular compact-WY factor
 *   (H_1...H_nb = I - V T V').  dhqr_panel_buffer_elems gives the total length. */
int64_t dhqr_panel_ldv(int64_t rows);
int64_t dhqr_panel_buffer_elems(int64_t rows);

/* Factor the rows x ncols panel dP in place (ncols <= DHQR_NB, rows >= ncols; row 0 of dP is the
 * panel's diagonal row) exactly like src:122-148 restricted to these columns, and emit dVT. Async. */
int32_t dhqr_panel_factor_f64(dhqr_ctx *ctx, double *dP, int64_t rows, int64_t ncols, int64_t ldp,
                              double *dVT);
/* Pack + T only, for a panel that is ALREADY factored (used when Q is re-applied, e.g. to form
 * Q*R for the residual on a column-split matrix). Async. */
int32_t dhqr_panel_pack_f64(dhqr_ctx *ctx, const double *dP, int64_t rows, int64_t ncols, int64_t ldp,
                            double *dVT);
/* dC (rows x ncols) <- (I - V T' V') dC  (trans = 1, the factorisation's trailing update,
 * src:198-213 blocked) or (I - V T V') dC (trans = 0). Async. */
int32_t dhqr_panel_apply_f64(dhqr_ctx *ctx, const double *dVT, int64_t rows, double *dC,
                             int64_t ncols, int64_t ldc, int32_t trans);

/* ------------------------------------------------------------------ multi-GPU: communicators
 * The multi-GPU drivers are SPMD programs: every rank (one per GPU) makes the same sequence of collective calls.
 * A communicator binds a rank to a context and to one of three transports (csrc/dhqr_comm.h):
 *   RCCL      ncclBroadcast / ncclAllReduce over xGMI (librccl.so is dlopen()ed on first use);
 *   LOCAL     peer copies + HIP events between the rank threads of ONE process (dhqr_mg_* when several ranks share
 *             a device, or DHQR_TRANSPORT=local);
 *   CALLBACK  the host layer supplies broadcast / all-reduce (MPI.jl, Distributed.jl, torch.distributed ...).
 * Multi-process use (one Julia worker / one torchrun rank per GPU; replaces the `@spawnat` fan-out of src:141-143
 * and the SharedArray alpha of src:301-304): rank 0 calls dhqr_comm_unique_id, the host layer ships the 128
 * bytes to the other processes, every process calls dhqr_comm_create_rank (collective: ncclCommInitRank). */
#define DHQR_UNIQUE_ID_BYTES 128
#define DHQR_COMM_SELF 0
#define DHQR_COMM_RCCL 1
#define DHQR_COMM_LOCAL 2
#define DHQR_COMM_CALLBACK 3
typedef struct dhqr_comm dhqr_comm;
/* dbuf: device pointer; must return 0 on success.  The library synchronises hip_stream before calling; the
 * callback must have completed (data visible to the device) when it returns. */
typedef int32_t (*dhqr_bcast_fn)(void *user, void *dbuf, int64_t bytes, int32_t root, void *hip_stream);
typedef int32_t (*dhqr_allreduce_fn)(void *user, void *dbuf, int64_t count_f64, void *hip_stream); /* in-place sum */
int32_t dhqr_comm_unique_id(void *id128);
int32_t dhqr_comm_create_rank(dhqr_comm **comm, dhqr_ctx *ctx, int32_t nranks, int32_t rank, const void *id128);
int32_t dhqr_comm_create_callbacks(dhqr_comm **comm, dhqr_ctx *ctx, int32_t nranks, int32_t rank,
                                   dhqr_bcast_fn bcast, dhqr_allreduce_fn allreduce, void *user);
int32_t dhqr_comm_destroy(dhqr_comm *comm);
int32_t dhqr_comm_info(dhqr_comm *comm, int32_t *kind, int32_t *nranks, int32_t *rank, int64_t *bytes_bcast);
/* Collectives this communicator has carried since it was created: out4 = {broadcasts, broadcast bytes, all-reduces, all-reduce
 * bytes} (the row-split lane's channel included).  All-reduces are counted as the drivers ISSUE them -- also at one rank,
 * where they move nothing -- so a single-GPU run reports the count and volume BASELINE.md section 2 asks for. */
int32_t dhqr_comm_counters(dhqr_comm *comm, int64_t *out4);
/* Device time this rank spent in its collectives (one hipEvent pair around every broadcast / all-reduce on the stream that
 * carries it, the wait for the peers included): out4 = {broadcast ms, broadcasts timed, all-reduce ms, all-reduces timed}
 * since the last call; on = 1 / 0 starts / stops collecting, -1 leaves it.  Synchronises the device.  What tells "the
 * broadcast is slow" from "the panel chain is slow" in a multi-GPU run (bench.py prints it per rank). */
int32_t dhqr_comm_timing(dhqr_comm *comm, int32_t on, double *out4);
/* RCCL transport: which algorithm large panel broadcasts use -- 0 ncclBroadcast (rings), 1 scatter + all-gather (the root
 * sends 1/P of the panel to each peer over its own xGMI link, then ncclAllGather) -- and what the timed trial at
 * communicator creation measured for a 16 MiB broadcast with each (ms; 0 when no trial ran: other transports, or
 * DHQR_BCAST=ring|sag set).  dhqr_mg_get_bcast_tuning: the same for rank 0 of a single-process handle. */
int32_t dhqr_comm_get_bcast_tuning(dhqr_comm *comm, int32_t *algo, double *ms_ring, double *ms_scatter_allgather);
/* Rank count RCCL itself reports (ncclCommCount) for the communicator's main channel and for the row-split lane's second
 * channel; 0 = that channel is not an RCCL communicator (other transport, single rank, DHQR_LANE_CHANNEL=0). */
int32_t dhqr_comm_rccl_nranks(dhqr_comm *comm, int32_t *main_channel, int32_t *lane_channel);

/* ------------------------------------------------------------------ multi-GPU: 1-D column split (SPMD, collective)
 * householder!(A::DArray, alpha) (src:115-120).  Layout: BLOCK-CYCLIC columns, block = DHQR_CS_BLOCK (a pair of
 * panels): rank r holds the global column blocks r, r+P, r+2P, ... contiguously (dA: m x dhqr_cs_local_cols(n,P,r), leading dimension lda);
 * dalpha (n) is replicated.  Per panel ONE broadcast of its (V, T, T', alpha) operands replaces the reference's
 * per-column fan-out; trailing updates apply two panels per pass (K = 256 MFMA update) under look-ahead.
 * Every rank of the communicator must make the call.  Synchronous on return.
 *   dhqr_cs_factor_f64     in-place factorisation of the local blocks, alpha on every rank.
 *   dhqr_cs_residual_f64   ||A - QR||_F / ||A||_F with A regenerated from `seed` (dhqr_cs_fill_uniform_f64);
 *                          dW, dA0: m x local_cols scratch matrices (leading dimension m).
 *   dhqr_cs_solve_f64      solve_householder!(b, H, alpha) (src:226-282): db (m, identical on every rank) is
 *                          overwritten, x = db[0:n] on every rank; dwork: m + 128 doubles.  One all-reduce of the
 *                          partial dots per block replaces sum(fetch.(futures)) (src:262-266).
 *   dhqr_cs_load/store_contiguous_f64   convert between the reference's DArray layout (ONE contiguous column
 *                          block per process, dhqr_cs_contiguous_range = DistributedArrays' default split) and the
 *                          block-cyclic layout; dstage: m x max(n/P + 1, 128) doubles.
 *   dhqr_cs_qr_darray_f64  qr!(A::DArray) for one process: host block in, factored host block + alpha out.
 *   dhqr_cs_ldiv_darray_f64  `qrA \ b` for a DArray factorisation (src:317-321 -> src:226-230, 256-270; the call
 *                          test/runtests.jl:77-78 makes): this process's FACTORED contiguous host block, the replicated
 *                          alpha and b (m, the same on every process) in, x (n) out on every process; nothing on the
 *                          host is modified (the reference copies b into a SharedArray, src:318). */
int64_t dhqr_cs_local_cols(int64_t n, int32_t nranks, int32_t rank);
void dhqr_cs_contiguous_range(int64_t n, int32_t nranks, int32_t rank, int64_t *lo, int64_t *hi);
int32_t dhqr_cs_fill_uniform_f64(dhqr_comm *comm, double *dA, int64_t m, int64_t n, int64_t lda, uint64_t seed);
int32_t dhqr_cs_factor_f64(dhqr_comm *comm, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha);
int32_t dhqr_cs_residual_f64(dhqr_comm *comm, const double *dA, int64_t m, int64_t n, int64_t lda,
                             const double *dalpha, uint64_t seed, double *dW, double *dA0, double *hrel);
int32_t dhqr_cs_solve_f64(dhqr_comm *comm, const double *dA, int64_t m, int64_t n, int64_t lda,
                          const double *dalpha, double *db, double *dwork);
int32_t dhqr_cs_load_contiguous_f64(dhqr_comm *comm, double *dA, int64_t m, int64_t n, int64_t lda,
                                    const double *dBlock, int64_t ldb, double *dstage);
int32_t dhqr_cs_store_contiguous_f64(dhqr_comm *comm, const double *dA, int64_t m, int64_t n, int64_t lda,
                                     double *dBlock, int64_t ldb, double *dstage);
int32_t dhqr_cs_qr_darray_f64(dhqr_comm *comm, double *hBlock, int64_t m, int64_t n, int64_t ldb, double *halpha);
int32_t dhqr_cs_ldiv_darray_f64(dhqr_comm *comm, const double *hBlock, int64_t m, int64_t n, int64_t ldb,
                                const double *halpha, const double *hb, double *hx);

/* ------------------------------------------------------------------ multi-GPU: ComplexF64 column split
 * The reference's householder! is generic over the element type (src:215-294; test/runtests.jl:42-63 runs ComplexF64).
 * Layout: cyclic blocks of 64 complex columns (one panel): rank r holds the global panels r, r+P, r+2P, ...
 * contiguously (dA: m x dhqr_cs_local_cols_c64(n,P,r) complex, interleaved re/im, leading dimension lda complex
 * elements); dalpha (n complex) is replicated.  Per panel ONE broadcast of (alpha, T, T', the factored panel as
 * rows x 64 complex) replaces the per-column fan-out (src:141-143); every rank forms the real 2 x 2 embedding of the 64
 * reflectors itself (twice the bytes, which therefore do not travel); the owner of the next panel looks ahead on a
 * high-priority stream.
 * Every rank of the communicator must make the call.  Asynchronous on the context's stream like dhqr_factor_c64_nb.
 *   dhqr_cs_solve_c64      solve_householder!(b, H, alpha) (src:226-282) on the cyclic layout: db (m complex, identical on
 *                          every rank) is overwritten, x = db[0:n] on every rank; dwork: dhqr_cs_solve_work_c64(m, P)
 *                          complex.  b is carried in double-double (the reference's acceptance statistic sees the
 *                          solve's rounding).  Q'b: two broadcasts of b's tail (high / low parts) per panel; back
 *                          substitution: one all-reduce that gathers the ranks' 64 partial dots
 *                          (sum(fetch.(futures)), src:262-266) + one broadcast of the solved block per panel.
 *   dhqr_cs_qr_darray_c64  qr!(A::DArray{ComplexF64}) for one process: its CONTIGUOUS host column block
 *                          (dhqr_cs_contiguous_range) in, factored block + alpha out; synchronous.
 *   dhqr_cs_ldiv_darray_c64  `qrA \ b` for a DArray{ComplexF64} factorisation (test/runtests.jl:77-78 with T = ComplexF64):
 *                          factored contiguous host block + alpha + b in, x (n complex) out on every process; synchronous.
 *   dhqr_mg_qr_c64   qr!(A; ndev) for a ComplexF64 host matrix (host in / host out).
 *   dhqr_mg_ldiv_c64 `H \ b` for the host-format result over the same devices (dhqr_cs_solve_c64 inside). */
int64_t dhqr_cs_local_cols_c64(int64_t n, int32_t nranks, int32_t rank);
int32_t dhqr_cs_factor_c64(dhqr_comm *comm, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha);
int64_t dhqr_cs_solve_work_c64(int64_t m, int32_t nranks);
int32_t dhqr_cs_solve_c64(dhqr_comm *comm, const double *dA, int64_t m, int64_t n, int64_t lda,
                          const double *dalpha, double *db, double *dwork);
int32_t dhqr_cs_qr_darray_c64(dhqr_comm *comm, double *hBlock, int64_t m, int64_t n, int64_t ldb, double *halpha);
int32_t dhqr_cs_ldiv_darray_c64(dhqr_comm *comm, const double *hBlock, int64_t m, int64_t n, int64_t ldb,
                                const double *halpha, const double *hb, double *hx);

/* ------------------------------------------------------------------ multi-GPU: single-process handle
 * One host process drives `ndev` GPUs (devices[i] = HIP device of rank i; NULL = 0..ndev-1): one context, one
 * communicator rank and one host thread per device run the SPMD drivers above.  Transport: RCCL
 * (ncclCommInitAll) when the devices are distinct, peer copies otherwise or on DHQR_TRANSPORT=local.
 * What `qr!(A; ndev = 8)` of the Julia module and `python bench.py --gpus N` bind.
 *   dhqr_mg_qr_f64 / dhqr_mg_ldiv_f64    host-in / host-out drop-ins for qr!(A) and `H \ b` (src:311-321).
 *   dhqr_mg_alloc/fill/factor/residual   device-resident path (inputs generated in HBM; what bench.py times).
 *   dhqr_mg_solve_f64                    `\` with the factored matrix resident in the handle. */
typedef struct dhqr_mg dhqr_mg;
int32_t dhqr_mg_create(dhqr_mg **mg, const int32_t *devices, int32_t ndev);
int32_t dhqr_mg_destroy(dhqr_mg *mg);
int32_t dhqr_mg_info(dhqr_mg *mg, int32_t *ndev, int32_t *transport, int64_t *m, int64_t *n);
int32_t dhqr_mg_get_bcast_tuning(dhqr_mg *mg, int32_t *algo, double *ms_ring, double *ms_scatter_allgather);
int32_t dhqr_mg_rccl_nranks(dhqr_mg *mg, int32_t *main_channel, int32_t *lane_channel); /* dhqr_comm_rccl_nranks of rank 0 */
int32_t dhqr_mg_comm_counters(dhqr_mg *mg, int32_t rank, int64_t *out4);                 /* dhqr_comm_counters of one rank */
int32_t dhqr_mg_comm_timing(dhqr_mg *mg, int32_t rank, int32_t on, double *out4);          /* dhqr_comm_timing of one rank */
int32_t dhqr_mg_alloc_f64(dhqr_mg *mg, int64_t m, int64_t n);
int32_t dhqr_mg_fill_uniform_f64(dhqr_mg *mg, uint64_t seed);
int32_t dhqr_mg_factor_f64(dhqr_mg *mg);
int32_t dhqr_mg_residual_f64(dhqr_mg *mg, uint64_t seed, double *hrel);
int32_t dhqr_mg_upload_f64(dhqr_mg *mg, const double *hA, int64_t lda, const double *halpha);
int32_t dhqr_mg_download_f64(dhqr_mg *mg, double *hA, int64_t lda, double *halpha);
int32_t dhqr_mg_qr_f64(dhqr_mg *mg, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha);
int32_t dhqr_mg_qr_c64(dhqr_mg *mg, double *hA, int64_t m, int64_t n, int64_t lda, double *halpha);
int32_t dhqr_mg_ldiv_c64(dhqr_mg *mg, const double *hA, int64_t m, int64_t n, int64_t lda, const double *halpha,
                         const double *hb, double *hx);
int32_t dhqr_mg_solve_f64(dhqr_mg *mg, const double *hb, double *hx);
int32_t dhqr_mg_ldiv_f64(dhqr_mg *mg, const double *hA, int64_t m, int64_t n, int64_t lda, const double *halpha,
                         const double *hb, double *hx);
int32_t dhqr_mg_set_profiling(dhqr_mg *mg, int32_t on);
int32_t dhqr_mg_reset_stats(dhqr_mg *mg);
int32_t dhqr_mg_get_stats(dhqr_mg *mg, int32_t rank, dhqr_stats *out, int64_t *n_fast, int64_t *n_fallback,
                          int64_t *bytes_bcast);

/* ------------------------------------------------------------------ multi-GPU: row split (BASELINE configs[4])
 * Tall-skinny matrices with the ROWS distributed (the reference cannot: `@assert rowrange == 1:size(A,1)`, src:33;
 * its per-column norm / partial dots, src:129,208, would need one cross-rank reduction per column).  SPMD and
 * collective like the column split.  Layout: rank r holds the rows [row0, row0 + mloc) of dhqr_rs_row_range
 * (128-row aligned slabs, so a panel's diagonal block lives on exactly one rank) as an mloc x n column-major
 * block; alpha (n) replicated.  Per 128-column panel: all-reduce of the 128 x 128 Gram matrices, broadcast of the
 * top-block result from the rank holding the diagonal rows, all-reduce of the V'C partial dots (128 x ncols) --
 * the "RCCL all-reduce of the cross-partition partial dots".  Panels are verified on the device; a rejected or
 * partial panel is redone column by column across the ranks (reference algorithm, one small all-reduce + broadcast
 * per column).  Same factor format, distributed by rows.  Synchronous on return.
 *   dhqr_rs_solve_f64: db = this rank's rows of b (overwritten), dx (n) <- x on every rank (src:215-254). */
void dhqr_rs_row_range(int64_t m, int32_t nranks, int32_t rank, int64_t *row0, int64_t *mloc);
int32_t dhqr_rs_fill_uniform_f64(dhqr_comm *comm, double *dA, int64_t m, int64_t n, int64_t lda, uint64_t seed);
int32_t dhqr_rs_factor_f64(dhqr_comm *comm, double *dA, int64_t m, int64_t n, int64_t lda, double *dalpha);
int32_t dhqr_rs_residual_f64(dhqr_comm *comm, const double *dA, int64_t m, int64_t n, int64_t lda,
                             const double *dalpha, uint64_t seed, double *dB, double *dA0, double *hrel);
int32_t dhqr_rs_solve_f64(dhqr_comm *comm, const double *dA, int64_t m, int64_t n, int64_t lda,
                          const double *dalpha, double *db, double *dx);
/* the same through the single-process handle (one host thread per device) */
int32_t dhqr_mg_rs_alloc_f64(dhqr_mg *mg, int64_t m, int64_t n);
int32_t dhqr_mg_rs_fill_uniform_f64(dhqr_mg *mg, uint64_t seed);
int32_t dhqr_mg_rs_factor_f64(dhqr_mg *mg);
int32_t dhqr_mg_rs_residual_f64(dhqr_mg *mg, uint64_t seed, double *hrel);
int32_t dhqr_mg_rs_transfer_f64(dhqr_mg *mg, double *hA, int64_t lda, double *halpha, int32_t upload);
int32_t dhqr_mg_rs_solve_f64(dhqr_mg *mg, const double *hb, double *hx);

/* Micro-benchmarks and the MFMA layout probe are NOT part of this library: include/dhqr_bench.h, libdhqr_bench.so. */

#ifdef __cplusplus
}
#endif
#endif /* DHQR_H */
