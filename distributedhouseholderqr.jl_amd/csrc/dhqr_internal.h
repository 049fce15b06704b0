// dhqr_internal.h -- what the translation units of libdhqr.so share: the context, the error / launch-check macros and the
// few host helpers that cross a unit boundary.  Units (csrc/build.sh compiles them side by side):
//   dhqr_api.hip        context, blocked drivers (single GPU, column split, row split, one-process multi-GPU), solve,
//                       ComplexF64, host I/O, every extern "C" entry point of include/dhqr.h
//   dhqr_unblocked.hip  the nb = 0 path: dhqr_rank1.h's K-reflectors-per-pass kernels (84 % of the library's device code)
//                       and their driver factor_unblocked_cols
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dhqr.h"
#include "dhqr_common.h"

// the calling thread's last error text (dhqr_last_error) and the one place that writes it -- defined in dhqr_api.hip
int32_t set_err(int32_t code, const char *fmt, ...);
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
  hipStream_t hi2 = nullptr;     // its side stream: products that need a panel's V but not its T (V_a' C_b of the pair's second
                                 // panel, the pair / quad cross terms) run here beside the panel's verification and commit
  int lane_side = 1;             // ... DHQR_LANE_SIDE=0: everything on the one lane stream.  Single rank only: a device has
                                 // GPU_MAX_HW_QUEUES = 4 hardware queues and streams beyond them share one -- with the
                                 // communication stream of P > 1 (and RCCL's own) a fifth stream serialises something
                                 // (measured with rank threads sharing one GPU: 32768^2 at 2 ranks 904 -> 971 ms)
  int hi_priority = 0;
  int tn_spare = 32;                // CUs a wide k_gemm_tn2 launch on a small trailing matrix leaves to the lane (wide_slots) ...
  int64_t tn_spare_cols = 16384;    // ... "small": at most this many trailing columns (DHQR_TUNE tn_spare, tn_spare_cols)
  int tn_model_min_tiles = 32;   // (r5: 32, with the direct-load kernel; 128 before: 8192^2 29.9 -> 29.1 ms, 16384^2 123.6 -> 120.2, 32768^2 783.7 -> 775.7, profiles/r05_ab_thresholds.txt) wide k_gemm_tn2 launches of at least this many column tiles: split-K factor from the round / partial-traffic estimate (below, the lane is the critical path and
                                 // prefers many short workgroups: a k_gemm_tn2 workgroup leaves no room for a lane kernel on its CU)
  int rankk_wgs = 256;           // ... bulk workgroups of 1024 threads resident at once (CU count; DHQR_RANKK_WGS)
  int rankk = 5;                 // unblocked path: reflectors applied per pass over the trailing columns (DHQR_RANKK=1..5; beyond 3 the further ones are held in LDS)
  int nn_chunk_tiles = 48;       // ... of at least this many 128-wide tiles each (DHQR_NN_CHUNK_TILES: the CPU emulator's tests set 1)
  int nn_split = 4;              // wide subtraction launches in up to this many chunks of columns (or rows) (nn_chunks; DHQR_NN_SPLIT=1: one launch)
  int rankk_pipe = 1;            // k_rankk_fused: the lead as K pipelined workgroups where the lead bounds the launch (launch_rankk; DHQR_RANKK_PIPE=0 never, 2 always)
  int rankk_max_min_cols = 4096; // ... while more than this many columns are left (DHQR_RANKK_MAX_MIN_COLS: below, a launch is bound by its lead's chain, which grows with K)
  int rankk_unfit = 0;           // set by launch_rankk if it was asked for a K its ladder step cannot hold (a driver bug: reported, never silent)
  int rankk_max = 8;             // nb = 0, default DHQR_RANKK=5: up to this many reflectors per pass where the CU can hold them (columns of <= 6144 rows; DHQR_RANKK_MAX=5: never more than 5)
  int rankk_xtall = 5;           // nb = 0, columns of 16384 < rows <= 32768: reflectors per pass (k_rankk_xtall; DHQR_RANKK_XTALL=1: one per launch)
  int rankk_tall = 5;            // nb = 0, columns of 8192 < rows <= 16384: reflectors per pass (k_rankk_tall; DHQR_RANKK_TALL=1: one per launch)
  int ncu = 256;                 // compute units of the device
  int spare_cus = 0;             // CUs the persistent wide k_gemm_tn2 launches leave free for the look-ahead lane's
                                 // single-workgroup kernels and for RCCL's kernels (DHQR_SPARE_CUS; multiple of 8: one per XCD).
                                 // Default 0 on one GPU (measured: 8 spare CUs cost the wide kernels more than the lane gains);
                                 // 8 when the context is bound to an RCCL communicator of more than one rank: the panel
                                 // broadcast sits on the critical chain there and RCCL's kernels need CUs of their own while a
                                 // persistent launch holds every CU it was given (comm_bind_rccl)
  bool spare_cus_set = false;    // DHQR_SPARE_CUS given: never overridden
  int quad = 1;                  // P == 1: two consecutive pairs applied in ONE K = 512 pass (quad_apply; DHQR_QUAD=0: pairs only)
  int64_t quad_min_cols = 10240;  // ... while at least this many columns lie to the right of the quad (DHQR_QUAD_MIN_COLS)
  struct WS { Buf w1, w1r, w2; } ws[3];  // [0] wide trailing update, [1] panel / narrow updates, [2] the lane's side stream (hi2)
  int cur_ws = 0;
  bool lookahead = true;
  Buf vbuf, vt, vts, spart, spart2, sfull, scratch, pbuf;  // spart2: split-K partials of the cross terms on the side stream
  Buf zsolve_lo;         // low parts of the double-double right-hand side of the ComplexF64 solve
  // the solve of dhqr_qtb.h: T' of every panel, Gram matrices, their slab partials, (partial dots | w | ints)
  Buf sv_T, sv_S, sv_part, sv_small;
  std::vector<int> sv_units;                 // host copy of the Gram pre-pass unit table, valid for (sv_m, sv_n)
  int64_t sv_m = -1, sv_n = -1, sv_rps = 0;
  const int *sv_units_dev = nullptr;         // where the table was uploaded (nullptr: not yet / shape changed)
  // kept T factors: a blocked single-GPU dhqr_factor_f64 leaves T_k' of every panel (and a copy of alpha) in the context;
  // dhqr_solve_f64 on the same (dA, m, n, lda) whose alpha still equals that copy (checked on the device) skips its Gram /
  // T' pre-pass.  tt_keep: where the panel being factored stores its T' (nullptr: nowhere).
  Buf tc_T, tc_alpha;
  double *tt_keep = nullptr, *tc_base = nullptr;
  const double *tc_A = nullptr;
  int64_t tc_m = 0, tc_n = 0, tc_lda = 0;
  bool tc_valid = false;
  int keep_t = 1;        // DHQR_KEEP_T=0: never keep / use them
  int solve_pipe = 1;    // DHQR_SOLVE_PIPE: 1 the solve of dhqr_qtb.h (persistent Q'b kernel when this context has its device to itself),
                         // 2 the same without the persistent kernel (one launch per panel step), 3 the persistent kernel whatever
                         // else lives on the device (tests), 0 the round-1 solve (blocked apply on the MFMA kernels + 64-row back
                         // substitution: no inter-workgroup waits at all)
  int qtb_vec = -1;      // DHQR_QTB_VEC=1/2: rows per lane of k_qtb_step (-1: by the matrix height)
  // k_qtb_persist's workgroups wait for each other in both directions; should one of its bounded waits expire (workgroups
  // that could not all become resident: another process or stream holding compute units), the solve is REPEATED with one
  // launch per panel step by the first synchronising entry point that finds the error word (pipe_error_check) instead of
  // being reported: b is saved before the persistent launch, the arguments are remembered here.
  struct SolveRetry {
    bool valid = false;
    const double *A = nullptr, *alpha = nullptr;
    double *b = nullptr;
    int64_t m = 0, n = 0, lda = 0;
  } retry;
  Buf sv_bkp;            // b as it was handed to the last solve that took the persistent kernel
  int64_t n_solve_retry = 0;  // solves repeated that way (dhqr_get_solve_retries)
  int gram_strips = 1;   // the panel chain's Gram products as four 32-row strips (gram128; DHQR_TUNE gram_strips=0: one tile)
  int small_spin_limit = 1 << 20;  // polls before a wait of the flag form gives up (dhqr_small.h: SMQ_SPIN_LIMIT)
  int partial_unblocked = 1;  // a panel that is not R-first eligible, of 257 .. partial_unblocked_max_rows rows: the K-reflector passes (1) or one launch per column (0)
  int64_t partial_unblocked_max_rows = 4608;
  int short_panel_small = 1;  // blocked drivers: a panel of at most 256 rows in ONE launch of the small route's kernel (factor_panel_v2)
  int small_flags = 1;   // small route above 128 rows: LDS flags instead of a barrier per column (dhqr_small.h)
  int tn2_rgroups = 1;   // row groups of a stream-K k_gemm_tn2 launch (dhqr_gemm.h: tn2_sk_group_of): 1 / 2 / 4 / 8, 0 = by height
  int64_t tn2_rg8_rows = 20480, tn2_rg4_rows = 10240, tn2_rg2_rows = 5120;
  int fuse_fix = 1;      // k_recon_fix in the epilogue of V = P M^{-1} (mul128; DHQR_TUNE fuse_fix=0: its own launch)
  int commit_off = -1;   // an accepted panel's commit (12 us of copies into the matrix) leaves the lane: -1 (default) with more
                         // than one rank, where it runs on the communication stream BEHIND the panel's broadcast; 0 never;
                         // 1 also at one rank, on a stream of its own -- measured there (profiles/r06_ab_chain.txt): a
                         // fourth busy stream beside wide / lane / side shares a hardware queue and the factorisation
                         // takes 1.9 x as long at 8192^2, 1.13 x at 32768^2 (DHQR_TUNE commit_off)
  hipStream_t cstream = nullptr;  // ... that stream (created on first use)
  hipEvent_t ev_commit[2] = {nullptr, nullptr};  // behind the off-lane commit of the last panel of each parity
  int small_route = 1;   // matrices that fit the registers of one compute unit: ONE single-workgroup launch per qr! / per
                         // `\` (dhqr_small.h; DHQR_SMALL=0 or dhqr_set_small_route(ctx, 0): the general drivers)
  double *small_pin = nullptr;  // pinned host staging of the host-array entry points on that route: the kernels read and
  size_t small_pin_cap = 0;     // write it across PCIe themselves (no hipMemcpy on the path); doubles
  Buf small_dev;                // device copy of a host factor inside k_small_ldiv (256 x 256)
  unsigned long long small_epoch = 0;  // launches that signalled their end through the pinned word behind small_pin
  bool coop = false;     // the device runs cooperative (all-resident) launches: false on the CPU emulator
  Buf host_mat;          // device copy of the caller's HOST matrix (+ alpha) of dhqr_qr_f64, kept between calls
  int pair = 1;                  // 1: wide updates apply two panels per pass (DHQR_PAIR=0 disables)
  int64_t pair_min_n = 4096;     // below this the longer look-ahead lane of the pair driver costs more than it saves (r2: 12288, profiles/r02_ab_pair_tail_and_threshold.txt; r5, with the direct-load GEMMs: 4096^2 12.11 -> 11.84 ms, 8192^2 31.97 -> 31.39, 12288^2 68.18 -> 67.71, profiles/r05_ab_thresholds.txt)
  int panel_impl = 3;  // 3: R-first (CholeskyQR + reconstruction, dhqr_recon.h) with fallback to 2;
                       // 2: row-split sub-panel kernels (dhqr_panel.h); 1: one workgroup per column
  int zpipe = 1;         // ComplexF64 panels of <= 128 columns and <= 8192 rows in one column-pipelined launch (k_zpanel_pipe; DHQR_ZPIPE=0: one launch per column)
  hipEvent_t zev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // look-ahead of the blocked ComplexF64 driver
  int *zflags = nullptr; // its 128 ready flags (device), never reset: a flag holds the number of the launch that set it
  int zepoch = 0;
  int *hflag = nullptr;  // pinned host copy of the device status block
  int *dstat = nullptr;  // device status block (ints): [0] first rejected panel of the running factorisation
                         // (INT_MAX: none), [1] Cholesky breakdown flag of the panel in flight
  int epoch = -1;        // >= 0 while an asynchronous factorisation is enqueued: matrix-writing launches carry
                         // (dstat, epoch) and are no-ops once a panel <= epoch was rejected
  Buf rbuf;              // R1, -R1^{-1}, R, Rref, -M^{-1}, alpha_tmp
  Buf tsq;               // R factors, reflectors and Q of the TSQR tree (dhqr_tsqr.h)
  int tsqr_rung = -1;    // rejected panels try TSQR-HR before the column-by-column kernels: 1 yes, 0 no, -1 (default)
                         // only where the last rung costs collectives per column (row split over more than one rank):
                         // on one GPU the column kernels redo a panel in ~1 ms, the tree takes ~7 ms at 32768 rows
  int n_tsqr = 0;        // panels whose R came from the TSQR tree (dhqr_get_panel_counters: counted as fast)
  int64_t n_fast = 0, n_fallback = 0;
  int cholqr_passes = 1;  // where the fast path gets R from: 1 Gram/Cholesky, 2 CholeskyQR2, 3 TSQR tree (DHQR_TSQR=1)
  double recon_tol = 2e-12;  // accepted deviation of ||v_j||^2 from 2 before falling back
  int ib = 64;  // DHQR_IB (dhqr_panel.h): sub-panel width of the column-by-column panel kernels
  struct CsState *cs = nullptr;  // streams / events / group buffers of the blocked driver (dhqr_dist.h)
  struct RsState *rs = nullptr;  // events / group ring of the row-split driver (dhqr_rowsplit.h)
  // host-in / host-out drop-in (dhqr_hostio.h): the blocked driver reports every committed panel (index, event) to this
  // hook so that the finished column block can travel to the host while later panels are factored
  int32_t (*panel_hook)(void *, int64_t, hipEvent_t) = nullptr;
  void *panel_hook_arg = nullptr;
  struct HostIo *hio = nullptr;
  int64_t n_resume = 0;  // passes of the blocked driver that resumed after a rejected panel
  // profiling
  struct Ev { hipEvent_t a, b; int cat; int start_from = -1; };  // start_from >= 0: the section starts at the END event of that entry (prof_switch)
  std::vector<Ev> evs;
  size_t ev_used = 0;
  dhqr_stats st;
};


// ---- host helpers defined in dhqr_api.hip
int32_t ensure(dhqr_ctx *c, Buf &b, size_t need);   // grow a workspace (synchronises the device when it reallocates)
bool tune_get(const char *key, long long *out);      // DHQR_TUNE="key=value,..."
static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
int32_t prof_begin(dhqr_ctx *c, int cat);            // one hipEvent pair per timed launch group (dhqr_get_stats)
int32_t prof_end(dhqr_ctx *c);
int32_t prof_switch(dhqr_ctx *c, int cat);
// ---- dhqr_unblocked.hip: householder!(A, alpha) column by column, K reflectors per pass (src:122-148,198-213) on the
// columns of a rows x ncols block whose row 0 is the diagonal row of column 0; launch groups are timed under `cat`
#define DHQR_RK_KMAX 8   // most reflectors a pass of the unblocked kernels applies (dhqr_rank1.h)
int32_t factor_unblocked_cols(dhqr_ctx *c, double *P, int64_t rows, int64_t ncols, int64_t ldp, double *alpha, int cat);
