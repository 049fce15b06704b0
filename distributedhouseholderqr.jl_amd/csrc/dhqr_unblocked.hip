// dhqr_unblocked.hip -- the nb = 0 path of libdhqr.so (BASELINE configs[1]): the reference's column-by-column algorithm
// (src:122-148, 198-213) on the HBM-bound kernels of dhqr_rank1.h, K reflectors per pass over the trailing columns.  Its
// own translation unit: the instantiation ladder of k_rankk_fused / _tall / _xtall and their lead kernels is 84 % of the
// library's device code, and compiles beside dhqr_api.hip instead of in front of it.
#include "dhqr_internal.h"
#include "dhqr_rank1.h"

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

// K steps per pass over the columns >= c0 (k_rankk_fused): the `kold` reflectors in `vold` are applied, the lead
// workgroup builds the next K into `vnew`
template <int VEC, int K>
static void launch_rankk(dhqr_ctx *c, double *P, int64_t ldp, int64_t rows, int64_t ncols, int64_t c0, int64_t jlo,
                         int kold, const double *vold, double *vnew, int64_t vlen, double *alpha) {
  const int64_t rtop = (VEC == 2) ? (jlo & ~(int64_t)1) : jlo;
  const int64_t cov = rows - rtop;
  // lead workgroup + persistent bulk workgroups: as many as run at once (one 1024-thread workgroup per CU), less the
  // lead's place
  const int64_t nbulk = (kold == 0) ? 0 : std::max<int64_t>(0, ncols - c0 - K);
  // The lead as K pipelined workgroups (rankk_lead_pipe) where the LEAD bounds the launch: few columns left for the bulk
  // (its traffic takes less than the one-workgroup lead's ~105 us x cov / 8192 below ~3000 columns, whatever cov) and
  // columns long enough for K - 1 hand-overs (~8 us each) to cost less than the one workgroup's K (K + 1) / 2 extra applies.
  // 8192 x 2048: 43 -> 25 ms; on squares K - 1 fewer bulk workgroups cost 1.6 % while the bulk bounds the launch, hence
  // not everywhere (profiles/r03_unblocked_pipelined_lead.txt).  DHQR_RANKK_PIPE=0: never, 2: always.
  const bool pipe = c->rankk_pipe == 2 || (c->rankk_pipe == 1 && nbulk <= 3072 && cov >= 2048);
  const int nlead = pipe ? K : 1;
  const int epoch = ++c->zepoch;  // the launch's number in the flags (a value, not an expression in the launch's argument list)
  // (workgroups per CU of the persistent bulk: as many as the threads allow unless the LDS-resident reflectors -- the lead's
  // slots, or more than two of the pass's own -- leave room for one only; a K the ladder step cannot hold is never asked for:
  // rankk_fit, factor_unblocked_cols)
#define DHQR_RK(T_, E_)                                                                                  \
  do {                                                                                                   \
    if constexpr (K <= rankk_fit(T_, E_))                                                                \
      hipLaunchKernelGGL((k_rankk_fused<T_, E_, VEC, K>),                                                \
                         dim3((unsigned)(nlead + std::min<int64_t>(nbulk, std::max<int64_t>(1, (int64_t)c->rankk_wgs * ((rankk_lead_slots(T_, E_, K) > 0 || (K - rankk_kr(E_, K)) * T_ * E_ * 8 > 80 * 1024) ? 1 : 1024 / T_) - nlead)))), \
                         dim3(T_), 0, c->stream, P, ldp, rows, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, \
                         pipe ? c->zflags : (int *)nullptr, epoch);                                      \
    else                                                                                                 \
      c->rankk_unfit = K;                                                                                \
  } while (0)
#define DHQR_RKT(E_)                                                                                     \
  hipLaunchKernelGGL((k_rankk_tall<512, E_, VEC, K>),                                                     \
                     dim3((unsigned)(K + std::min<int64_t>(nbulk, std::max<int64_t>(1, (int64_t)c->rankk_wgs - K)))), dim3(512), 0, c->stream, P, ldp, \
                     rows, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, c->zflags, epoch)
#define DHQR_RKX(E_)                                                                                     \
  hipLaunchKernelGGL((k_rankk_xtall<512, E_, VEC, K>),                                                    \
                     dim3((unsigned)(K + std::min<int64_t>(nbulk, std::max<int64_t>(1, (int64_t)c->rankk_wgs - K)))), dim3(512), 0, c->stream, P, ldp, \
                     rows, ncols, c0, rtop, kold, vold, vnew, vlen, alpha, c->zflags, epoch)
  // columns of 16384 < rows <= 32768 (DHQR_RANKK_XTALL >= 2): one column per workgroup in registers, reflectors streamed
  if (cov > 512 * 48) { DHQR_RKX(64); return; }
  if (cov > 512 * 32) { DHQR_RKX(48); return; }
  // columns of 8192 < rows <= 16384 (factor_unblocked_cols sends them here when DHQR_RANKK_TALL >= 2)
  if (cov > 512 * 24) { DHQR_RKT(32); return; }
  if (cov > 1024 * 8) { DHQR_RKT(24); return; }
  // six per pass for columns of 6145 ... 8192 rows: 16 elements per thread, four reflectors in registers (rankk_kr)
  if constexpr (K == 6 && VEC == 2) {
    if (cov > 448 * 16) { DHQR_RK(512, 16); return; }
    if (cov > 768 * 8) { DHQR_RK(448, 16); return; }
  }
  if (cov <= 256 * 2) DHQR_RK(256, 2);
  else if (cov <= 256 * 4) DHQR_RK(256, 4);
  else if (cov <= 256 * 8) DHQR_RK(256, 8);
  else if (cov <= 384 * 8) DHQR_RK(384, 8);
  else if (cov <= 512 * 8) DHQR_RK(512, 8);
  else if (cov <= 640 * 8) DHQR_RK(640, 8);
  else if (cov <= 768 * 8) DHQR_RK(768, 8);
  else if (cov <= 896 * 8) DHQR_RK(896, 8);
  else DHQR_RK(1024, 8);
#undef DHQR_RK
#undef DHQR_RKT
#undef DHQR_RKX
}
static void launch_rankk(dhqr_ctx *c, bool vec, int K, double *P, int64_t ldp, int64_t rows, int64_t ncols, int64_t c0,
                         int64_t jlo, int kold, const double *vold, double *vnew, int64_t vlen, double *alpha) {
#define DHQR_RKK(K_)                                                                              \
  (vec ? launch_rankk<2, K_>(c, P, ldp, rows, ncols, c0, jlo, kold, vold, vnew, vlen, alpha)      \
       : launch_rankk<1, K_>(c, P, ldp, rows, ncols, c0, jlo, kold, vold, vnew, vlen, alpha))
  if (K == 2) DHQR_RKK(2);
  else if (K == 3) DHQR_RKK(3);
  else if (K == 4) DHQR_RKK(4);
  else if (K == 5) DHQR_RKK(5);
  else if (K == 6) DHQR_RKK(6);
  else if (K == 7) DHQR_RKK(7);
  else DHQR_RKK(8);
#undef DHQR_RKK
}

// most reflectors per pass the k_rankk_fused instantiation launch_rankk picks for columns of `cov` rows can hold
static inline int rankk_fit_rows(int64_t cov, bool vec) {
  if (cov <= 256 * 2) return rankk_fit(256, 2);
  if (cov <= 256 * 4) return rankk_fit(256, 4);
  if (cov <= 256 * 8) return rankk_fit(256, 8);
  if (cov <= 384 * 8) return rankk_fit(384, 8);
  if (cov <= 512 * 8) return rankk_fit(512, 8);
  if (cov <= 640 * 8) return rankk_fit(640, 8);
  if (cov <= 768 * 8) return rankk_fit(768, 8);
  return vec ? 6 : 5;  // <= 8192 rows: the 16-elements-per-thread instantiations (launch_rankk; 16-byte path only: the scalar path spills 120 registers there) for six, 896 / 1024 x 8 for up to five
}
int32_t factor_unblocked_cols(dhqr_ctx *c, double *P, int64_t rows, int64_t ncols, int64_t ldp, double *alpha, int cat) {
  const int K = c->rankk;  // reflectors per pass over the trailing columns (1: one launch per reflector)
  // ... and, at the default K = 5, as many as the CU can hold once the columns are short enough (rankk_fit: 6 at <= 6144 rows,
  // 7 at <= 4096, 8 at <= 3072; DHQR_RANKK_MAX)
  const int Kmax = (K >= 5) ? std::max(K, std::min(c->rankk_max, DHQR_RK_KMAX)) : K;
  const int Kt = std::min(K, c->rankk_tall);  // ... while a column has 8192 < rows <= 16384 (k_rankk_tall; < 2: one per launch)
  const int Kx = Kt >= 2 ? std::min(Kt, c->rankk_xtall) : 1;  // ... 16384 < rows <= 32768 (k_rankk_xtall; < 2: one per launch)
  // a reflector slot: the column's rows, zero-padded so that k_rankk_tall (512 threads x 32 elements from a row > rows -
  // 16384) and k_rankk_xtall (512 x 64 from a row > rows - 32768) read reflectors without clamping or masking -- the
  // kernels never write beyond row `rows`
  const bool padded = (Kt >= 2 && rows > 1024 * 8) || c->rankk_pipe;
  const bool xpadded = Kx >= 2 && rows > 1024 * 16;
  const size_t vlen = (size_t)((rows + (xpadded ? 1024 * 16 : (padded ? 1024 * 8 : 0)) + 17) & ~(int64_t)15);
  CHECK(ensure(c, c->vbuf, 2 * (size_t)std::max(Kmax, 1) * vlen));
  if (padded || xpadded) HIPCHECK(hipMemsetAsync(c->vbuf.p, 0, 2 * (size_t)std::max(Kmax, 1) * vlen * sizeof(double), c->stream));
  double *vset[2] = {c->vbuf.p, c->vbuf.p + (size_t)std::max(Kmax, 1) * vlen};  // two sets of Kmax reflectors
  const bool vec = (ldp % 2 == 0) && (rows % 2 == 0) && aligned16(P);
  auto account = [&](int64_t jlo, int64_t ncol_upd) {
    if (!c->profiling) return;
    // algorithmic HBM bytes of the launch as implemented: every column it touches is read once and written once
    const double by = 16.0 * (double)(rows - jlo) * (double)ncol_upd;
    if (cat == CAT_RANK1) c->st.bytes_rank1 += by;
    else c->st.bytes_panel += by;
  };
  // Phase 1 -- columns taller than one workgroup's registers (> 8192 rows below the diagonal), or DHQR_RANKK=1: one
  // reflector per launch, v_j / v_j+1 ping-pong between slot 0 of the two sets.
  int64_t j = 0;
  int cur = 0;
  auto height = [&](int64_t jj) { return rows - (vec ? (jj & ~(int64_t)1) : jj); };
  auto tall = [&](int64_t jj) { return K < 2 || height(jj) > (Kx >= 2 ? 1024 * 32 : (Kt >= 2 ? 1024 * 16 : 1024 * 8)); };
  bool have_v = false;  // v_j built (in vset[cur][0])
  if (tall(0)) {
    CHECK(prof_begin(c, cat));
    hipLaunchKernelGGL((k_reflector<1024>), dim3(1), dim3(1024), 0, c->stream, P, rows, (int64_t)0, vset[0], alpha);
    CHECK(prof_end(c));
    have_v = true;
    for (; j + 1 < ncols && tall(j); ++j) {
      const int64_t nupd = ncols - (j + 1);
      CHECK(prof_begin(c, cat));
      if (vec) launch_rank1<2>(c, P, ldp, rows, j, nupd, vset[cur], vset[cur ^ 1], alpha);
      else launch_rank1<1>(c, P, ldp, rows, j, nupd, vset[cur], vset[cur ^ 1], alpha);
      CHECK(prof_end(c));
      account(j, nupd);
      cur ^= 1;
    }
  }
  // Phase 2 -- K reflectors per pass: each trailing column is loaded once, updated by v_jlo .. v_jlo+K-1 in the
  // reference's order and arithmetic, and stored once (16/K B of HBM traffic per element and reflector); the lead
  // workgroup of the pass builds the next K reflectors, so one launch per K columns.
  if (K >= 2 && (have_v ? j + 1 < ncols : ncols > 0)) {
    int64_t jlo;  // oldest reflector not yet applied to the trailing columns
    int kold;
    if (have_v) {  // continue from phase 1: v_j alone
      jlo = j;
      kold = 1;
    } else {       // first K reflectors from scratch (one workgroup)
      jlo = 0;
      kold = 0;
    }
    for (;;) {
      const int64_t c0 = jlo + kold;  // first column not yet final
      if (c0 >= ncols) break;
      // reflectors this pass builds (and the next one applies)
      // (more than K is STARTED only while the bulk clearly bounds the launch: a factorisation that begins with fewer than
      // ~4000 columns never goes above K -- 4096^2 31.4 -> 31.8 ms with 7 per pass.  Once a pass has built kold > K reflectors
      // the count does not come down again: the kernel that builds Kp holds at most Kp OLD reflectors on the CU, so the tail of
      // a larger factorisation -- the last 4096 columns of 8192^2 -- keeps 6-7 per pass; a step-down pass was not built.)
      int Kp;
      if (height(jlo) > 1024 * 16) Kp = Kx;
      else if (height(jlo) > 1024 * 8) Kp = Kt;
      else {
        const int fit = rankk_fit_rows(height(jlo), vec);
        Kp = (ncols - c0 > c->rankk_max_min_cols + K) ? std::max(K, std::min(Kmax, fit)) : K;
        Kp = std::max(Kp, std::min(kold, fit));
      }
      CHECK(prof_begin(c, cat));
      launch_rankk(c, vec, Kp, P, ldp, rows, ncols, c0, jlo, kold, vset[cur], vset[cur ^ 1], (int64_t)vlen, alpha);
      CHECK(prof_end(c));
      account(jlo, kold == 0 ? std::min<int64_t>(Kp, ncols) : ncols - c0);
      cur ^= 1;
      jlo = c0;
      kold = Kp;
    }
  }
  if (c->rankk_unfit) {
    const int k = c->rankk_unfit;
    c->rankk_unfit = 0;
    return set_err(DHQR_EINVAL, "internal: %d reflectors per pass asked of a kernel that cannot hold them", k);
  }
  LAUNCHCHECK();
  return DHQR_OK;
}

