// dhqr_mg.h -- single-process multi-GPU handle: one context, one communicator rank and one host thread per
// device; every call fans a job out to the rank threads, which run the SPMD drivers of dhqr_dist.h.
// This is what `qr!(A; ndev = 8)` of the Julia module and `python bench.py --gpus N` bind (included by
// dhqr_api.hip).  The multi-PROCESS form of the same drivers (one Julia worker / torchrun rank per GPU) is
// dhqr_comm_create_rank + dhqr_cs_*.
#pragma once
#include <condition_variable>
#include <functional>
#include <string>

struct MgRank {
  dhqr_ctx *c = nullptr;
  dhqr_comm *cm = nullptr;
  std::thread th;
  double *A = nullptr, *alpha = nullptr;  // local block-cyclic columns (m x ncl, lda), replicated alpha
  int64_t lda = 0, ncl = 0;
  size_t capA = 0;
  double *W = nullptr, *A0 = nullptr;     // residual scratch (allocated on first use)
  double *vec = nullptr;                  // solve: b / x (m), u (m + 128)
  double resid = 0.0;
};

struct dhqr_mg {
  int ndev = 0;
  std::vector<int> dev;
  std::vector<MgRank> rk;
  int transport = COMM_SELF;
  int64_t m = 0, n = 0;
  bool rowsplit = false;  // layout of the resident matrix: block-cyclic columns (default) or 128-row aligned row slabs
  // job dispatch
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  int64_t gen = 0;
  int pending = 0;
  bool quit = false;
  std::function<int32_t(int)> job;
  std::vector<int32_t> rc;
  std::vector<std::string> err;
};

static void mg_worker(dhqr_mg *g, int r) {
  (void)hipSetDevice(g->dev[r]);
  int64_t seen = 0;
  for (;;) {
    std::function<int32_t(int)> job;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_job.wait(lk, [&] { return g->quit || g->gen != seen; });
      if (g->quit) return;
      seen = g->gen;
      job = g->job;
    }
    g_err[0] = 0;
    int32_t rc = job(r);
    if (rc != DHQR_OK) comm_abort(g->rk[r].cm);  // peers stop waiting for this rank
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->rc[r] = rc;
      g->err[r] = g_err;
      if (--g->pending == 0) g->cv_done.notify_all();
    }
  }
}

// Run job(rank) on every rank thread; returns the first failure (its message becomes the caller's last error).
static int32_t mg_run(dhqr_mg *g, std::function<int32_t(int)> job) {
  // A failed job leaves the abort flag of the in-process transport set (peers stop waiting) and the ranks' collective
  // counters out of step.  Every rank thread is idle here: restart the mailbox from a clean state, otherwise one
  // failure poisons the handle for ever.
  for (auto &k : g->rk)
    for (dhqr_comm *cm : {k.cm, k.cm ? k.cm->lane : nullptr})
      if (cm && cm->world && cm->world->abort.load()) {
        for (auto &k2 : g->rk) {
          (void)hipSetDevice(k2.c->device);
          (void)hipDeviceSynchronize();
          for (dhqr_comm *c2 : {k2.cm, k2.cm ? k2.cm->lane : nullptr})
            if (c2 && c2->world == cm->world) c2->seq = 0;
        }
        LocalWorld *w = cm->world;
        for (int sidx = 0; sidx < DHQR_COMM_RING; ++sidx) {
          w->slot[sidx].seq.store(-1);
          w->slot[sidx].posted.store(0);
          w->slot[sidx].pulled.store(0);
          w->slot[sidx].need = 0;
        }
        w->bar_count.store(0);
        w->abort.store(0);
      }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->job = std::move(job);
    g->pending = g->ndev;
    for (int r = 0; r < g->ndev; ++r) g->rc[r] = DHQR_OK;
    g->gen++;
  }
  g->cv_job.notify_all();
  {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv_done.wait(lk, [&] { return g->pending == 0; });
  }
  // prefer a root-cause message over "a peer rank failed"
  int bad = -1;
  for (int r = 0; r < g->ndev; ++r)
    if (g->rc[r] != DHQR_OK && (bad < 0 || (g->err[bad].find("peer rank failed") != std::string::npos &&
                                            g->err[r].find("peer rank failed") == std::string::npos)))
      bad = r;
  if (bad < 0) return DHQR_OK;
  return set_err(g->rc[bad], "rank %d: %s", bad, g->err[bad].c_str());
}

static CsProblem mg_problem(dhqr_mg *g, int r) {
  CsProblem pr;
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
  pr.K = cs_nblocks(g->n);
  pr.ncl = k.ncl;
  return pr;
}

static int32_t mg_free_matrix(dhqr_mg *g) {
  return mg_run(g, [g](int r) -> int32_t {
    MgRank &k = g->rk[r];
    HIPCHECK(hipDeviceSynchronize());
    double **ps[] = {&k.A, &k.alpha, &k.W, &k.A0, &k.vec};
    for (double **p : ps)
      if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
      }
    k.capA = 0;
    return DHQR_OK;
  });
}
