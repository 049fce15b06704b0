// dhqr_comm.h -- rank-to-rank transport of the multi-GPU drivers (dhqr_dist.h).
//
// The reference ships every reflector to every process with `@spawnat` + `@sync` (src:141-143) and sums
// partial dots with `sum(fetch.(futures))` (src:262-266).  Here the drivers are SPMD programs -- every rank
// runs the same sequence of collective calls -- over one of three transports behind the same two operations
// (broadcast of a device buffer, sum all-reduce of a small device buffer), all stream ordered:
//
//   RCCL      ncclBroadcast / ncclAllReduce over xGMI.  librccl.so is dlopen()ed when the first communicator
//             is created (single-GPU users never load it).  One communicator per rank: created by
//             ncclCommInitAll (all ranks in this process, one host thread each: dhqr_mg_*) or by
//             ncclCommInitRank from a 128-byte unique id the host layer ships between processes
//             (dhqr_comm_unique_id / dhqr_comm_create_rank: one Julia worker / one torchrun rank per GPU).
//   LOCAL     ranks are host threads of ONE process: the receiver pulls the root's buffer with a peer copy
//             (hipMemcpyAsync device-to-device; over xGMI when the devices differ) ordered by HIP events, the
//             host threads hand the events over through a small sequence-numbered mailbox.  Used when RCCL
//             cannot form the communicator (several ranks on one device, as in the 1-GPU tests) or on request
//             (DHQR_TRANSPORT=local).
//   CALLBACK  the host layer supplies the two operations (e.g. MPI.jl, Distributed.jl or torch.distributed):
//             "bring your own communicator".  Host synchronous.
//
// Send-buffer reuse: after bcast() returns, the ROOT may only overwrite the buffer on a stream that has
// called wait_consumed(ticket) (LOCAL: waits for the receivers' copy events; RCCL: stream order on the comm
// stream already guarantees it; CALLBACK: synchronous).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <thread>

#define DHQR_COMM_RING 32   // LOCAL transport: mailbox slots (collectives in flight between host threads)
#define DHQR_MAX_RANKS 64

// ---- RCCL entry points resolved at run time ------------------------------------------------------
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  // optional (scatter + all-gather broadcast); when one is missing the plain ncclBroadcast is the only algorithm
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional: the bench line reports the communicator's own size
  bool has_sag() const { return AllGather && Send && Recv && GroupStart && GroupEnd; }
};
static RcclApi g_rccl;
static std::atomic<int> g_rccl_state{0};  // 0 untried, 1 loaded, -1 unavailable

static int32_t rccl_load() {
  int st = g_rccl_state.load(std::memory_order_acquire);
  if (st == 1) return DHQR_OK;
  if (st == -1) return set_err(DHQR_ECOMM, "librccl.so is not loadable");
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (g_rccl_state.load() == 1) return DHQR_OK;
  // DHQR_RCCL_LIB: path of the RCCL build to use (a site's own librccl; the CPU tests point it at their stand-in)
  const char *names[] = {getenv("DHQR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *nm : names)
    if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    g_rccl_state.store(-1);
    return set_err(DHQR_ECOMM, "dlopen(librccl.so) failed: %s", dlerror());
  }
  bool ok = true;
  auto sym = [&](const char *nm) {
    void *p = dlsym(h, nm);
    if (!p) ok = false;
    return p;
  };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!ok) {
    g_rccl_state.store(-1);
    return set_err(DHQR_ECOMM, "librccl.so lacks a required symbol");
  }
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.Send = (decltype(g_rccl.Send))dlsym(h, "ncclSend");
  g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(h, "ncclRecv");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
  g_rccl.handle = h;
  g_rccl_state.store(1, std::memory_order_release);
  return DHQR_OK;
}
#define RCCLCHECK(expr)                                                                              \
  do {                                                                                               \
    ncclResult_t r_ = (expr);                                                                        \
    if (r_ != ncclSuccess)                                                                           \
      return set_err(DHQR_ECOMM, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// ---- LOCAL transport: mailbox shared by the host threads of one process ---------------------------
struct LocalSlot {
  std::atomic<int64_t> seq{-1};       // collective sequence number currently published in this slot
  const void *src[DHQR_MAX_RANKS];    // bcast: src[root]; all-reduce: every rank's buffer
  hipEvent_t ready[DHQR_MAX_RANKS];   // recorded by the publisher after the data is final on its stream
  std::atomic<int> posted{0};         // all-reduce: ranks that have published
  std::atomic<int> pulled{0};         // ranks that have enqueued their copies
  int need = 0;                       // value of `pulled` at which the slot may be recycled (P-1 bcast, P all-reduce)
};
struct LocalWorld {
  int nranks = 0;
  LocalSlot slot[DHQR_COMM_RING];
  hipEvent_t done[DHQR_MAX_RANKS][DHQR_COMM_RING];  // done[r][s]: rank r's copies out of slot s have finished
  std::atomic<int> refs{0};
  std::atomic<int> abort{0};          // a rank failed: everybody stops waiting
  std::atomic<int> bar_count{0};      // host barrier of the rank threads (sense reversing)
  std::atomic<int> bar_gen{0};
};

static inline void comm_pause(int &spins) {
  if (++spins < 2000) return;
  std::this_thread::yield();
}

enum CommKind { COMM_SELF = 0, COMM_RCCL = 1, COMM_LOCAL = 2, COMM_CALLBACK = 3 };

struct dhqr_comm {
  dhqr_ctx *ctx = nullptr;
  int kind = COMM_SELF, nranks = 1, rank = 0;
  ncclComm_t nccl = nullptr;
  LocalWorld *world = nullptr;
  int64_t seq = 0;                     // LOCAL: number of collectives this rank has issued
  double *scratch = nullptr;           // LOCAL all-reduce staging (nranks x cap doubles)
  size_t scratch_cap = 0;
  dhqr_bcast_fn cb_bcast = nullptr;
  dhqr_allreduce_fn cb_allreduce = nullptr;
  void *cb_user = nullptr;
  int64_t bytes_bcast = 0, n_bcast = 0;  // statistics
  int64_t bytes_allreduce = 0, n_allreduce = 0;  // all-reduces the drivers ISSUED (counted at one rank too, where they move nothing)
  // Second channel over the same ranks (own RCCL communicator / own mailbox): the row-split driver's look-ahead lane
  // issues its small latency-bound collectives here so they do not queue behind the wide stream's all-reduces (the
  // operations of ONE channel are ordered).  nullptr: CALLBACK transport (host synchronous anyway) or a single rank.
  dhqr_comm *lane = nullptr;
  // RCCL: algorithm of large broadcasts.  0 = ncclBroadcast (rings: the panel travels link by link), 1 = scatter +
  // all-gather (the root sends 1/P of the panel to every peer over its own xGMI link, then an all-gather among all ranks:
  // every link of the fully connected node carries 1/P of the bytes).  Chosen by comm_tune_bcast (a timed trial of both
  // on this node when the communicator is created) or DHQR_BCAST=ring|sag.
  int bcast_algo = 0;
  int64_t bcast_sag_min = (int64_t)1 << 17;  // doubles (1 MiB): below this a single ncclBroadcast
  double tune_ms[2] = {0.0, 0.0};            // what the trial measured (ring, scatter + all-gather), 16 MiB
  // Device time of this rank's collectives while the context profiles (dhqr_set_profiling): one hipEvent pair per
  // operation on the stream that carries it -- the wait for the peers included, which is what a multi-GPU run needs to
  // tell "the broadcast is slow" from "the panel chain is slow" (dhqr_comm_timing).  kind 0 broadcast, 1 all-reduce.
  struct TimedOp { hipEvent_t a, b; int kind; };
  std::vector<TimedOp> tev;
  size_t tev_used = 0;
  bool timing = false;
};

__global__ __launch_bounds__(256) void k_sum_ranks(const double *__restrict__ part, int nranks, int64_t stride,
                                                   int64_t count, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  double s = 0.0;
  for (int r = 0; r < nranks; ++r) s += part[(int64_t)r * stride + e];  // fixed rank order: identical on every rank
  out[e] = s;
}

// RCCL broadcast as scatter + all-gather (count % nranks == 0): chunk q of the buffer goes from the root to rank q
// (P - 1 concurrent point-to-point sends, one per xGMI link of the root), then an in-place all-gather.
static int32_t rccl_bcast_sag(dhqr_comm *cm, double *dbuf, int64_t count, int root, hipStream_t stream) {
  const int P = cm->nranks;
  const size_t chunk = (size_t)(count / P);
  RCCLCHECK(g_rccl.GroupStart());
  if (cm->rank == root) {
    for (int q = 0; q < P; ++q)
      if (q != root) RCCLCHECK(g_rccl.Send(dbuf + (size_t)q * chunk, chunk, ncclFloat64, q, cm->nccl, stream));
  } else {
    RCCLCHECK(g_rccl.Recv(dbuf + (size_t)cm->rank * chunk, chunk, ncclFloat64, root, cm->nccl, stream));
  }
  RCCLCHECK(g_rccl.GroupEnd());
  RCCLCHECK(g_rccl.AllGather(dbuf + (size_t)cm->rank * chunk, dbuf, chunk, ncclFloat64, cm->nccl, stream));
  return DHQR_OK;
}

// Timed trial of the two broadcast algorithms on THIS node (collective over cm; RCCL only): 16 MiB from root 0 and from
// the last rank, 3 repetitions after a warm-up that also sets up the point-to-point connections.  Every rank takes the
// same decision from the all-reduced (summed) times.  DHQR_BCAST=ring|sag skips the trial.
static int32_t comm_tune_bcast(dhqr_comm *cm, hipStream_t stream) {
  if (!cm || cm->kind != COMM_RCCL || cm->nranks < 2) return DHQR_OK;
  if (const char *e = getenv("DHQR_BCAST")) {
    if (!strcmp(e, "ring")) { cm->bcast_algo = 0; return DHQR_OK; }
    if (!strcmp(e, "sag")) { cm->bcast_algo = g_rccl.has_sag() ? 1 : 0; return DHQR_OK; }
  }
  if (!g_rccl.has_sag()) return DHQR_OK;
  const int P = cm->nranks;
  const int64_t count = ((int64_t)2 << 20) / P * P;  // ~16 MiB of doubles, divisible by P
  double *buf = nullptr, *dt = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc((void **)&buf, (size_t)count * 8) != hipSuccess) return set_err(DHQR_ENOMEM, "hipMalloc of the broadcast trial buffer failed");
  auto trial = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&dt, 2 * sizeof(double)));
    HIPCHECK(hipMemsetAsync(buf, 0, (size_t)count * 8, stream));
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    double ms[2] = {0.0, 0.0};
    for (int algo = 0; algo < 2; ++algo) {
      for (int rep = -1; rep < 3; ++rep) {  // rep -1: warm-up
        if (rep == 0) HIPCHECK(hipEventRecord(e0, stream));
        for (int root : {0, P - 1}) {
          if (algo == 0) RCCLCHECK(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat64, root, cm->nccl, stream));
          else CHECK(rccl_bcast_sag(cm, buf, count, root, stream));
        }
      }
      HIPCHECK(hipEventRecord(e1, stream));
      HIPCHECK(hipStreamSynchronize(stream));
      float t = 0.f;
      HIPCHECK(hipEventElapsedTime(&t, e0, e1));
      ms[algo] = (double)t / 6.0;
    }
    HIPCHECK(hipMemcpyAsync(dt, ms, 2 * sizeof(double), hipMemcpyHostToDevice, stream));
    RCCLCHECK(g_rccl.AllReduce(dt, dt, 2, ncclFloat64, ncclSum, cm->nccl, stream));
    HIPCHECK(hipMemcpyAsync(ms, dt, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    cm->tune_ms[0] = ms[0] / P;
    cm->tune_ms[1] = ms[1] / P;
    cm->bcast_algo = (ms[1] < 0.9 * ms[0]) ? 1 : 0;  // the same sums on every rank: the same decision
    return DHQR_OK;
  };
  const int32_t rc = trial();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (dt) (void)hipFree(dt);
  (void)hipFree(buf);
  return rc;
}

// Broadcast `count` doubles at dbuf from rank `root`, ordered on `stream`.  *ticket (optional) identifies the
// operation for comm_wait_consumed.
static int32_t comm_time_begin(dhqr_comm *cm, int kind, hipStream_t stream) {
  if (cm->tev_used == cm->tev.size()) {
    dhqr_comm::TimedOp e;
    HIPCHECK(hipEventCreate(&e.a));
    HIPCHECK(hipEventCreate(&e.b));
    e.kind = kind;
    cm->tev.push_back(e);
  }
  cm->tev[cm->tev_used].kind = kind;
  HIPCHECK(hipEventRecord(cm->tev[cm->tev_used].a, stream));
  return DHQR_OK;
}
static int32_t comm_time_end(dhqr_comm *cm, hipStream_t stream) {
  HIPCHECK(hipEventRecord(cm->tev[cm->tev_used].b, stream));
  cm->tev_used++;
  return DHQR_OK;
}
static int32_t comm_bcast_impl(dhqr_comm *cm, double *dbuf, int64_t count, int root, hipStream_t stream, int64_t *ticket);
static int32_t comm_bcast(dhqr_comm *cm, double *dbuf, int64_t count, int root, hipStream_t stream, int64_t *ticket) {
  const bool timed = cm->timing && cm->nranks > 1 && count > 0;
  if (timed) CHECK(comm_time_begin(cm, 0, stream));
  const int32_t rc = comm_bcast_impl(cm, dbuf, count, root, stream, ticket);
  if (timed) CHECK(comm_time_end(cm, stream));
  return rc;
}
static int32_t comm_bcast_impl(dhqr_comm *cm, double *dbuf, int64_t count, int root, hipStream_t stream, int64_t *ticket) {
  if (ticket) *ticket = -1;
  if (cm->nranks == 1 || count <= 0) return DHQR_OK;
  cm->bytes_bcast += count * 8;
  cm->n_bcast++;
  if (cm->kind == COMM_RCCL) {
    if (cm->bcast_algo == 1 && count >= cm->bcast_sag_min && count % cm->nranks == 0) return rccl_bcast_sag(cm, dbuf, count, root, stream);
    RCCLCHECK(g_rccl.Broadcast(dbuf, dbuf, (size_t)count, ncclFloat64, root, cm->nccl, stream));
    return DHQR_OK;
  }
  if (cm->kind == COMM_CALLBACK) {
    HIPCHECK(hipStreamSynchronize(stream));
    const int32_t rc = cm->cb_bcast(cm->cb_user, dbuf, count * 8, root, (void *)stream);
    if (rc != 0) return set_err(DHQR_ECOMM, "broadcast callback failed (%d)", rc);
    return DHQR_OK;
  }
  // LOCAL
  LocalWorld *w = cm->world;
  const int64_t s = cm->seq++;
  LocalSlot &sl = w->slot[s % DHQR_COMM_RING];
  int spins = 0;
  if (cm->rank == root) {
    // the slot's previous collective (s - RING) must have been consumed by everybody
    while (sl.pulled.load(std::memory_order_acquire) < sl.need) {
      if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
      comm_pause(spins);
    }
    HIPCHECK(hipEventRecord(sl.ready[root], stream));
    sl.src[root] = dbuf;
    sl.need = w->nranks - 1;
    sl.pulled.store(0, std::memory_order_relaxed);
    sl.posted.store(0, std::memory_order_relaxed);
    sl.seq.store(s, std::memory_order_release);
    if (ticket) *ticket = s;
    return DHQR_OK;
  }
  while (sl.seq.load(std::memory_order_acquire) != s) {
    if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
    comm_pause(spins);
  }
  HIPCHECK(hipStreamWaitEvent(stream, sl.ready[root], 0));
  HIPCHECK(hipMemcpyAsync(dbuf, sl.src[root], (size_t)count * 8, hipMemcpyDeviceToDevice, stream));
  HIPCHECK(hipEventRecord(w->done[cm->rank][s % DHQR_COMM_RING], stream));
  sl.pulled.fetch_add(1, std::memory_order_acq_rel);
  return DHQR_OK;
}

// Make `stream` wait until every receiver of broadcast `ticket` (rooted here) has copied the data out.
static int32_t comm_wait_consumed(dhqr_comm *cm, int64_t ticket, hipStream_t stream) {
  if (cm->kind != COMM_LOCAL || ticket < 0) return DHQR_OK;
  LocalWorld *w = cm->world;
  LocalSlot &sl = w->slot[ticket % DHQR_COMM_RING];
  int spins = 0;
  while (sl.seq.load(std::memory_order_acquire) == ticket && sl.pulled.load(std::memory_order_acquire) < w->nranks - 1) {
    if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
    comm_pause(spins);
  }
  // Slot already recycled (seq != ticket): recycling only needs every receiver to have ENQUEUED its copy, so the copies
  // may still be in flight -- wait on the receivers' events all the same (a re-recorded event completes after the
  // earlier work of the stream it was recorded on, so waiting on it is at least as strong).
  for (int r = 0; r < w->nranks; ++r)
    if (r != cm->rank) HIPCHECK(hipStreamWaitEvent(stream, w->done[r][ticket % DHQR_COMM_RING], 0));
  return DHQR_OK;
}

// In-place sum over the ranks of `count` doubles at dbuf (small buffers: partial dots, norms).  The result is
// bitwise identical on every rank (fixed summation order).
static int32_t comm_allreduce_sum_impl(dhqr_comm *cm, double *dbuf, int64_t count, hipStream_t stream);
static int32_t comm_allreduce_sum(dhqr_comm *cm, double *dbuf, int64_t count, hipStream_t stream) {
  const bool timed = cm->timing && cm->nranks > 1 && count > 0;
  if (timed) CHECK(comm_time_begin(cm, 1, stream));
  const int32_t rc = comm_allreduce_sum_impl(cm, dbuf, count, stream);
  if (timed) CHECK(comm_time_end(cm, stream));
  return rc;
}
static int32_t comm_allreduce_sum_impl(dhqr_comm *cm, double *dbuf, int64_t count, hipStream_t stream) {
  if (count > 0) {
    cm->bytes_allreduce += count * 8;
    cm->n_allreduce++;
  }
  if (cm->nranks == 1 || count <= 0) return DHQR_OK;
  if (cm->kind == COMM_RCCL) {
    RCCLCHECK(g_rccl.AllReduce(dbuf, dbuf, (size_t)count, ncclFloat64, ncclSum, cm->nccl, stream));
    return DHQR_OK;
  }
  if (cm->kind == COMM_CALLBACK) {
    HIPCHECK(hipStreamSynchronize(stream));
    const int32_t rc = cm->cb_allreduce(cm->cb_user, dbuf, count, (void *)stream);
    if (rc != 0) return set_err(DHQR_ECOMM, "all-reduce callback failed (%d)", rc);
    return DHQR_OK;
  }
  LocalWorld *w = cm->world;
  const int P = w->nranks;
  if ((size_t)count * P > cm->scratch_cap) {
    HIPCHECK(hipStreamSynchronize(stream));
    if (cm->scratch) HIPCHECK(hipFree(cm->scratch));
    cm->scratch = nullptr;
    cm->scratch_cap = 0;
    const size_t cap = ((size_t)count * P + 1023) & ~(size_t)1023;
    if (hipMalloc((void **)&cm->scratch, cap * sizeof(double)) != hipSuccess)
      return set_err(DHQR_ENOMEM, "hipMalloc of the all-reduce staging buffer failed");
    cm->scratch_cap = cap;
  }
  const int64_t s = cm->seq++;
  LocalSlot &sl = w->slot[s % DHQR_COMM_RING];
  int spins = 0;
  // rank 0 opens the slot (after its previous use was consumed), the others wait for it
  if (cm->rank == 0) {
    while (sl.pulled.load(std::memory_order_acquire) < sl.need) {
      if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
      comm_pause(spins);
    }
    sl.need = P;
    sl.pulled.store(0, std::memory_order_relaxed);
    sl.posted.store(0, std::memory_order_relaxed);
    sl.seq.store(s, std::memory_order_release);
  } else {
    while (sl.seq.load(std::memory_order_acquire) != s) {
      if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
      comm_pause(spins);
    }
  }
  HIPCHECK(hipEventRecord(sl.ready[cm->rank], stream));
  sl.src[cm->rank] = dbuf;
  sl.posted.fetch_add(1, std::memory_order_acq_rel);
  while (sl.posted.load(std::memory_order_acquire) < P) {
    if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
    comm_pause(spins);
  }
  for (int r = 0; r < P; ++r) {
    if (r != cm->rank) HIPCHECK(hipStreamWaitEvent(stream, sl.ready[r], 0));
    HIPCHECK(hipMemcpyAsync(cm->scratch + (size_t)r * count, sl.src[r], (size_t)count * 8, hipMemcpyDeviceToDevice, stream));
  }
  HIPCHECK(hipEventRecord(w->done[cm->rank][s % DHQR_COMM_RING], stream));
  sl.pulled.fetch_add(1, std::memory_order_acq_rel);
  // nobody may overwrite its dbuf (below) before every rank has copied it: wait for all copies
  while (sl.pulled.load(std::memory_order_acquire) < P) {
    if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
    comm_pause(spins);
  }
  for (int r = 0; r < P; ++r)
    if (r != cm->rank) HIPCHECK(hipStreamWaitEvent(stream, w->done[r][s % DHQR_COMM_RING], 0));
  hipLaunchKernelGGL(k_sum_ranks, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, (const double *)cm->scratch, P,
                     count, count, dbuf);
  LAUNCHCHECK();
  return DHQR_OK;
}

// Host barrier of the in-process ranks (LOCAL transport only; a no-op elsewhere): used where a broadcast SOURCE
// buffer is recycled immediately (residual / solve / layout conversion), after each rank synchronised its stream.
static int32_t comm_host_barrier(dhqr_comm *cm) {
  if (!cm || cm->kind != COMM_LOCAL) return DHQR_OK;
  LocalWorld *w = cm->world;
  const int gen = w->bar_gen.load(std::memory_order_acquire);
  if (w->bar_count.fetch_add(1, std::memory_order_acq_rel) == w->nranks - 1) {
    w->bar_count.store(0, std::memory_order_relaxed);
    w->bar_gen.store(gen + 1, std::memory_order_release);
    return DHQR_OK;
  }
  int spins = 0;
  while (w->bar_gen.load(std::memory_order_acquire) == gen) {
    if (w->abort.load()) return set_err(DHQR_ECOMM, "a peer rank failed");
    comm_pause(spins);
  }
  return DHQR_OK;
}

static void comm_abort(dhqr_comm *cm) {
  if (cm && cm->world) cm->world->abort.store(1);
  if (cm && cm->lane && cm->lane->world) cm->lane->world->abort.store(1);
}

static int32_t comm_free(dhqr_comm *cm) {
  if (!cm) return DHQR_OK;
  if (cm->lane) (void)comm_free(cm->lane);
  cm->lane = nullptr;
  if (cm->ctx) (void)hipSetDevice(cm->ctx->device);
  if (cm->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(cm->nccl);
  if (cm->scratch) (void)hipFree(cm->scratch);
  for (auto &e : cm->tev) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  if (cm->world && cm->world->refs.fetch_sub(1) == 1) {
    for (int r = 0; r < cm->world->nranks; ++r)
      for (int s = 0; s < DHQR_COMM_RING; ++s) {
        if (cm->world->done[r][s]) (void)hipEventDestroy(cm->world->done[r][s]);
        if (cm->world->slot[s].ready[r]) (void)hipEventDestroy(cm->world->slot[s].ready[r]);
      }
    delete cm->world;
  }
  delete cm;
  return DHQR_OK;
}

// LOCAL world for `n` in-process ranks on devices dev[0..n): events are created on the owning device.
static int32_t local_world_create(LocalWorld **out, const int *dev, int n) {
  LocalWorld *w = new LocalWorld();
  w->nranks = n;
  for (int s = 0; s < DHQR_COMM_RING; ++s)
    for (int r = 0; r < DHQR_MAX_RANKS; ++r) {
      w->slot[s].ready[r] = nullptr;
      w->slot[s].src[r] = nullptr;
      w->done[r][s] = nullptr;
    }
  *out = w;
  for (int r = 0; r < n; ++r) {
    HIPCHECK(hipSetDevice(dev[r]));
    for (int s = 0; s < DHQR_COMM_RING; ++s) {
      HIPCHECK(hipEventCreateWithFlags(&w->slot[s].ready[r], hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&w->done[r][s], hipEventDisableTiming));
    }
  }
  return DHQR_OK;
}
