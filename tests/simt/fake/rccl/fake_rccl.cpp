// tests/simt/fake/rccl/fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in librccl for the EMULATED library.
//
// The product reaches RCCL only through dlopen() (csrc/dhqr_comm.h: rccl_load).  On the CPU emulator "device" memory
// is host memory and every stream executes in host call order, so a collective can be a blocking rendezvous of the
// rank threads of one process.  This file implements the entry points the product resolves with the call semantics
// that matter for the host logic, and CHECKS what real RCCL silently assumes:
//   * communicators: ncclCommInitAll refuses duplicate devices (as the real one does); ncclCommInitRank blocks until
//     all `nranks` ranks have joined the same unique id; ncclCommDestroy is reference counted;
//   * every collective of one communicator is matched by sequence number and must agree on operation, count,
//     datatype and root across the ranks -- a mismatch is an error on every rank, not a hang;
//   * in-place ncclAllGather must be called with sendbuff == recvbuff + rank * count (the documented in-place rule);
//   * ncclSend / ncclRecv are rendezvous operations: outside a group a send blocks until the matching receive has
//     taken the data; inside ncclGroupStart / ncclGroupEnd all operations are deferred to the outermost GroupEnd,
//     sends are posted first, then receives complete, then the sends are waited for (so the scatter phase of the
//     scatter + all-gather broadcast cannot deadlock, and an ungrouped exchange of two ranks that both send first DOES);
//   * nothing waits for ever: every wait has a deadline (FAKE_RCCL_TIMEOUT_S, default 60 s); on expiry the call returns
//     ncclInternalError and names the operation, communicator, rank and sequence number on stderr -- e.g. two
//     communicators used in different orders by different ranks.
// Sums run in rank order (bitwise identical on every rank).  fake_rccl_stats() lets the tests assert which paths ran.
// Never shipped, never loadable by the product on a GPU box (DHQR_RCCL_LIB points at it only inside the tests).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <unistd.h>

extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef void *fakeStream_t;
}

namespace {
constexpr int MAXR = 64;
enum Op { OP_BCAST = 0, OP_ALLREDUCE = 1, OP_ALLGATHER = 2, OP_SEND = 3, OP_RECV = 4 };
const char *op_name(int op) {
  static const char *nm[] = {"ncclBroadcast", "ncclAllReduce", "ncclAllGather", "ncclSend", "ncclRecv"};
  return nm[op];
}
size_t dt_size(int dt) {
  switch (dt) {
    case ncclChar: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
double timeout_s() {
  const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 60.0;
}

struct Args {
  int op = -1, dtype = 0, root = 0;
  size_t count = 0;
  const void *send = nullptr;
  void *recv = nullptr;
  uint64_t seq = 0;
};
struct P2P {
  const void *ptr;
  size_t count;
  int dtype;
  std::shared_ptr<std::atomic<int>> done;
};
struct World {
  int id = 0, n = 0;
  std::mutex mu;
  std::condition_variable cv;
  Args args[MAXR];
  int arrived = 0;
  uint64_t gen = 0;
  bool broken = false;
  int joined = 0, destroyed = 0;
  std::vector<char> tmp[MAXR];
  std::deque<P2P> q[MAXR][MAXR];  // q[src][dst]
};
}  // namespace
struct ncclComm {
  World *w;
  int rank;
  uint64_t seq = 0;
};
namespace {
std::mutex g_mu;
std::map<std::string, World *> g_by_id;
std::atomic<int> g_world_ids{0};
std::atomic<int64_t> g_stat[16];  // 0 bcast 1 allreduce 2 allgather 3 send 4 recv 5 groups 6 comms 7 live comms 8 max live 9 timeouts 10 mismatches

struct Deferred {
  int op;
  ncclComm *cm;
  Args a;
  int peer;
};
thread_local int tl_depth = 0;
thread_local std::vector<Deferred> tl_ops;

using Clock = std::chrono::steady_clock;
Clock::time_point deadline() { return Clock::now() + std::chrono::milliseconds((int64_t)(timeout_s() * 1e3)); }

ncclResult_t fail_timeout(ncclComm *cm, const char *what, uint64_t seq, int arrived) {
  g_stat[9]++;
  fprintf(stderr, "fake rccl: rank %d timed out in %s #%llu of communicator %d (%d of %d ranks arrived) -- collectives "
                  "issued in different orders on different ranks?\n",
          cm->rank, what, (unsigned long long)seq, cm->w->id, arrived, cm->w->n);
  return ncclInternalError;
}

// barrier over the ranks of the world; false on timeout / broken world
bool barrier(ncclComm *cm, const char *what, uint64_t seq) {
  World *w = cm->w;
  std::unique_lock<std::mutex> lk(w->mu);
  if (w->broken) return false;
  const uint64_t my = w->gen;
  if (++w->arrived == w->n) {
    w->arrived = 0;
    w->gen++;
    w->cv.notify_all();
    return true;
  }
  if (!w->cv.wait_until(lk, deadline(), [&] { return w->gen != my || w->broken; }) || w->broken) {
    const int arrived = w->arrived;
    if (!w->broken) (void)fail_timeout(cm, what, seq, arrived);
    w->broken = true;
    w->cv.notify_all();
    return false;
  }
  return true;
}

ncclResult_t collective(ncclComm *cm, Args a) {
  World *w = cm->w;
  a.seq = cm->seq++;
  g_stat[a.op]++;
  if (w->n == 1) {  // single rank: a copy
    if (a.send != a.recv && a.count) memmove(a.recv, a.send, a.count * dt_size(a.dtype));
    return ncclSuccess;
  }
  w->args[cm->rank] = a;
  if (!barrier(cm, op_name(a.op), a.seq)) return ncclInternalError;
  // every rank validates the same table -> the same verdict everywhere
  bool ok = true;
  for (int r = 0; r < w->n; ++r) {
    const Args &b = w->args[r];
    if (b.op != a.op || b.count != a.count || b.dtype != a.dtype || b.seq != a.seq || (a.op == OP_BCAST && b.root != a.root)) ok = false;
  }
  const size_t bytes = a.count * dt_size(a.dtype);
  if (ok && a.op == OP_BCAST && (a.root < 0 || a.root >= w->n)) ok = false;
  if (ok && a.op == OP_ALLGATHER) {  // in-place rule
    const char *s = (const char *)a.send, *d = (const char *)a.recv;
    if (s >= d && s < d + bytes * w->n && s != d + bytes * cm->rank) {
      fprintf(stderr, "fake rccl: rank %d: in-place ncclAllGather needs sendbuff == recvbuff + rank * count\n", cm->rank);
      ok = false;  // only this rank sees it; the barrier below then reports the rest as broken
      std::lock_guard<std::mutex> lk(w->mu);
      w->broken = true;
      w->cv.notify_all();
    }
  }
  if (!ok) {
    g_stat[10]++;
    if (cm->rank == 0) {
      fprintf(stderr, "fake rccl: communicator %d: mismatched collective #%llu:", w->id, (unsigned long long)a.seq);
      for (int r = 0; r < w->n; ++r)
        fprintf(stderr, " [rank %d: %s count %zu dtype %d root %d seq %llu]", r, op_name(w->args[r].op), w->args[r].count,
                w->args[r].dtype, w->args[r].root, (unsigned long long)w->args[r].seq);
      fprintf(stderr, "\n");
    }
    (void)barrier(cm, "mismatch", a.seq);
    return ncclInvalidArgument;
  }
  // phase 1: read the peers' send buffers (nobody writes yet)
  std::vector<char> &tmp = w->tmp[cm->rank];
  if (a.op == OP_BCAST) {
    if (cm->rank != a.root && bytes) memcpy(a.recv, w->args[a.root].send, bytes);  // the root's buffer is not written by anyone
    else if (cm->rank == a.root && a.send != a.recv && bytes) memmove(a.recv, a.send, bytes);
  } else if (a.op == OP_ALLREDUCE) {
    tmp.resize(bytes);
    if (a.dtype == ncclFloat64) {
      double *t = (double *)tmp.data();
      for (size_t e = 0; e < a.count; ++e) {
        double s = 0.0;
        for (int r = 0; r < w->n; ++r) s += ((const double *)w->args[r].send)[e];
        t[e] = s;
      }
    } else {
      fprintf(stderr, "fake rccl: ncclAllReduce supports ncclFloat64 only\n");
      return ncclInvalidArgument;
    }
  } else {  // all-gather
    tmp.resize(bytes * w->n);
    for (int r = 0; r < w->n; ++r)
      if (bytes) memcpy(tmp.data() + bytes * r, w->args[r].send, bytes);
  }
  if (a.op != OP_BCAST) {
    if (!barrier(cm, op_name(a.op), a.seq)) return ncclInternalError;
    if (tmp.size()) memcpy(a.recv, tmp.data(), tmp.size());  // phase 2: write
  }
  if (!barrier(cm, op_name(a.op), a.seq)) return ncclInternalError;  // nobody leaves while a peer still reads its buffer
  return ncclSuccess;
}

ncclResult_t run_p2p(std::vector<Deferred> &ops) {
  // sends first (posted, not waited), then receives, then wait for the sends
  std::vector<std::pair<ncclComm *, std::shared_ptr<std::atomic<int>>>> pending;
  for (auto &d : ops)
    if (d.op == OP_SEND) {
      World *w = d.cm->w;
      auto done = std::make_shared<std::atomic<int>>(0);
      {
        std::lock_guard<std::mutex> lk(w->mu);
        w->q[d.cm->rank][d.peer].push_back(P2P{d.a.send, d.a.count, d.a.dtype, done});
      }
      w->cv.notify_all();
      pending.push_back({d.cm, done});
    }
  for (auto &d : ops)
    if (d.op == OP_RECV) {
      World *w = d.cm->w;
      std::unique_lock<std::mutex> lk(w->mu);
      auto &q = w->q[d.peer][d.cm->rank];
      if (!w->cv.wait_until(lk, deadline(), [&] { return !q.empty() || w->broken; }) || w->broken) {
        if (!w->broken) (void)fail_timeout(d.cm, "ncclRecv", d.cm->seq, 0);
        w->broken = true;
        w->cv.notify_all();
        return ncclInternalError;
      }
      P2P m = q.front();
      q.pop_front();
      lk.unlock();
      if (m.count != d.a.count || m.dtype != d.a.dtype) {
        g_stat[10]++;
        fprintf(stderr, "fake rccl: rank %d: ncclRecv of %zu elements from rank %d met a send of %zu\n", d.cm->rank, d.a.count, d.peer, m.count);
        m.done->store(1);
        w->cv.notify_all();
        return ncclInvalidArgument;
      }
      if (m.count) memcpy(d.a.recv, m.ptr, m.count * dt_size(m.dtype));
      m.done->store(1);
      w->cv.notify_all();
    }
  for (auto &p : pending) {
    World *w = p.first->w;
    std::unique_lock<std::mutex> lk(w->mu);
    if (!w->cv.wait_until(lk, deadline(), [&] { return p.second->load() != 0 || w->broken; }) || w->broken) {
      if (!w->broken) (void)fail_timeout(p.first, "ncclSend", p.first->seq, 0);
      w->broken = true;
      w->cv.notify_all();
      return ncclInternalError;
    }
  }
  return ncclSuccess;
}

ncclResult_t submit(Deferred d) {
  if (tl_depth > 0) {
    tl_ops.push_back(d);
    return ncclSuccess;
  }
  if (d.op == OP_SEND || d.op == OP_RECV) {
    std::vector<Deferred> one{d};
    return run_p2p(one);
  }
  return collective(d.cm, d.a);
}

ncclComm *new_comm(World *w, int rank) {
  ncclComm *c = new ncclComm{w, rank};
  g_stat[6]++;
  const int64_t live = ++g_stat[7];
  int64_t mx = g_stat[8].load();
  while (live > mx && !g_stat[8].compare_exchange_weak(mx, live)) {
  }
  return c;
}
}  // namespace

extern "C" {
const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInternalError: return "internal error (fake rccl: timeout / broken communicator)";
    case ncclInvalidArgument: return "invalid argument (fake rccl: mismatched collective)";
    case ncclInvalidUsage: return "invalid usage";
    default: return "error";
  }
}
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  static std::atomic<uint64_t> ctr{1};
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "fake-rccl-%d-%llu", (int)getpid(), (unsigned long long)ctr++);
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs) {
  if (!comms || n < 1 || n > MAXR) return ncclInvalidArgument;
  if (devs)
    for (int a = 0; a < n; ++a)
      for (int b = a + 1; b < n; ++b)
        if (devs[a] == devs[b]) return ncclInvalidUsage;  // "Duplicate GPU detected"
  World *w = new World();
  w->id = ++g_world_ids;
  w->n = n;
  w->joined = n;
  for (int r = 0; r < n; ++r) comms[r] = new_comm(w, r);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank) {
  if (!comm || n < 1 || n > MAXR || rank < 0 || rank >= n) return ncclInvalidArgument;
  const std::string key(id.internal, sizeof(id.internal));
  World *w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_by_id.find(key);
    if (it == g_by_id.end()) {
      w = new World();
      w->id = ++g_world_ids;
      w->n = n;
      g_by_id[key] = w;
    } else {
      w = it->second;
    }
  }
  if (w->n != n) return ncclInvalidArgument;
  *comm = new_comm(w, rank);
  std::unique_lock<std::mutex> lk(w->mu);
  if (++w->joined == n) {
    w->cv.notify_all();
    std::lock_guard<std::mutex> lk2(g_mu);
    g_by_id.erase(key);  // the id is used up
    return ncclSuccess;
  }
  if (!w->cv.wait_until(lk, deadline(), [&] { return w->joined >= n; })) {
    fprintf(stderr, "fake rccl: rank %d: ncclCommInitRank timed out (%d of %d joined)\n", rank, w->joined, n);
    return ncclInternalError;
  }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  World *w = c->w;
  bool last;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    last = ++w->destroyed == w->n;
  }
  delete c;
  g_stat[7]--;
  if (last) delete w;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) {
  if (!c || !n) return ncclInvalidArgument;
  *n = c->w->n;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) {
  if (!c || !r) return ncclInvalidArgument;
  *r = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void *s, void *d, size_t count, ncclDataType_t dt, int root, ncclComm_t c, fakeStream_t) {
  Args a;
  a.op = OP_BCAST; a.count = count; a.dtype = dt; a.root = root; a.send = s; a.recv = d;
  return submit(Deferred{OP_BCAST, c, a, -1});
}
ncclResult_t ncclAllReduce(const void *s, void *d, size_t count, ncclDataType_t dt, ncclRedOp_t, ncclComm_t c, fakeStream_t) {
  Args a;
  a.op = OP_ALLREDUCE; a.count = count; a.dtype = dt; a.send = s; a.recv = d;
  return submit(Deferred{OP_ALLREDUCE, c, a, -1});
}
ncclResult_t ncclAllGather(const void *s, void *d, size_t count, ncclDataType_t dt, ncclComm_t c, fakeStream_t) {
  Args a;
  a.op = OP_ALLGATHER; a.count = count; a.dtype = dt; a.send = s; a.recv = d;
  return submit(Deferred{OP_ALLGATHER, c, a, -1});
}
ncclResult_t ncclSend(const void *s, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, fakeStream_t) {
  if (peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
  g_stat[OP_SEND]++;
  Args a;
  a.op = OP_SEND; a.count = count; a.dtype = dt; a.send = s;
  return submit(Deferred{OP_SEND, c, a, peer});
}
ncclResult_t ncclRecv(void *d, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, fakeStream_t) {
  if (peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
  g_stat[OP_RECV]++;
  Args a;
  a.op = OP_RECV; a.count = count; a.dtype = dt; a.recv = d;
  return submit(Deferred{OP_RECV, c, a, peer});
}
ncclResult_t ncclGroupStart() {
  ++tl_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (tl_depth <= 0) return ncclInvalidUsage;
  if (--tl_depth > 0) return ncclSuccess;
  g_stat[5]++;
  std::vector<Deferred> ops;
  ops.swap(tl_ops);
  // point-to-point operations of the group complete together; collectives run in issue order
  std::vector<Deferred> p2p;
  for (auto &d : ops)
    if (d.op == OP_SEND || d.op == OP_RECV) p2p.push_back(d);
  ncclResult_t rc = p2p.empty() ? ncclSuccess : run_p2p(p2p);
  for (auto &d : ops)
    if (rc == ncclSuccess && d.op != OP_SEND && d.op != OP_RECV) rc = collective(d.cm, d.a);
  return rc;
}
// test hook: counters (0 bcast 1 allreduce 2 allgather 3 send 4 recv 5 groups 6 comms created 7 live comms
// 8 max live comms 9 timeouts 10 mismatches); reset != 0 clears them afterwards
void fake_rccl_stats(int64_t *out16, int reset) {
  for (int i = 0; i < 16; ++i) {
    out16[i] = g_stat[i].load();
    if (reset && i != 7) g_stat[i].store(i == 8 ? g_stat[7].load() : 0);
  }
}
}
