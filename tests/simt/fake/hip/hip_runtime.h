// tests/simt/fake/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a SIMT emulator + a host runtime subset.
//
// Put on the include path IN FRONT of ROCm's headers, it lets the UNMODIFIED sources of csrc/ compile
// with the host clang++ and run on the CPU.  Two execution modes for the threads of a workgroup:
//   default        every HIP thread is an OS thread; __syncthreads() is a (futex) barrier over the live
//                  threads of the block.  Built with -fsanitize=thread a missing barrier / LDS double-
//                  buffering mistake becomes a reported data race (tests/test_simt_emulation.py).
//   -DSIMT_FIBERS  every HIP thread is a cooperative fiber (ucontext) of ONE OS thread; barriers switch
//                  fibers round robin.  ~100x faster, deterministic, detects divergent barriers as a
//                  deadlock; used to run the whole library end to end (tests/test_emulated_library.py).
// Common semantics: __shfl/__shfl_xor and v_mfma_f64_16x16x4_f64 are wave-synchronous exchanges through
// per-wavefront slots (all 64 lanes must call, as on the hardware with a full EXEC mask); `__shared__`
// becomes `static` (one workgroup runs at a time); workgroups of a grid run one after the other.
// It never ships and is never a fallback: the product is libdhqr.so, built by hipcc for gfx950 only.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#ifdef SIMT_FIBERS
#include <ucontext.h>
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct simt_uint3 { unsigned x, y, z; };
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

namespace simt {
#ifndef SIMT_FIBERS
// ------------------------------------------------------------------------------------------------
// OS-thread mode.  Barrier whose participant count shrinks when a thread's kernel function returns.
// Lock free: the arrival state (generation | waiting | live) is one 64-bit word updated by compare-and-
// swap; sleeping / waking is a futex wait on a separate 32-bit generation word (std::atomic::wait, C++20),
// so a release wakes the 1024 threads of a workgroup in parallel (a mutex + condition-variable barrier
// hands the lock from thread to thread: 30 ms per barrier with 1024 threads on 8 cores).
struct Barrier {
  std::atomic<uint64_t> state{0};  // gen << 32 | waiting << 16 | live
  std::atomic<uint32_t> gen_word{0};
  static uint64_t pack(uint32_t gen, uint32_t waiting, uint32_t live) {
    return ((uint64_t)gen << 32) | ((uint64_t)waiting << 16) | (uint64_t)live;
  }
  void reset(int n) {  // only called while no thread is inside wait()
    const uint32_t g = gen_word.load(std::memory_order_relaxed);
    state.store(pack(g, 0, (uint32_t)n), std::memory_order_release);
  }
  void release(uint32_t newgen) {
    gen_word.store(newgen, std::memory_order_release);
    gen_word.notify_all();
  }
  void wait() {
    uint64_t s = state.load(std::memory_order_acquire);
    for (;;) {
      const uint32_t g = (uint32_t)(s >> 32), w = (uint32_t)((s >> 16) & 0xffff) + 1, live = (uint32_t)(s & 0xffff);
      if (w == live) {  // last arrival: open the next generation
        if (state.compare_exchange_weak(s, pack(g + 1, 0, live), std::memory_order_acq_rel)) {
          release(g + 1);
          return;
        }
      } else if (state.compare_exchange_weak(s, pack(g, w, live), std::memory_order_acq_rel)) {
        while (gen_word.load(std::memory_order_acquire) == g) gen_word.wait(g, std::memory_order_acquire);
        return;
      }
    }
  }
  void leave() {
    uint64_t s = state.load(std::memory_order_acquire);
    for (;;) {
      const uint32_t g = (uint32_t)(s >> 32), w = (uint32_t)((s >> 16) & 0xffff), live = (uint32_t)(s & 0xffff) - 1;
      if (live > 0 && w == live) {  // everybody else is already waiting
        if (state.compare_exchange_weak(s, pack(g + 1, 0, live), std::memory_order_acq_rel)) {
          release(g + 1);
          return;
        }
      } else if (state.compare_exchange_weak(s, pack(g, w, live), std::memory_order_acq_rel)) {
        return;
      }
    }
  }
};
inline thread_local int tl_tid = 0;
inline thread_local unsigned tl_xchg = 0;  // exchanges (shuffles / MFMAs) executed by this lane in this workgroup
inline int cur_tid() { return tl_tid; }
inline unsigned &cur_xchg() { return tl_xchg; }
#else
// ------------------------------------------------------------------------------------------------
// Fiber mode: one OS thread, cooperative round-robin scheduling.  A barrier is a counter; a waiting fiber
// yields until the generation changes.  Every switch lets the target fiber advance by one barrier, so the
// cost is (number of fibers) switches per barrier and nothing is wasted on polling.
struct Fiber {
  ucontext_t ctx;
  std::unique_ptr<char[]> stack;
  bool done = true;
  simt_uint3 tidx{0, 0, 0};
  unsigned xchg = 0;
};
struct Sched {
  std::vector<Fiber> fb;
  ucontext_t main_ctx;
  int cur = 0, n = 0;
  const std::function<void()> *body = nullptr;
  struct Block *blk = nullptr;
};
inline Sched &sched() { static Sched s; return s; }
inline uint64_t &progress() { static uint64_t p = 0; return p; }  // bumped whenever any fiber moves forward
inline int cur_tid() { return sched().cur; }
inline unsigned &cur_xchg() { return sched().fb[sched().cur].xchg; }
inline bool yield_to_next() {  // false when no other unfinished fiber exists
  Sched &s = sched();
  for (int i = 1; i < s.n; ++i) {
    const int c = (s.cur + i) % s.n;
    if (!s.fb[c].done) {
      const int prev = s.cur;
      s.cur = c;
      swapcontext(&s.fb[prev].ctx, &s.fb[c].ctx);
      return true;
    }
  }
  return false;
}
struct Barrier {
  int live = 0, waiting = 0;
  uint32_t gen = 0;
  void reset(int n) { live = n; waiting = 0; }
  void wait() {
    const uint32_t g = gen;
    ++progress();
    if (++waiting == live) { waiting = 0; ++gen; return; }
    while (gen == g) {
      const uint64_t p0 = progress();
      // a full round over the other fibers without anybody arriving anywhere or finishing: nobody can move
      if (!yield_to_next() || (progress() == p0 && gen == g)) {
        std::fprintf(stderr, "SIMT emulator: deadlock -- a barrier / shuffle is not reached by every live thread "
                             "of its scope (thread %d)\n", sched().cur);
        std::abort();
      }
    }
  }
  void leave() {
    ++progress();
    --live;
    if (live > 0 && waiting == live) { waiting = 0; ++gen; }
  }
};
#endif

struct Wave {
  Barrier bar;
  uint64_t buf[2][64];            // shuffle payloads, double buffered by exchange parity
  double mfa[2][64], mfb[2][64];  // operands of an emulated MFMA
};
struct Block {
  Barrier bar;
  std::vector<Wave> waves;
};
#ifndef SIMT_FIBERS
inline Block *&cur_block() { static Block *b = nullptr; return b; }
#else
inline Block *&cur_block() { return sched().blk; }
#endif
}  // namespace simt

#ifndef SIMT_FIBERS
inline thread_local simt_uint3 threadIdx{0, 0, 0};
#else
#define threadIdx (simt::sched().fb[simt::sched().cur].tidx)
#endif
inline simt_uint3 blockIdx{0, 0, 0};  // one workgroup at a time: shared by its threads
inline dim3 blockDim, gridDim;

inline void __syncthreads() { simt::cur_block()->bar.wait(); }

// Wave-synchronous exchange: every lane deposits its payload in the slot of the current parity, ONE wave
// barrier, then reads its source lane.  The next exchange uses the other parity, and a lane can only reach
// the exchange after that once every lane has passed the barrier in between, i.e. has finished reading.
template <typename T>
inline T simt_shfl_bits(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  simt::Wave &w = simt::cur_block()->waves[simt::cur_tid() >> 6];
  const int lane = simt::cur_tid() & 63;
  const unsigned par = simt::cur_xchg()++ & 1u;
  uint64_t bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  w.buf[par][lane] = bits;
  w.bar.wait();
  const uint64_t got = w.buf[par][src_lane & 63];
  T r;
  std::memcpy(&r, &got, sizeof(T));
  return r;
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  const int lane = simt::cur_tid() & 63;
  return simt_shfl_bits(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  const int lane = simt::cur_tid() & 63;
  (void)width;
  return simt_shfl_bits(v, lane ^ mask);
}

// ---- DPP (v_mov_b32_dpp through __builtin_amdgcn_update_dpp) and v_readlane: wave-synchronous exchanges too.  The
// controls the kernels use: quad_perm (0x00-0xFF), row_shr:n (0x111-0x11F), row_ror:n (0x121-0x12F), row_mirror (0x140),
// row_half_mirror (0x141), row_bcast:15 (0x142), row_bcast:31 (0x143); bank_mask must be 0xf; disabled rows and lanes
// without a source keep `old` (bound_ctrl = false).
inline int simt_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = simt::cur_tid() & 63, row = lane >> 4, r = lane & 15;
  int s = -1;  // source lane, -1: none
  if (ctrl >= 0x00 && ctrl <= 0xFF) s = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x111 && ctrl <= 0x11F) s = (r - (ctrl & 15) >= 0) ? lane - (ctrl & 15) : -1;
  else if (ctrl >= 0x121 && ctrl <= 0x12F) s = (lane & ~15) | ((r - (ctrl & 15)) & 15);
  else if (ctrl == 0x140) s = (lane & ~15) | (15 - r);
  else if (ctrl == 0x141) s = (lane & ~7) | (7 - (lane & 7));
  else if (ctrl == 0x142) s = (row >= 1) ? 16 * (row - 1) + 15 : -1;
  else if (ctrl == 0x143) s = (row >= 2) ? 31 : -1;
  const int got = simt_shfl_bits(src, s < 0 ? lane : s);  // every lane takes part in the exchange
  if (bank_mask != 0xf) std::abort();
  if (!((row_mask >> row) & 1)) return old;
  if (s < 0) return bound_ctrl ? 0 : old;
  return got;
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) simt_update_dpp(old, src, ctrl, rm, bm, bc)
inline int simt_readlane(int v, int l) { return simt_shfl_bits(v, l); }
#define __builtin_amdgcn_readlane(v, l) simt_readlane(v, l)
#define __builtin_amdgcn_readfirstlane(v) simt_readlane(v, 0)
// s_nop-sized on the hardware (a wave executes in lockstep; the builtin only pins the compiler's schedule).  Here the lanes
// of a wave are independent fibers / threads: lanes that hand data to each other through LDS meet at an exchange.
#define __builtin_amdgcn_wave_barrier() ((void)simt_readlane(0, 0))
inline int __double2loint(double d) { uint64_t b; std::memcpy(&b, &d, 8); return (int)(uint32_t)b; }
inline int __double2hiint(double d) { uint64_t b; std::memcpy(&b, &d, 8); return (int)(uint32_t)(b >> 32); }
inline double __hiloint2double(int hi, int lo) {
  const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double d;
  std::memcpy(&d, &b, 8);
  return d;
}

// ---- v_mfma_f64_16x16x4_f64, wave-synchronous like the instruction: every lane contributes one A and one
// B element, D[i][j] += sum_k A[i][k] B[k][j] with the gfx950 operand maps documented in csrc/dhqr_gemm.h
//   A: lane l holds A[i = l&15][k = l>>4]   B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, register g holds D[i = (l>>4) + 4g][j = l&15]
// (the map itself is pinned on the device by tests/test_gpu_kernels.py::test_mfma_layout_probe)
typedef double simt_d4 __attribute__((ext_vector_type(4)));
inline simt_d4 simt_mfma_f64_16x16x4(double a, double b, simt_d4 c) {
  simt::Wave &w = simt::cur_block()->waves[simt::cur_tid() >> 6];
  const int lane = simt::cur_tid() & 63;
  const unsigned par = simt::cur_xchg()++ & 1u;
  w.mfa[par][lane] = a;
  w.mfb[par][lane] = b;
  w.bar.wait();
  const int j = lane & 15;
  for (int g = 0; g < 4; ++g) {
    const int i = (lane >> 4) + 4 * g;
    double s = c[g];
    for (int k = 0; k < 4; ++k) s = std::fma(w.mfa[par][i + 16 * k], w.mfb[par][j + 16 * k], s);
    c[g] = s;
  }
  return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) simt_mfma_f64_16x16x4(a, b, c)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt((double)(x)))  // v_rsq_f64 / v_rcp_f64: the refinement steps that follow
#define __builtin_amdgcn_rcp(x) (1.0 / (double)(x))             // them in csrc/ are run as written
inline long long clock64() { return 0; }
inline long long wall_clock64() {
  static std::atomic<long long> t{0};
  return t += 1000;  // every poll advances the fake constant-rate counter: bounded spins terminate
}
// s_sleep sits in polling loops (inter-workgroup flags: never taken here, workgroups run one after the other; LDS flags
// between the waves of one workgroup: dhqr_small.h): give the other fibers / threads of the workgroup the processor
#ifdef SIMT_FIBERS
#define __builtin_amdgcn_s_sleep(x) ((void)simt::yield_to_next())
#else
#define __builtin_amdgcn_s_sleep(x) (std::this_thread::yield())
#endif
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
// inter-workgroup flags (k_zpanel_pipe): workgroups run one after the other here, the atomics are plain host atomics
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_max(p, v, order, scope) __atomic_fetch_max((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

namespace simt {
#ifndef SIMT_FIBERS
// Run a gx x gy grid of INDEPENDENT workgroups of `nthreads` threads, one workgroup after the other.
// The OS threads are created once and walk the grid in lockstep: [fixed barrier] thread 0 re-arms the
// dynamic barriers [fixed barrier] kernel body, leave() -> next workgroup.  The fixed barriers also keep a
// fast thread from touching the (static) LDS arrays of workgroup n+1 while a slow one still reads n.
inline void launch_grid(int gx, int gy, int nthreads, const std::function<void()> &body, int gz = 1) {
  Block blk;
  blk.waves = std::vector<Wave>((nthreads + 63) / 64);
  Barrier fence;  // fixed participant count, nobody leaves
  fence.reset(nthreads);
  cur_block() = &blk;
  blockDim = dim3(nthreads);
  gridDim = dim3(gx, gy, gz);
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      tl_tid = t;
      threadIdx = simt_uint3{(unsigned)t, 0, 0};
      for (int yz = 0; yz < gy * gz; ++yz)
        for (int x = 0; x < gx; ++x) {
          const int y = yz % gy, z = yz / gy;
          fence.wait();
          if (t == 0) {
            blk.bar.reset(nthreads);
            for (int w = 0; w < (int)blk.waves.size(); ++w) blk.waves[w].bar.reset(std::min(64, nthreads - 64 * w));
            blockIdx = simt_uint3{(unsigned)x, (unsigned)y, (unsigned)z};
          }
          tl_xchg = 0;
          fence.wait();
          body();
          blk.waves[t >> 6].bar.leave();
          blk.bar.leave();
        }
    });
  for (auto &x : th) x.join();
  cur_block() = nullptr;
}
#else
inline void fiber_entry() {
  Sched &s = sched();
  (*s.body)();
  Fiber &f = s.fb[s.cur];
  f.done = true;
  s.blk->waves[s.cur >> 6].bar.leave();
  s.blk->bar.leave();
  if (!yield_to_next()) swapcontext(&f.ctx, &s.main_ctx);  // last one out returns to the launcher
  std::abort();                                             // a finished fiber is never resumed
}
inline void launch_grid(int gx, int gy, int nthreads, const std::function<void()> &body, int gz = 1) {
  constexpr size_t STACK = 256 * 1024;
  Sched &s = sched();
  Block blk;
  blk.waves = std::vector<Wave>((nthreads + 63) / 64);
  if ((int)s.fb.size() < nthreads) s.fb.resize(nthreads);
  s.n = nthreads;
  s.body = &body;
  s.blk = &blk;
  blockDim = dim3(nthreads);
  gridDim = dim3(gx, gy, gz);
  for (int yz = 0; yz < gy * gz; ++yz)
    for (int x = 0; x < gx; ++x) {
      const int y = yz % gy, z = yz / gy;
      blk.bar.reset(nthreads);
      for (int w = 0; w < (int)blk.waves.size(); ++w) blk.waves[w].bar.reset(std::min(64, nthreads - 64 * w));
      blockIdx = simt_uint3{(unsigned)x, (unsigned)y, (unsigned)z};
      for (int t = 0; t < nthreads; ++t) {
        Fiber &f = s.fb[t];
        if (!f.stack) f.stack.reset(new char[STACK]);
        f.done = false;
        f.tidx = simt_uint3{(unsigned)t, 0, 0};
        f.xchg = 0;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.get();
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_entry, 0);
      }
      s.cur = 0;
      swapcontext(&s.main_ctx, &s.fb[0].ctx);  // returns when the last fiber of the workgroup has finished
    }
  s.blk = nullptr;
  s.body = nullptr;
}
#endif
inline void launch_block(int nthreads, const std::function<void()> &body) { launch_grid(1, 1, nthreads, body); }
}  // namespace simt

// =================================================================================================
// Host runtime subset: enough of the HIP runtime API for csrc/dhqr_api.hip to compile with the host
// compiler into an EMULATED library (tests/test_emulated_library.py).  "Device" memory is host memory,
// every stream executes immediately in host call order (a valid schedule: the library records events before
// it waits on them), a kernel launch runs the grid on the SIMT emulator above.  This exists to exercise the
// HOST logic of the library (drivers, look-ahead bookkeeping, fast path / retry / fallback decisions,
// workspaces) on tiny problems on the CPU.  It is never shipped, is not loadable by the product package,
// and is not a fallback.
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct simt_stream *hipStream_t;
struct simt_event { double t_ms; };
typedef simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; int cooperativeLaunch; };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// SIMT_DEVICES=N: N emulated "devices" (one address space); lets dhqr_mg_* put its ranks on distinct devices, which the
// RCCL transport requires (tests/simt/fake/rccl/fake_rccl.cpp stands in for librccl there)
inline hipError_t hipGetDeviceCount(int *n) {
  const char *e = std::getenv("SIMT_DEVICES");
  *n = (e && std::atoi(e) > 0) ? std::atoi(e) : 1;
  return hipSuccess;
}
inline thread_local int simt_cur_device = 0;
inline hipError_t hipSetDevice(int d) { simt_cur_device = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = simt_cur_device; return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  std::strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");  // what the emulated kernels are written for
  p->multiProcessorCount = 256;
  p->cooperativeLaunch = 0;  // workgroups of a grid run one after the other here: nothing that needs all of them resident
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t bytes) {
  *p = std::aligned_alloc(256, (bytes + 255) / 256 * 256 + 256);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
typedef void (*hipHostFn_t)(void *);
inline hipError_t hipLaunchHostFunc(hipStream_t, hipHostFn_t fn, void *arg) { fn(arg); return hipSuccess; }  // streams run synchronously here
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memmove(d, s, n);
  return hipSuccess;
}
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipMemcpyToSymbol(void *sym, const void *s, size_t n) { std::memmove(sym, s, n); return hipSuccess; }
inline hipError_t hipMemcpyFromSymbol(void *d, const void *sym, size_t n) { std::memmove(d, sym, n); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1, hipDeviceAttributeMultiprocessorCount = 2 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) {
  *v = (a == hipDeviceAttributeMultiprocessorCount) ? 4 : 100000;
  return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height,
                                   hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) std::memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline double simt_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new simt_event{0.0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t_ms = simt_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }

// kernel launch: the grid runs to completion on the emulator before the call returns.  One workgroup runs at a
// time (static LDS, one scheduler), so launches from several host threads (the rank threads of dhqr_mg_*) are
// serialised by a process-wide mutex: every host thread sees its own launches complete in program order, which
// is a valid execution of its streams.
inline std::mutex &simt_launch_mutex() { static std::mutex m; return m; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                   \
  do {                                                                                                 \
    const dim3 simt_g_ = (grid), simt_b_ = (block);                                                    \
    (void)(stream);                                                                                    \
    std::lock_guard<std::mutex> simt_lk_(simt_launch_mutex());                                         \
    simt::launch_grid((int)simt_g_.x, (int)simt_g_.y, (int)simt_b_.x, [&] { kernel(__VA_ARGS__); }, (int)simt_g_.z);   \
  } while (0)
