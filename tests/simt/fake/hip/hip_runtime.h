// tests/simt/fake/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a one-workgroup SIMT emulator.
//
// Put on the include path IN FRONT of ROCm's headers, it lets the UNMODIFIED single-workgroup kernel
// sources of csrc/ (dhqr_recon.h ...) compile with the host clang++ and run on the CPU: every HIP
// thread of the workgroup is an OS thread, __syncthreads() is a barrier over the live threads of the
// block, __shfl/__shfl_xor exchange through a per-wavefront buffer (all 64 lanes must call, as on the
// hardware for a full-EXEC shuffle), `__shared__` becomes `static` (one workgroup runs at a time).
// Built with -fsanitize=thread this turns a missing barrier / LDS double-buffering mistake into a
// reported data race.  It never ships and is never a fallback: only tests/test_simt_emulation.py
// compiles it; the product is libdhqr.so (HIP, gfx950).
#pragma once
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

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
// barrier whose participant count shrinks when a thread's kernel function returns
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int live = 0, waiting = 0;
  uint64_t gen = 0;
  void reset(int n) { live = n; waiting = 0; gen = 0; }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const uint64_t g = gen;
    if (++waiting == live) { waiting = 0; ++gen; cv.notify_all(); return; }
    cv.wait(lk, [&] { return gen != g; });
  }
  void leave() {
    std::unique_lock<std::mutex> lk(m);
    --live;
    if (live > 0 && waiting == live) { waiting = 0; ++gen; cv.notify_all(); }
  }
};
struct Wave {
  Barrier bar;
  uint64_t buf[64];
  double mfa[64], mfb[64];  // operands of an emulated MFMA
};
struct Block {
  Barrier bar;
  std::vector<Wave> waves;
};
inline Block *&cur_block() { static Block *b = nullptr; return b; }
inline thread_local int tl_tid = 0;
}  // namespace simt

inline thread_local simt_uint3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0};
inline thread_local dim3 blockDim, gridDim;

inline void __syncthreads() { simt::cur_block()->bar.wait(); }

template <typename T>
inline T simt_shfl_bits(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  simt::Wave &w = simt::cur_block()->waves[simt::tl_tid >> 6];
  const int lane = simt::tl_tid & 63;
  uint64_t bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  w.buf[lane] = bits;
  w.bar.wait();
  const uint64_t got = w.buf[src_lane & 63];
  w.bar.wait();  // nobody overwrites buf before every lane has read
  T r;
  std::memcpy(&r, &got, sizeof(T));
  return r;
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  const int lane = simt::tl_tid & 63;
  return simt_shfl_bits(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  const int lane = simt::tl_tid & 63;
  (void)width;
  return simt_shfl_bits(v, lane ^ mask);
}

// ---- v_mfma_f64_16x16x4_f64, wave-synchronous like the instruction: every lane contributes one A and one
// B element, D[i][j] += sum_k A[i][k] B[k][j] with the gfx950 operand maps documented in csrc/dhqr_gemm.h
//   A: lane l holds A[i = l&15][k = l>>4]   B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, register g holds D[i = (l>>4) + 4g][j = l&15]
// (the map itself is pinned on the device by tests/test_gpu_kernels.py::test_mfma_layout_probe)
typedef double simt_d4 __attribute__((ext_vector_type(4)));
inline simt_d4 simt_mfma_f64_16x16x4(double a, double b, simt_d4 c) {
  simt::Wave &w = simt::cur_block()->waves[simt::tl_tid >> 6];
  const int lane = simt::tl_tid & 63;
  w.mfa[lane] = a;
  w.mfb[lane] = b;
  w.bar.wait();
  const int j = lane & 15;
  for (int g = 0; g < 4; ++g) {
    const int i = (lane >> 4) + 4 * g;
    double s = c[g];
    for (int k = 0; k < 4; ++k) s = std::fma(w.mfa[i + 16 * k], w.mfb[j + 16 * k], s);
    c[g] = s;
  }
  w.bar.wait();
  return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) simt_mfma_f64_16x16x4(a, b, c)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline long long clock64() { return 0; }

namespace simt {
// Run a gx x gy grid of INDEPENDENT workgroups of `nthreads` threads, one workgroup after the other.
// The OS threads are created once and walk the grid in lockstep: [fixed barrier] thread 0 re-arms the
// dynamic barriers [fixed barrier] kernel body, leave() -> next workgroup.  The fixed barriers also keep a
// fast thread from touching the (static) LDS arrays of workgroup n+1 while a slow one still reads n.
inline void launch_grid(int gx, int gy, int nthreads, const std::function<void()> &body) {
  Block blk;
  blk.waves = std::vector<Wave>((nthreads + 63) / 64);
  Barrier fence;  // fixed participant count, nobody leaves
  fence.reset(nthreads);
  cur_block() = &blk;
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      tl_tid = t;
      threadIdx = simt_uint3{(unsigned)t, 0, 0};
      blockDim = dim3(nthreads);
      gridDim = dim3(gx, gy);
      for (int y = 0; y < gy; ++y)
        for (int x = 0; x < gx; ++x) {
          fence.wait();
          if (t == 0) {
            blk.bar.reset(nthreads);
            for (int w = 0; w < (int)blk.waves.size(); ++w) blk.waves[w].bar.reset(std::min(64, nthreads - 64 * w));
          }
          fence.wait();
          blockIdx = simt_uint3{(unsigned)x, (unsigned)y, 0};
          body();
          blk.waves[t >> 6].bar.leave();
          blk.bar.leave();
        }
    });
  for (auto &x : th) x.join();
  cur_block() = nullptr;
}
inline void launch_block(int nthreads, const std::function<void()> &body) { launch_grid(1, 1, nthreads, body); }
}  // namespace simt
