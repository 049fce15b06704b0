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

namespace simt {
// run `body` (a kernel call) once per thread of a single workgroup of `nthreads` threads
inline void launch_block(int nthreads, const std::function<void()> &body, unsigned bx = 0, unsigned gx = 1) {
  Block blk;
  blk.bar.reset(nthreads);
  blk.waves = std::vector<Wave>((nthreads + 63) / 64);
  for (int w = 0; w < (int)blk.waves.size(); ++w) blk.waves[w].bar.reset(std::min(64, nthreads - 64 * w));
  cur_block() = &blk;
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      tl_tid = t;
      threadIdx = simt_uint3{(unsigned)t, 0, 0};
      blockIdx = simt_uint3{bx, 0, 0};
      blockDim = dim3(nthreads);
      gridDim = dim3(gx);
      body();
      blk.waves[t >> 6].bar.leave();
      blk.bar.leave();
    });
  for (auto &x : th) x.join();
  cur_block() = nullptr;
}
}  // namespace simt
