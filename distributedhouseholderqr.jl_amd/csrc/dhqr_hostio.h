// dhqr_hostio.h -- the PCIe side of the host-in / host-out drop-in dhqr_qr_f64 (`qr!(A::Matrix)`, src:311-315).
//
// Round 3: hipMemcpy2D of the whole (pageable) matrix up, factorisation, hipMemcpy2D down -- three serial phases, the
// copies at the runtime's single-threaded staging rate: 32768^2 in ~1.5 s against 0.85 s device-resident.  Now:
//   * upload   128-column chunks through PINNED staging buffers on two copy streams: a stream-ordered host function copies
//              the caller's columns into a staging buffer with several threads, an asynchronous DMA takes it to the
//              device; chunk k + 1 is staged while chunk k travels;
//   * download a column block is FINAL once its panel is committed (its reflectors below the diagonal, its rows of R from
//              the earlier steps above), so the blocked driver hands every committed panel's event to a hook
//              (dhqr_ctx::panel_hook) and the block travels device -> staging -> caller behind that event while the
//              later panels are still being factored: the download hides behind the factorisation;
//   * whatever the hook did not cover (the unblocked path, the simple driver of 1-2 panels) or may have caught in an
//              unfinished state (a panel that was rejected on the device and redone by the resume pass) is downloaded
//              at the end.
#pragma once
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define HIO_NSTAGE 4   // staging buffers per direction (two per copy stream)
#define HIO_THREADS 8  // host threads of one staging copy (4: 32 MiB in ~1 ms, the upload was bound by it: 0.25 s for 8 GiB)

struct HioJob {
  const double *src;
  double *dst;
  int64_t lds, ldd, rows, cols;
};
static void hio_copy_cols(void *arg) {  // stream-ordered host function
  const HioJob *j = static_cast<const HioJob *>(arg);
  const int nt = (int)std::min<int64_t>(HIO_THREADS, std::max<int64_t>(1, j->cols / 8));
  auto work = [j](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) memcpy(j->dst + c * j->ldd, j->src + c * j->lds, (size_t)j->rows * sizeof(double));
  };
  if (nt <= 1) {
    work(0, j->cols);
    return;
  }
  std::vector<std::thread> th;
  const int64_t per = (j->cols + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, std::min(j->cols, t * per), std::min(j->cols, (t + 1) * per));
  work(0, std::min(j->cols, per));
  for (auto &t : th) t.join();
}

struct HostIo {
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  double *stage[HIO_NSTAGE] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap = 0;
  // one download in flight
  double *hA = nullptr;
  const double *dA = nullptr;
  int64_t m = 0, n = 0, lda = 0, ldd = 0;
  std::vector<std::unique_ptr<HioJob>> jobs;
  std::vector<char> done;  // per 128-column block: download enqueued
  int64_t use = 0;
};

// ldd: leading dimension of the device copy = of the staging buffers, so that a block of w columns is ONE contiguous range
// on both sides and travels as a 1-D copy (the copy engines; the pitched 2-D copies of round 4 ran as blit KERNELS --
// __amd_rocclr_copyBuffer in the trace -- that competed with the trailing update for CUs: the factorisation stretched from
// 0.83 to ~1.0 s under the download hook).  Staging: four buffers of ldd x min(n, 128) doubles (not m x 128
// whatever n).
static int32_t hio_init(HostIo &h, int64_t ldd, int64_t n) {
  const size_t need = (size_t)ldd * (size_t)std::min<int64_t>(n, DHQR_NBV);
  if (!h.s[0])
    for (int i = 0; i < 2; ++i) {
      HIPCHECK(hipStreamCreateWithFlags(&h.s[i], hipStreamNonBlocking));
      HIPCHECK(hipEventCreateWithFlags(&h.ev[i], hipEventDisableTiming));
    }
  if (need > h.cap) {
    // cap describes ALL four buffers: it is zero from the moment the first one is released until the last new one exists, so
    // a failure part-way (the caller then falls back to the plain copy form) never leaves a non-zero cap over null or
    // mismatched buffers for a later, smaller call to trust
    h.cap = 0;
    for (double *&p : h.stage) {
      double *q = p;
      p = nullptr;
      if (q) HIPCHECK(hipHostFree(q));
    }
    for (double *&p : h.stage) HIPCHECK(hipHostMalloc((void **)&p, need * sizeof(double), hipHostMallocDefault));
    h.cap = need;
  }
  return DHQR_OK;
}
static void hio_free(HostIo &h) {
  for (int i = 0; i < 2; ++i) {
    if (h.s[i]) (void)hipStreamDestroy(h.s[i]);
    if (h.ev[i]) (void)hipEventDestroy(h.ev[i]);
  }
  for (double *p : h.stage)
    if (p) (void)hipHostFree(p);
}

// hA (m x n, ld lda, pageable or not) -> dA (ld ldd); `after`: the stream that will consume dA waits for both copy streams
static int32_t hio_upload(HostIo &h, const double *hA, int64_t m, int64_t n, int64_t lda, double *dA, int64_t ldd, hipStream_t after) {
  CHECK(hio_init(h, ldd, n));
  const int64_t NB = DHQR_NBV, K = (n + NB - 1) / NB;
  for (int64_t k = 0; k < K; ++k) {
    const int si = (int)(k & 1);
    double *st = h.stage[2 * si + (int)((k >> 1) & 1)];
    const int64_t c0 = k * NB, w = std::min<int64_t>(NB, n - c0);
    h.jobs.emplace_back(new HioJob{hA + c0 * lda, st, lda, ldd, m, w});
    HIPCHECK(hipLaunchHostFunc(h.s[si], hio_copy_cols, h.jobs.back().get()));
    HIPCHECK(hipMemcpyAsync(dA + c0 * ldd, st, (size_t)ldd * (size_t)w * sizeof(double), hipMemcpyHostToDevice, h.s[si]));
  }
  for (int i = 0; i < 2; ++i) {
    HIPCHECK(hipEventRecord(h.ev[i], h.s[i]));
    HIPCHECK(hipStreamWaitEvent(after, h.ev[i], 0));
  }
  return DHQR_OK;
}

// columns [c0, c0 + w) of the device matrix -> the caller's matrix, behind `ready` (nullptr: behind nothing)
static int32_t hio_download_block(HostIo &h, int64_t c0, int64_t w, hipEvent_t ready) {
  const int64_t k = h.use++;
  const int si = (int)(k & 1);
  double *st = h.stage[2 * si + (int)((k >> 1) & 1)];
  if (ready) HIPCHECK(hipStreamWaitEvent(h.s[si], ready, 0));
  HIPCHECK(hipMemcpyAsync(st, h.dA + c0 * h.ldd, (size_t)h.ldd * (size_t)w * sizeof(double), hipMemcpyDeviceToHost, h.s[si]));
  h.jobs.emplace_back(new HioJob{st, h.hA + c0 * h.lda, h.ldd, h.lda, h.m, w});
  HIPCHECK(hipLaunchHostFunc(h.s[si], hio_copy_cols, h.jobs.back().get()));
  return DHQR_OK;
}
// the blocked driver's hook: panel x is committed when `ready` completes
static int32_t hio_panel_hook(void *arg, int64_t x, hipEvent_t ready) {
  HostIo &h = *static_cast<HostIo *>(arg);
  const int64_t c0 = x * DHQR_NBV;
  if (c0 >= h.n || h.done[(size_t)x]) return DHQR_OK;
  h.done[(size_t)x] = 1;
  return hio_download_block(h, c0, std::min<int64_t>(DHQR_NBV, h.n - c0), ready);
}
static int32_t hio_drain(HostIo &h) {
  for (int i = 0; i < 2; ++i) HIPCHECK(hipStreamSynchronize(h.s[i]));
  h.jobs.clear();
  return DHQR_OK;
}
