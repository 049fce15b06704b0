// dhqr_bench.h -- micro-benchmarks and the MFMA layout probe (include/dhqr_bench.h).  NOT part of the product library:
// compiled only into libdhqr_bench.so (-DDHQR_BENCH_BUILD), a superset of libdhqr.so that bench.py's diagnostics, the
// tools/ scripts and tests/test_gpu_kernels.py load for these calls.  Included by dhqr_api.hip outside extern "C".
#pragma once

// Raw MFMA layout probe (test hook): out[lane*4 + g] = D register g of lane, with
// A[i][k] = a[i*4+k], B[k][j] = b[k*16+j] loaded per the operand maps documented above, C = 0.
__global__ void k_mfma_probe(const double *__restrict__ a, const double *__restrict__ b,
                             double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i16 = lane & 15, k4 = lane >> 4;
  dhqr_d4 acc = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
  acc = mfma_f64(a[i16 * 4 + k4], b[k4 * 16 + i16], acc);
  for (int g = 0; g < 4; ++g) out[lane * 4 + g] = acc[g];
}

// FP64 MFMA issue-rate micro-benchmark: every wave runs `iters` x 16 independent accumulators.
__global__ __launch_bounds__(256) void k_mfma_bench(double *__restrict__ out, int iters) {
  dhqr_d4 acc[16];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = mfma_f64(a, b, acc[x]);
  }
  double s = 0.0;
#pragma unroll
  for (int x = 0; x < 16; ++x) s += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
  out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Issue-rate probes in SHADER cycles (s_memtime), independent of DVFS: per wave, `iters` x 16
// independent v_mfma_f64_16x16x4_f64 (kind 0) or v_fma_f64 (kind 1) chains; cyc[wave] = cycles.
template <int KIND>
__global__ __launch_bounds__(256) void k_issue_probe(double *__restrict__ sink,
                                                     long long *__restrict__ cyc, int iters) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  long long t0, t1;
  double s = 0.0;
  if constexpr (KIND == 0) {
    dhqr_d4 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[x] = mfma_f64(a, b, acc[x]);
    }
    t1 = clock64();
#pragma unroll
    for (int x = 0; x < 16; ++x) s += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
  } else {
    double acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = threadIdx.x * 1e-3 + x;
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[x] = fma(acc[x], a, b);
    }
    t1 = clock64();
#pragma unroll
    for (int x = 0; x < 16; ++x) s += acc[x];
  }
  sink[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// Second probe: several waves per SIMD and MFMA/VALU co-issue.  mode 0: every wave MFMA; mode 1:
// every wave v_fma_f64; mode 2: waves 0-3 of the workgroup MFMA, the others VALU (blockDim 512:
// one MFMA wave + one VALU wave per SIMD).  8 independent chains per wave (low register use, so
// blockDim up to 1024 = 4 waves per SIMD fits).
__global__ __launch_bounds__(1024) void k_issue_probe2(double *__restrict__ sink,
                                                       long long *__restrict__ cyc, int iters,
                                                       int mode) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode == 2 && wave < 4);
  long long t0, t1;
  double s = 0.0;
  if (do_mfma) {
    dhqr_d4 acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) acc[x] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int x = 0; x < 8; ++x) acc[x] = mfma_f64(a, b, acc[x]);
    }
    t1 = clock64();
#pragma unroll
    for (int x = 0; x < 8; ++x) s += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
  } else {
    double acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = threadIdx.x * 1e-3 + x;
    t0 = clock64();
    for (int it = 0; it < iters * 8; ++it) {  // 16 FMA per trip: ~same duration as the MFMA waves
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[x] = fma(acc[x], a, b);
    }
    t1 = clock64();
#pragma unroll
    for (int x = 0; x < 16; ++x) s += acc[x];
  }
  sink[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// Shader-clock probe: one wave sleeps for `wall_ticks` ticks of the constant-rate counter (wall_clock64) and reports
// how many shader cycles (s_memtime) passed: launched beside a GEMM on a second stream it gives the clock the chip
// sustains under that kernel (the chip clocks to its power budget).  out = {shader cycles, wall ticks}.
__global__ void k_clock_probe(long long *__restrict__ out, long long wall_ticks) {
  const long long w0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - w0 < wall_ticks) __builtin_amdgcn_s_sleep(64);
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
}

// MFMA cadence probe: the inner loop of the GEMM kernels (4 x 4 MFMA tiles per wave, fragments from LDS) without
// staging, barriers or global memory: cycles per MFMA per wave (s_memtime).  MODE 0: register operands only;
// 1: fragments by ds_read from the k-contiguous layout, stride S = 18 doubles (what k_gemm_tn* use; the compiler
// merges the kk / kk+1 reads into ds_read2_b64); 2: same layout, one opaque base per kk (plain ds_read_b64 only);
// 3: stride 17 (odd: conflict-free for ds_read2_b64's 16-lane groups); 4: the NN kernel's operands (V tile
// row-contiguous with stride 144, W tile stride 18).  blockDim 256 (one wave per SIMD) or 512 (two).
template <int MODE>
__global__ __launch_bounds__(512) void k_mma_probe(double *__restrict__ sink, long long *__restrict__ cyc, int iters) {
  constexpr int S = (MODE == 3) ? 17 : G_LDK;
  __shared__ double Vs[256 * 19];
  __shared__ double Cs[128 * 19];
  __shared__ double Vr[G_KT * G_LDR];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int i16 = lane & 15, k4 = lane >> 4;
  for (int e = t; e < 256 * 19; e += blockDim.x) Vs[e] = 1.0 + 1e-6 * e;
  for (int e = t; e < 128 * 19; e += blockDim.x) Cs[e] = 1.0 - 1e-6 * e;
  for (int e = t; e < G_KT * G_LDR; e += blockDim.x) Vr[e] = 0.5 + 1e-6 * e;
  __syncthreads();
  const int wc = (w >> 2) & 1, wp = w & 3;
  dhqr_d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (dhqr_d4){0.0, 0.0, 0.0, 0.0};
  const double *cs = &Cs[(wc * 64 + i16) * S + k4];
  const double *vs = (MODE == 4) ? &Vr[k4 * G_LDR + (wp & 1) * 64 + i16] : &Vs[(wp * 64 + i16) * S + k4];
  int offc[4], offv[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    offc[kk] = kk * 4;
    offv[kk] = (MODE == 4) ? kk * 4 * G_LDR : kk * 4;
    if (MODE == 2) {  // opaque offsets: the kk and kk+1 reads cannot be paired into ds_read2_b64
      asm volatile("" : "+v"(offc[kk]));
      asm volatile("" : "+v"(offv[kk]));
    }
  }
  double ra[4] = {1.0 + lane * 1e-9, 1.1, 1.2, 1.3}, rb[4] = {1.0 - lane * 1e-9, 0.9, 0.8, 0.7};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");  // the fragments are re-read every iteration, as in the GEMM kernels
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (MODE == 0) {
          a[x] = ra[x];
          b[x] = rb[x];
        } else {
          a[x] = cs[offc[kk] + x * 16 * S];
          b[x] = (MODE == 4) ? vs[offv[kk] + x * 16] : vs[offv[kk] + x * 16 * S];
        }
      }
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) acc[ci][pi] = mfma_f64(a[ci], b[pi], acc[ci][pi]);
    }
  }
  const long long t1 = clock64();
  double sum = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  sink[(int64_t)blockIdx.x * blockDim.x + t] = sum;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
}

// streaming read+write micro-benchmark (y = x + 1 on double2)
__global__ __launch_bounds__(256) void k_stream_bench(const double2 *__restrict__ x,
                                                      double2 *__restrict__ y, int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += stride) {
    double2 v = x[e];
    v.x += 1.0;
    v.y += 1.0;
    y[e] = v;
  }
}

extern "C" {
int32_t dhqr_bench_mfma_f64(dhqr_ctx *c, double *tflops) {
  ENTER(c);
  if (!tflops) return set_err(DHQR_EINVAL, "null output");
  const int nblk = 256 * 8, iters = 4000;
  CHECK(ensure(c, c->scratch, (size_t)nblk * 256 + 4096));
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_mfma_bench, dim3(nblk), dim3(256), 0, c->stream, c->scratch.p, 100);
  HIPCHECK(hipEventRecord(a, c->stream));
  hipLaunchKernelGGL(k_mfma_bench, dim3(nblk), dim3(256), 0, c->stream, c->scratch.p, iters);
  HIPCHECK(hipEventRecord(b, c->stream));
  HIPCHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  const double flops = (double)nblk * 4.0 * (double)iters * 16.0 * 2048.0;
  *tflops = flops / ((double)ms * 1e-3) / 1e12;
  return DHQR_OK;
}

int32_t dhqr_bench_issue_f64(dhqr_ctx *c, int32_t kind, int32_t nblocks, double *cycles_per_instr,
                             double *tflops) {
  ENTER(c);
  if (!cycles_per_instr || !tflops || nblocks <= 0 || nblocks > 4096 || (kind != 0 && kind != 1))
    return set_err(DHQR_EINVAL, "bad arguments");
  const int iters = 2000;
  CHECK(ensure(c, c->scratch, (size_t)nblocks * 256 + 4096 + (size_t)nblocks * 4 + 16));
  double *sink = c->scratch.p;
  long long *cyc = (long long *)(c->scratch.p + (size_t)nblocks * 256 + 4096);
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {  // first pass warms clocks / code
    HIPCHECK(hipEventRecord(a, c->stream));
    if (kind == 0) hipLaunchKernelGGL((k_issue_probe<0>), dim3(nblocks), dim3(256), 0, c->stream, sink, cyc, iters);
    else hipLaunchKernelGGL((k_issue_probe<1>), dim3(nblocks), dim3(256), 0, c->stream, sink, cyc, iters);
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  std::vector<long long> h((size_t)nblocks * 4);
  HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (long long v : h) sum += (double)v;
  *cycles_per_instr = sum / (double)h.size() / ((double)iters * 16.0);
  const double flop_per_instr = kind == 0 ? 2048.0 : 128.0;
  *tflops = (double)nblocks * 4.0 * iters * 16.0 * flop_per_instr / ((double)ms * 1e-3) / 1e12;
  return DHQR_OK;
}

int32_t dhqr_bench_issue2_f64(dhqr_ctx *c, int32_t mode, int32_t threads, int32_t nblocks, double *out4) {
  ENTER(c);
  if (!out4 || nblocks <= 0 || nblocks > 4096 || mode < 0 || mode > 2 || threads % 256 || threads > 1024)
    return set_err(DHQR_EINVAL, "bad arguments");
  const int iters = 1000, wpb = threads / 64;
  CHECK(ensure(c, c->scratch, (size_t)nblocks * threads + 4096 + (size_t)nblocks * wpb + 16));
  double *sink = c->scratch.p;
  long long *cyc = (long long *)(c->scratch.p + (size_t)nblocks * threads + 4096);
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCHECK(hipEventRecord(a, c->stream));
    hipLaunchKernelGGL(k_issue_probe2, dim3(nblocks), dim3(threads), 0, c->stream, sink, cyc, iters, (int)mode);
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  std::vector<long long> h((size_t)nblocks * wpb);
  HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double sm = 0, sv = 0;
  int64_t nm = 0, nv = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const int wave = (int)(i % wpb);
    const bool mf = (mode == 0) || (mode == 2 && wave < 4);
    if (mf) { sm += (double)h[i]; nm++; } else { sv += (double)h[i]; nv++; }
  }
  out4[0] = nm ? sm / nm / (iters * 8.0) : 0.0;          // cycles per MFMA per wave
  out4[1] = nv ? sv / nv / (iters * 8.0 * 16.0) : 0.0;   // cycles per v_fma_f64 per wave
  out4[2] = (double)nm * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
  out4[3] = (double)nv * iters * 8.0 * 16.0 * 128.0 / (ms * 1e-3) / 1e12;
  return DHQR_OK;
}

int32_t dhqr_bench_stream_f64(dhqr_ctx *c, int64_t bytes, double *gbps) {
  ENTER(c);
  if (!gbps || bytes < 4096) return set_err(DHQR_EINVAL, "bad arguments");
  const int64_t n2 = bytes / 16;
  double *x = nullptr, *y = nullptr;
  if (hipMalloc((void **)&x, (size_t)n2 * 16) != hipSuccess || hipMalloc((void **)&y, (size_t)n2 * 16) != hipSuccess) {
    if (x) (void)hipFree(x);
    return set_err(DHQR_ENOMEM, "hipMalloc failed in dhqr_bench_stream_f64");
  }
  (void)hipMemsetAsync(x, 0, (size_t)n2 * 16, c->stream);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const unsigned grid = 256 * 16;
  hipLaunchKernelGGL(k_stream_bench, dim3(grid), dim3(256), 0, c->stream, (const double2 *)x, (double2 *)y, n2);
  (void)hipEventRecord(a, c->stream);
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_stream_bench, dim3(grid), dim3(256), 0, c->stream, (const double2 *)x, (double2 *)y, n2);
  (void)hipEventRecord(b, c->stream);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(x);
  (void)hipFree(y);
  *gbps = 2.0 * (double)n2 * 16.0 * reps / ((double)ms * 1e-3) / 1e9;
  return DHQR_OK;
}

// GEMM micro-benchmark of the two wide trailing-update kernels on synthetic operands (not a product entry point):
// kind 0: k_gemm_nn_sub<2,256> (C -= [V_a V_b] W, rows x ncols), kind 1: k_gemm_tn2 + its split-K reduction as the driver launches them (pair_vtc: Y = [V_a V_b]' C), kind 2: k_gemm_nn_quad (K = 512).
// `reps` timed launches after one warm-up; a one-wave clock probe runs beside them on a second stream.
// out = {ms per launch, TFLOP/s, shader MHz under the kernel, 0}.  The A/B switches of the context apply.
int32_t dhqr_bench_gemm_f64(dhqr_ctx *c, int32_t kind, int64_t rows, int64_t ncols, int32_t reps, double *out4) {
  ENTER(c);
  if (!out4 || rows < 256 || ncols < 128 || rows % 128 || ncols % 128 || reps < 1 || kind < 0 || kind > 2)
    return set_err(DHQR_EINVAL, "bad arguments");
  const int64_t ldv = rows, ldc = rows, ld2 = (kind == 2 ? 4 : 2) * DHQR_NBV;  // kind 2: k_gemm_nn_quad (K = 512)
  double *V = nullptr, *W = nullptr, *C = nullptr, *Y = nullptr;
  long long *clk = nullptr;
  hipStream_t s2 = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&V, (size_t)ldv * ld2 * 8));
    HIPCHECK(hipMalloc((void **)&W, (size_t)ld2 * ncols * 8));
    HIPCHECK(hipMalloc((void **)&C, (size_t)ldc * ncols * 8));
    HIPCHECK(hipMalloc((void **)&clk, 64));
    HIPCHECK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1));
    HIPCHECK(hipEventCreate(&a));
    HIPCHECK(hipEventCreate(&b));
    CHECK(dhqr_fill_uniform_f64(c, V, rows, ld2, ldv, 1, rows, 0, 128, 1, 0));
    CHECK(dhqr_fill_uniform_f64(c, W, ld2, ncols, ld2, 2, ld2, 0, 128, 1, 0));
    CHECK(dhqr_fill_uniform_f64(c, C, rows, ncols, ldc, 3, rows, 0, 128, 1, 0));
    const int64_t ntiles = ncols / 128, gx = rows / 128;
    int64_t nsplit = 1, rps = rows;
    if (kind == 1) {  // through pair_vtc: the decomposition the driver ships (stream-K for wide launches) + its reduction
      (void)nsplit;
      (void)rps;
      HIPCHECK(hipMalloc((void **)&Y, (size_t)ld2 * ncols * 8));
      CHECK(ensure(c, c->ws[c->cur_ws].w1, (size_t)16 * ld2 * ncols));
    }
    bool timed_nn = false;
    auto launch = [&]() {
      if (kind == 2) {
        const int swz = (gx >= 16 && ntiles >= 16) ? 1 : 0;
        dim3 grid((unsigned)gx, (unsigned)ntiles);
        if (swz) grid = dim3((unsigned)((((gx + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
        hipLaunchKernelGGL((k_gemm_nn_quad<2, 128>), grid, dim3(256), 0, c->stream, (const double *)V, (const double *)(V + 256 * ldv),
                           ldv, (int64_t)0, (const double *)W, ld2, C, ldc, rows, ncols, swz, (const int *)nullptr, 0);
      } else if (kind == 0) {
        const int swz = (gx >= 16 && ntiles >= 16) ? 1 : 0;
        dim3 grid((unsigned)gx, (unsigned)ntiles);
        if (swz) grid = dim3((unsigned)((((gx + 7) / 8) * ((ntiles + 7) / 8) + 7) / 8 * 512), 1);
        if (timed_nn)
          hipLaunchKernelGGL((k_gemm_nn_sub<2, 256, false, true>), grid, dim3(256), 0, c->stream, (const double *)V, ldv,
                             (const double *)W, ld2, C, ldc, rows, ncols, swz, (const int *)nullptr, 0);
        else
          launch_nn_sub<256>(c, true, grid, V, ldv, W, ld2, C, ldc, rows, ncols, swz, false);
      } else {
        const int keep = c->tn_spare;  // the kernel ALONE: no CUs left to a lane that is not running (wide_slots)
        c->tn_spare = 0;
        (void)pair_vtc(c, V, ldv, rows, C, ldc, ncols, true, Y);
        c->tn_spare = keep;
      }
    };
    launch();
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (kind == 0 && getenv("DHQR_NN_TIME")) {  // phase clock of one launch (instrumented instantiation)
      unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_nn_phase), z, sizeof(z)));
      timed_nn = true;
      launch();
      timed_nn = false;
      HIPCHECK(hipStreamSynchronize(c->stream));
      HIPCHECK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_nn_phase), sizeof(z)));
      const double nt = (double)std::max<unsigned long long>(1, z[4]);
      fprintf(stderr, "k_gemm_nn_sub phase clock (wave 0, cycles per tile over %.0f tiles): prologue + C tile %.0f, K loop %.0f, "
                      "store issue %.0f, store drain %.0f\n", nt, z[0] / nt, z[1] / nt, z[2] / nt, z[3] / nt);
    }
    HIPCHECK(hipEventRecord(a, c->stream));
    launch();
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
    float ms1 = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms1, a, b));
    // clock probe for about 60 % of the timed region, started right behind the first timed launch
    int wall_khz = 100000;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device);
    const long long ticks = (long long)(0.6 * ms1 * reps * wall_khz);
    HIPCHECK(hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; ++r) {
      launch();
      if (r == 0) hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s2, clk, ticks);
    }
    HIPCHECK(hipEventRecord(b, c->stream));
    HIPCHECK(hipEventSynchronize(b));
    HIPCHECK(hipStreamSynchronize(s2));
    LAUNCHCHECK();
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, a, b));
    long long h[2] = {0, 0};
    HIPCHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    out4[0] = ms / reps;
    out4[1] = 2.0 * (double)ld2 * (double)rows * (double)ncols / (out4[0] * 1e-3) / 1e12;
    out4[2] = h[1] > 0 ? (double)h[0] / (double)h[1] * (double)wall_khz * 1e-3 : 0.0;
    out4[3] = 0.0;
    return DHQR_OK;
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(c->stream);
  if (s2) (void)hipStreamDestroy(s2);
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(V); (void)hipFree(W); (void)hipFree(C); (void)hipFree(Y); (void)hipFree(clk);
  return rc;
}

// MFMA cadence probe (k_mma_probe<mode>): 256 workgroups of `threads` (256 / 512) threads; out2 = {mean cycles per
// MFMA per wave, wall TFLOP/s}.
int32_t dhqr_bench_mma_probe_f64(dhqr_ctx *c, int32_t mode, int32_t threads, double *out2) {
  ENTER(c);
  if (!out2 || (threads != 256 && threads != 512) || mode < 0 || mode > 4) return set_err(DHQR_EINVAL, "bad arguments");
  const int nblk = 256, iters = 400, nw = threads / 64;
  double *sink = nullptr;
  long long *cyc = nullptr;
  HIPCHECK(hipMalloc((void **)&sink, (size_t)nblk * threads * 8));
  HIPCHECK(hipMalloc((void **)&cyc, (size_t)nblk * nw * 8));
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  auto launch = [&](int it) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(k_mma_probe<0>, dim3(nblk), dim3(threads), 0, c->stream, sink, cyc, it); break;
      case 1: hipLaunchKernelGGL(k_mma_probe<1>, dim3(nblk), dim3(threads), 0, c->stream, sink, cyc, it); break;
      case 2: hipLaunchKernelGGL(k_mma_probe<2>, dim3(nblk), dim3(threads), 0, c->stream, sink, cyc, it); break;
      case 3: hipLaunchKernelGGL(k_mma_probe<3>, dim3(nblk), dim3(threads), 0, c->stream, sink, cyc, it); break;
      default: hipLaunchKernelGGL(k_mma_probe<4>, dim3(nblk), dim3(threads), 0, c->stream, sink, cyc, it); break;
    }
  };
  launch(20);
  HIPCHECK(hipEventRecord(a, c->stream));
  launch(iters);
  HIPCHECK(hipEventRecord(b, c->stream));
  HIPCHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<long long> h((size_t)nblk * nw);
  HIPCHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double tot = 0.0;
  for (long long v : h) tot += (double)v;
  out2[0] = tot / (double)h.size() / ((double)iters * 64.0);
  out2[1] = (double)nblk * nw * (double)iters * 64.0 * 2048.0 / ((double)ms * 1e-3) / 1e12;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(sink);
  (void)hipFree(cyc);
  return DHQR_OK;
}

// ---- feasibility probe for a look-ahead lane that CO-RESIDES with the wide subtraction (tools/thin_lane_probe.py) --------
// Two k_gemm_nn_quad workgroups fill the register file of every SIMD (2 x 4 waves x 256 VGPRs) and 140 of the CU's 160 KB of
// LDS; a slot frees when one of them retires: 4 waves x 256 registers (or 16 waves x 64) and ~90 KB of LDS.  Could the lane's
// kernels be placed in such slots while the wide launch runs, and at what price?  This entry point runs, on the context's
// HIGH-PRIORITY stream, `reps` times: (1) a Gram product of a rows x 128 panel by k_gemm_tn with `nsplit` workgroups (the
// footprint of one subtraction workgroup: 256 threads, 72 KB of LDS) + its reduction; (2) a stand-in for the single-workgroup
// panel kernels: ONE workgroup of 1024 threads, few registers, `lds_kb` KB of dynamic LDS, 128 dependent barrier steps
// (LDS write -> barrier -> LDS read -> a few FMAs).  out4 = {ms per Gram + reduction, ms per stand-in, total ms per repetition,
// 0}.  The caller runs it alone and beside dhqr_bench_gemm_f64 on ANOTHER context.
// NR live doubles per thread: 8 (~30 VGPRs: 16 waves fit the 256 registers per SIMD one retiring subtraction workgroup frees)
// or 40 (~90 VGPRs: 16 waves need 360 per SIMD, i.e. BOTH subtraction workgroups of a CU gone -- k_panel_top / k_build_t today)
extern "C++" {
template <int NR>
__global__ __launch_bounds__(1024) void k_lane_standin(double *__restrict__ out, int steps) {
  extern __shared__ double lsm[];
  const int t = threadIdx.x;
  double acc[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) acc[i] = 1.0 + 1e-9 * (t + i);
  for (int s = 0; s < steps; ++s) {
    double *row = lsm + (s & 1) * 1024;
    double w = 0.0;
#pragma unroll
    for (int i = 0; i < NR; i += 8) w += acc[i];
    row[t] = w;
    __syncthreads();
    const double x = row[(t + 1 + s) & 1023], y = row[(t * 7 + s) & 1023];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = fma(acc[i], 0.999999, x * 1e-9 + y * 1e-10 * (i + 1));
  }
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < NR; ++i) r += acc[i];
  if (r == 12345.678) out[t] = r;  // keep the chain alive
}
}  // extern "C++"
int32_t dhqr_bench_lane_probe_f64(dhqr_ctx *c, int64_t rows, int32_t nsplit, int32_t lds_kb_in, int32_t reps, double *out4) {
  ENTER(c);
  const bool real = lds_kb_in == 0;  // 0: the REAL single-workgroup kernels, k_panel_top (on the Gram matrix just formed) + k_build_t
  const bool heavy = lds_kb_in < 0;  // negative: the register-heavy stand-in (~90 VGPRs) with |lds_kb| KB of LDS
  const int32_t lds_kb = real ? 64 : (heavy ? -lds_kb_in : lds_kb_in);
  if (!out4 || rows < 1024 || rows % 16 || nsplit < 1 || nsplit > 1024 || lds_kb < 16 || lds_kb > 160 || reps < 1)
    return set_err(DHQR_EINVAL, "bad arguments");
  const int64_t NB = DHQR_NBV, ldp = rows;
  double *P = nullptr, *part = nullptr, *G = nullptr, *sink = nullptr, *scr = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  hipStream_t st = c->hi;
  auto body = [&]() -> int32_t {
    HIPCHECK(hipMalloc((void **)&P, (size_t)ldp * NB * 8));
    HIPCHECK(hipMalloc((void **)&part, (size_t)nsplit * NB * NB * 8));
    HIPCHECK(hipMalloc((void **)&G, (size_t)NB * NB * 8));
    HIPCHECK(hipMalloc((void **)&sink, 1024 * 8));
    HIPCHECK(hipMalloc((void **)&scr, (size_t)(4096 + 4 * NB * NB) * 8));
    for (auto &e : ev) HIPCHECK(hipEventCreate(&e));
    CHECK(dhqr_fill_uniform_f64(c, P, rows, NB, ldp, 5, rows, 0, 128, 1, 0));
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_standin<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_standin<40>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    int64_t rps = (rows + nsplit - 1) / nsplit;
    rps = (rps + G_KT - 1) / G_KT * G_KT;
    const int64_t ns = (rows + rps - 1) / rps;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int r = 0; r <= reps; ++r) {  // r == 0: warm-up
      HIPCHECK(hipEventRecord(ev[0], st));
      hipLaunchKernelGGL((k_gemm_tn<2, 1, 128>), dim3(1, (unsigned)ns), dim3(256), 0, st, (const double *)P, ldp, (const double *)P, ldp, 1,
                         (int64_t)0, rows, NB, rps, part, NB, NB * NB);
      hipLaunchKernelGGL(k_reduce_splits, dim3((unsigned)(NB * NB / 64)), dim3(256), 0, st, (const double *)part, (int)ns, NB * NB, NB * NB, G);
      HIPCHECK(hipEventRecord(ev[1], st));
      if (real) {
        hipLaunchKernelGGL((k_panel_top<false>), dim3(1), dim3(1024), 0, st, (const double *)G, (const double *)P, ldp, scr, scr + 1024,
                           scr + 1024 + NB * NB, (int *)(scr + 1024 + 2 * NB * NB));
        hipLaunchKernelGGL(k_build_t, dim3(1), dim3(1024), 0, st, (const double *)G, (int)NB, scr + 2048 + 2 * NB * NB, scr + 2048 + 3 * NB * NB,
                           0.0, (int *)nullptr, 0, (double *)nullptr, (double *)nullptr);
      } else if (heavy)
        hipLaunchKernelGGL(k_lane_standin<40>, dim3(1), dim3(1024), (size_t)lds_kb * 1024, st, sink, 128);
      else
        hipLaunchKernelGGL(k_lane_standin<8>, dim3(1), dim3(1024), (size_t)lds_kb * 1024, st, sink, 128);
      HIPCHECK(hipEventRecord(ev[2], st));
      HIPCHECK(hipEventSynchronize(ev[2]));
      float a = 0.f, b = 0.f;
      HIPCHECK(hipEventElapsedTime(&a, ev[0], ev[1]));
      HIPCHECK(hipEventElapsedTime(&b, ev[1], ev[2]));
      if (r > 0) {
        acc[0] += a;
        acc[1] += b;
        acc[2] += a + b;
      }
    }
    LAUNCHCHECK();
    for (int i = 0; i < 3; ++i) out4[i] = acc[i] / reps;
    out4[3] = 0.0;
    return DHQR_OK;
  };
  const int32_t rc = body();
  (void)hipStreamSynchronize(st);
  for (auto &e : ev)
    if (e) (void)hipEventDestroy(e);
  double *ps[] = {P, part, G, sink, scr};
  for (double *p_ : ps)
    if (p_) (void)hipFree(p_);
  return rc;
}

// test hook (not in dhqr.h's stable surface, declared in the test binding only):
// raw MFMA D registers for the documented operand maps
int32_t dhqr_debug_mfma_probe(dhqr_ctx *c, const double *da, const double *db, double *dout) {
  ENTER(c);
  hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, c->stream, da, db, dout);
  LAUNCHCHECK();
  HIPCHECK(hipStreamSynchronize(c->stream));
  return DHQR_OK;
}

// ---- test hook: hold compute units (include/dhqr_bench.h) ---------------------------------------------------------------
extern "C++" {
// 1024 threads + 150 KB of LDS: one workgroup per CU, and nothing else fits beside it.  Spins on a word of pinned host
// memory; gives up after `max_cycles` of the constant 100 MHz counter (s_memrealtime) whatever the host does.
__global__ __launch_bounds__(1024) void k_hold_cu(volatile int *release, unsigned long long max_ticks, int *sink) {
  __shared__ int ballast[150 * 1024 / 4];
  if (threadIdx.x == 0) ballast[blockIdx.x & 1023] = 1;
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    while (*release == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(64);
  }
  __syncthreads();
  if (ballast[threadIdx.x] == 12345 && sink) sink[0] = 1;  // keep the LDS allocation alive
}
static hipStream_t g_hold_stream = nullptr;
static int *g_hold_flag = nullptr;  // pinned host word
}  // extern "C++"
int32_t dhqr_debug_hold_cus(dhqr_ctx *c, int32_t nwg, int32_t max_ms, int32_t release) {
  ENTER(c);
  if (!g_hold_stream) {
    HIPCHECK(hipStreamCreateWithFlags(&g_hold_stream, hipStreamNonBlocking));
    HIPCHECK(hipHostMalloc((void **)&g_hold_flag, 64, hipHostMallocDefault));
  }
  if (release) {
    *(volatile int *)g_hold_flag = 1;
    HIPCHECK(hipStreamSynchronize(g_hold_stream));
    return DHQR_OK;
  }
  if (nwg < 1 || nwg > 1024 || max_ms < 1 || max_ms > 5000) return set_err(DHQR_EINVAL, "bad arguments");
  *(volatile int *)g_hold_flag = 0;
  hipLaunchKernelGGL(k_hold_cu, dim3((unsigned)nwg), dim3(1024), 0, g_hold_stream, (volatile int *)g_hold_flag,
                     (unsigned long long)max_ms * 100000ull, (int *)nullptr);  // wall_clock64: 100 MHz
  LAUNCHCHECK();
  return DHQR_OK;
}
// phase clock of the last k_small_qr_d launches since the previous call (dhqr_small.h: g_smq_phase); out: 9 x 6 values
int32_t dhqr_debug_smq_phases(dhqr_ctx *c, double *out54) {
  ENTER(c);
  unsigned long long h[9][6];
  HIPCHECK(hipStreamSynchronize(c->stream));
  HIPCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_smq_phase), sizeof(h)));
  for (int w = 0; w < 9; ++w)
    for (int q = 0; q < 6; ++q) out54[w * 6 + q] = (double)h[w][q];
  memset(h, 0, sizeof(h));
  HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_smq_phase), h, sizeof(h)));
  return DHQR_OK;
}

}  // extern "C"
