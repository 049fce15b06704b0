/* dhqr_bench.h -- micro-benchmarks and one test hook, exported by libdhqr_bench.so ONLY.
 *
 * libdhqr_bench.so is the same source as libdhqr.so compiled with -DDHQR_BENCH_BUILD: every entry point of dhqr.h plus
 * the ones below (and the instrumented kernel instantiations they launch).  The drop-in library a Julia / Python host
 * binds (libdhqr.so) exports none of them.  Users: bench.py's diagnostic fields, tools/, tests/test_gpu_kernels.py.
 * Contexts are per library: create the dhqr_ctx with the library whose entry points you call. */
#ifndef DHQR_BENCH_H
#define DHQR_BENCH_H
#include "dhqr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ micro-benchmarks
 * Device ceilings measured on the box itself (bench.py reports them next to the spec peaks):
 * FP64 MFMA issue-bound TFLOP/s (v_mfma_f64_16x16x4_f64 only) and a read+write streaming
 * copy in GB/s over `bytes` bytes. Synchronous. */
int32_t dhqr_bench_mfma_f64(dhqr_ctx *ctx, double *tflops);
int32_t dhqr_bench_stream_f64(dhqr_ctx *ctx, int64_t bytes, double *gbps);
/* Issue-rate probe in shader cycles (s_memtime, DVFS independent): kind 0 = v_mfma_f64_16x16x4_f64,
 * kind 1 = v_fma_f64; nblocks workgroups of 4 waves (one per SIMD), 16 independent chains per wave.
 * Returns mean cycles per instruction per wave and the wall-clock TFLOP/s of the launch. */
int32_t dhqr_bench_issue_f64(dhqr_ctx *ctx, int32_t kind, int32_t nblocks, double *cycles_per_instr,
                             double *tflops);

/* ------------------------------------------------------------------ test hook
 * One v_mfma_f64_16x16x4_f64 with A[i][k] = da[i*4+k], B[k][j] = db[k*16+j], C = 0, operands
 * loaded with the lane maps documented in csrc/dhqr_gemm.h; dout[lane*4+g] = raw D register g.
 * tests/test_gpu_kernels.py uses it to pin the f64 C/D fragment layout on the device. Synchronous. */
int32_t dhqr_debug_mfma_probe(dhqr_ctx *ctx, const double *da, const double *db, double *dout);

/* Probe 2: `threads`/256 waves per SIMD; mode 0 all-MFMA, 1 all-v_fma_f64, 2 mixed (waves 0-3 MFMA,
 * rest VALU).  out4 = {cycles/MFMA/wave, cycles/v_fma_f64/wave, MFMA TFLOP/s, VALU TFLOP/s}. */
int32_t dhqr_bench_issue2_f64(dhqr_ctx *ctx, int32_t mode, int32_t threads, int32_t nblocks, double *out4);

/* GEMM micro-benchmark of the two wide trailing-update kernels on synthetic operands: kind 0 = C -= [V_a V_b] W
 * (k_gemm_nn_sub, K = 256), kind 1 = Y = [V_a V_b]' C (k_gemm_tn2); rows, ncols multiples of 128.
 * out4 = {ms per launch, TFLOP/s, shader clock in MHz sustained under the kernel (one-wave s_memtime probe on a
 * second stream), 0}.  Synchronous. */
int32_t dhqr_bench_gemm_f64(dhqr_ctx *ctx, int32_t kind, int64_t rows, int64_t ncols, int32_t reps, double *out4);
/* MFMA cadence probe: the GEMM kernels' inner loop (4 x 4 MFMA tiles per wave, fragments from LDS) alone; mode 0 register
 * operands, 1 k-contiguous LDS layout stride 18 (merged ds_read2_b64), 2 same with plain ds_read_b64, 3 stride 17,
 * 4 the NN kernel's operand layouts; threads = 256 / 512 (one / two waves per SIMD).
 * out2 = {cycles per MFMA per wave, TFLOP/s}.  Synchronous. */
int32_t dhqr_bench_mma_probe_f64(dhqr_ctx *ctx, int32_t mode, int32_t threads, double *out2);
/* Feasibility probe for lane kernels that co-reside with the wide subtraction (csrc/dhqr_bench.h, tools/thin_lane_probe.py):
 * on the context's high-priority stream, `reps` times a Gram product of a rows x 128 panel in `nsplit` workgroups of the
 * subtraction's footprint + its reduction, then a one-workgroup stand-in for the panel kernels (1024 threads, lds_kb KB of
 * LDS, 128 barrier steps).  out4 = {ms Gram + reduction, ms stand-in, ms total, 0} per repetition.  Synchronous. */
int32_t dhqr_bench_lane_probe_f64(dhqr_ctx *ctx, int64_t rows, int32_t nsplit, int32_t lds_kb, int32_t reps, double *out4);

/* Test hook: occupy compute units.  release = 0: launch, on a stream of its own, `nwg` workgroups of 1024 threads with
 * 150 KB of LDS each (one per CU; nothing else fits beside one) that spin until released or until `max_ms` have passed
 * (bounded: the hook can never hang the device), and return at once; release = 1: let them go and wait for them.
 * tests/test_gpu_kernels.py holds all but a few CUs this way while a solve's persistent kernel needs every workgroup
 * resident: its bounded waits expire and the solve must be repeated with the per-step kernels. */
int32_t dhqr_debug_hold_cus(dhqr_ctx *ctx, int32_t nwg, int32_t max_ms, int32_t release);
/* Phase clock of k_small_qr_d (csrc/dhqr_small.h; instrumented in libdhqr_bench.so only): shader cycles summed over the
 * column steps of the launches since the previous call, per wave w < 9: out54[6 w + q], q = 0 everything up to the end of
 * the update, 1 reflector construction (barrier form), 2 wait at the barrier, 3 / 4 flag form: wait for reflector j / the
 * build block, 5 number of steps (tools/smq_phases.py prints them). */
int32_t dhqr_debug_smq_phases(dhqr_ctx *ctx, double *out54);

#ifdef __cplusplus
}
#endif
#endif /* DHQR_BENCH_H */
