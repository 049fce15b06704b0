"""MFMA cadence probe (dhqr_bench_mma_probe_f64): cycles per v_mfma_f64_16x16x4_f64 per wave in the GEMM inner loop."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

NAMES = {0: "register operands", 1: "LDS stride 18, ds_read2_b64 (as shipped)", 2: "LDS stride 18, plain ds_read_b64",
         3: "LDS stride 17, ds_read2_b64", 4: "NN operands (V rows stride 144 + W stride 18)"}


def main():
    pkg = g.import_package()
    L, _bh = pkg.bench_context(0)  # libdhqr_bench.so
    ctx = type("BenchCtx", (), {"handle": _bh})
    for threads in (256, 512):
        for mode in range(5):
            out = (ctypes.c_double * 2)()
            pkg.bench_check(L, L.dhqr_bench_mma_probe_f64(ctx.handle, mode, threads, out))
            print(f"waves/SIMD={threads // 256} mode {mode} ({NAMES[mode]}): {out[0]:.1f} cycles/MFMA/wave, {out[1]:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
