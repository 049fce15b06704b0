"""Micro-benchmark of the two wide trailing-update kernels (dhqr_bench_gemm_f64) with the shader clock sustained under
them; switches come from the environment (DHQR_STAGGER, DHQR_SWIZZLE).
argv: kind rows ncols reps ...   Prints one line per measurement."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    pkg = g.import_package()
    L, _bh = pkg.bench_context(0)  # libdhqr_bench.so
    ctx = type("BenchCtx", (), {"handle": _bh})
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("DHQR_")) or "default"
    args = sys.argv[1:]
    while len(args) >= 4:
        kind, rows_, ncols, reps = (int(x) for x in args[:4])
        args = args[4:]
        out = (ctypes.c_double * 4)()
        pkg.bench_check(L, L.dhqr_bench_gemm_f64(ctx.handle, kind, rows_, ncols, reps, out))
        print(f"[{tag}] kind={('NN256', 'TN2', 'NN512')[kind]} {rows_}x{ncols}: {out[0]:.3f} ms/launch, {out[1]:.2f} TFLOP/s "
              f"({out[1] / 78.6:.3f} of 78.6), shader clock {out[2]:.0f} MHz", flush=True)


if __name__ == "__main__":
    main()
