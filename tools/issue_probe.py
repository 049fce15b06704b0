import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.import_package()
L, _bh = pkg.bench_context(0)  # libdhqr_bench.so
ctx = type("BenchCtx", (), {"handle": _bh})
for kind, name in ((0, "v_mfma_f64_16x16x4_f64"), (1, "v_fma_f64")):
    for nb in (1, 8, 256, 1024, 2048):
        c, t = ctypes.c_double(), ctypes.c_double()
        pkg.bench_check(L, L.dhqr_bench_issue_f64(ctx.handle, kind, nb, ctypes.byref(c), ctypes.byref(t)))
        print(f"{name}: blocks={nb:5d} cycles/instr/wave={c.value:8.2f} wall TFLOP/s={t.value:8.2f}", flush=True)

print("--- probe2: waves/SIMD and MFMA+VALU co-issue (256 CUs x 1 block)")
for mode, name in ((0, "all MFMA"), (1, "all VALU"), (2, "mixed")):
    for threads in (256, 512, 1024):
        if mode == 2 and threads == 256:
            continue
        for nb in (256,):
            o = (ctypes.c_double * 4)()
            pkg.bench_check(L, L.dhqr_bench_issue2_f64(ctx.handle, mode, threads, nb, o))
            print(f"{name:9s} waves/SIMD={threads//256} cyc/MFMA/wave={o[0]:7.1f} cyc/FMA/wave={o[1]:6.2f} MFMA TF/s={o[2]:6.1f} VALU TF/s={o[3]:6.1f} sum={o[2]+o[3]:6.1f}", flush=True)
