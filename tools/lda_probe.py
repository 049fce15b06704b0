"""Does a power-of-two leading dimension cost the wide GEMMs anything?  The same 32768^2 (or n^2) blocked factorisation with
the matrix stored at lda = n, n + 16, n + 32, n + 128, n + 2048 (a view into a larger column-major buffer).
  python tools/lda_probe.py [n=32768]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = pkg.get_context(0)
import ctypes
L = pkg._lib.lib()
for pad in (0, 16, 32, 128, 2048, 0):
    lda = n + pad
    buf = torch.empty((n, lda), dtype=torch.float64, device="cuda:0")   # row-major (n, lda) == column-major lda x n
    A = buf.t()[:n, :]                                                  # n x n view, column stride lda
    al = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    ts = []
    for r in range(3):
        ctx.use_torch_stream()
        pkg._lib.check(L.dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), n, n, lda, 0, n, 0, pkg.NB, 1, 0))
        torch.cuda.synchronize()
        ctx.reset_stats()
        ctx.set_profiling(r == 2)
        t0 = time.perf_counter()
        pkg.householder_(A, al, nb=128)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = ctx.stats()
    ctx.set_profiling(False)
    print(json.dumps({"n": n, "lda": lda, "ms": [round(t * 1e3, 1) for t in ts], "tn_ms": round(st["ms_gemm_vta"], 1),
                      "nn_ms": round(st["ms_gemm_avw"], 1), "tn_tflops": round(st["flops_gemm_vta"] / st["ms_gemm_vta"] / 1e9, 2),
                      "nn_tflops": round(st["flops_gemm_avw"] / st["ms_gemm_avw"] / 1e9, 2)}), flush=True)
    del buf, A
    torch.cuda.empty_cache()
