"""Unblocked (nb = 0) factorisations with columns of more than 16384 rows: ms and TFLOP/s with k_rankk_xtall
(DHQR_RANKK=5, default) and with one reflector per launch up there (=1).  usage: python tools/xtall_bench.py [m,n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
shapes = [(32768, 4096), (24576, 24576), (32768, 32768), (20000, 2048), (16384, 16384)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for m, n in shapes:
    for K in (os.environ.get("XTALL_KS", "1,5").split(",")):
        os.environ["DHQR_RANKK"] = K
        pkg.api._contexts.pop(0, None)  # a fresh context reads the switch
        A0 = pkg.rand_colmajor(m, n, 3, "cuda:0")
        best = 1e30
        for rep in range(2):
            A = A0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            H = pkg.qr_(A, nb=0)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        fl = 2.0 * m * n * n - 2.0 / 3.0 * n ** 3
        r = pkg.residual(H, A0)
        print(f"{m} x {n} nb=0 DHQR_RANKK={K}: {best * 1e3:9.1f} ms {fl / best / 1e12:6.2f} TFLOP/s resid {r:.2e}", flush=True)
        del A, H, A0
        torch.cuda.empty_cache()
