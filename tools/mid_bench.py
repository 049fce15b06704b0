"""qr!(A) on host arrays at mid sizes, blocked (nb = 128) against unblocked (nb = 0): where the default should switch.
usage: mid_bench.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.import_package()
for m, n in ((300, 260), (440, 400), (520, 512), (600, 300), (1024, 512), (2048, 512), (4096, 400), (8192, 512), (16384, 300), (660, 600), (1500, 640), (4096, 640)):
    A0 = np.asfortranarray(pkg.rand_colmajor(m, n, 0, "cuda:0").cpu().numpy())
    r = {}
    for nb in (128, 0):
        ts = []
        for _ in range(10):
            A = A0.copy(order="F")
            t0 = time.perf_counter(); H = pkg.qr_(A, nb=nb); ts.append(time.perf_counter() - t0)
        r[nb] = min(ts) * 1e6
    print(m, n, "blocked us", round(r[128], 1), "unblocked us", round(r[0], 1), "ratio", round(r[0] / r[128], 3), flush=True)
