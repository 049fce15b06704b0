"""column-split solve over R rank threads sharing one GPU (in-process transport): ms per `H \\ b` on a resident factor.
usage: mg_solve_bench.py [ranks ...]   (DHQR_SOLVE_PIPE=0: the per-panel form of rounds 1-5)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
ranks = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
for R in ranks:
    for n in (4096, 8192, 16384):
        mg = pkg.MultiGpuQR(devices=[0] * R)
        try:
            mg.alloc(n, n).fill(3)
            mg.factor()
            b = np.random.default_rng(1).random(n)
            x = mg.solve(b)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                x = mg.solve(b)
                ts.append((time.perf_counter() - t0) * 1e3)
            A0, _ = (None, None)
            print(json.dumps({"ranks": R, "n": n, "solve_ms": round(min(ts), 3), "pipe": os.environ.get("DHQR_SOLVE_PIPE", "1"),
                              "x_norm": float(np.linalg.norm(x))}), flush=True)
        finally:
            mg.close()
