"""row-split solve over R rank threads sharing one GPU (in-process transport): ms per `H \\ b` on a resident factor.
usage: mg_rs_solve_bench.py [ranks ...]   (DHQR_SOLVE_PIPE=0: the per-panel form of rounds 1-5)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
ranks = [int(a) for a in sys.argv[1:]] or [2, 8]
for R in ranks:
    for m, n in ((65536, 2048), (262144, 4096)):
        mg = pkg.MultiGpuQR(devices=[0] * R)
        try:
            mg.rs_alloc(m, n).rs_fill(3)
            mg.rs_factor()
            b = np.random.default_rng(1).random(m)
            x = mg.rs_solve(b)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                x = mg.rs_solve(b)
                ts.append((time.perf_counter() - t0) * 1e3)
            print(json.dumps({"ranks": R, "m": m, "n": n, "solve_ms": round(min(ts), 3), "pipe": os.environ.get("DHQR_SOLVE_PIPE", "1"),
                              "x_norm": float(np.linalg.norm(x))}), flush=True)
        finally:
            mg.close()
