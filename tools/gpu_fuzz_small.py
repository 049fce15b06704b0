"""Randomised shapes through the small route (csrc/dhqr_small.h: one single-workgroup launch per qr! / per \\) on host arrays
and device tensors against the oracle: |dH|, |dalpha| <= 8 n eps max|H|, |dx| <= 1e-9 max|x|, the reference's acceptance
inequality against LAPACK (test/runtests.jl:61-63) counted, not asserted.   usage: gpu_fuzz_small.py [seed] [cases]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
from oracle import dhqr_oracle as orc  # noqa: E402  (the checker)
import scipy.linalg as sl  # noqa: E402

pkg = g.import_package()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fail = 0
worst = 0.0
ratios = []
for case in range(ncases):
    m = rng.choice([rng.randint(1, 256), rng.randint(100, 130), rng.randint(200, 256), 128, 224, 256])
    nmax = 224 if m <= 224 else 192
    n = min(m, rng.choice([rng.randint(1, nmax), nmax, max(1, m - rng.randint(0, 10))]))
    n = min(n, nmax)
    seed = rng.randint(0, 1 << 30)
    A0 = np.asfortranarray(orc.rand_matrix(m, n, seed))
    b = orc.rand_vector(m, seed + 1)
    Ho, ao = orc.householder(A0)
    xo = orc.solve(Ho, ao, b)
    H = pkg.qr_(A0.copy(order="F"))
    x = pkg.ldiv(H, b)
    scale = max(np.abs(Ho).max(), 1e-300)
    tol = 8.0 * max(n, 8) * np.finfo(np.float64).eps
    eH = np.abs(np.asarray(H.A) - Ho).max() / scale
    ea = np.abs(np.asarray(H.α) - ao).max() / scale
    ex = np.abs(x - xo).max() / max(np.abs(xo).max(), 1e-300)
    worst = max(worst, eH, ea)
    ok = eH <= tol and ea <= tol and ex <= 1e-9
    if m > n:
        q, r = sl.qr(A0, mode="economic")
        x1 = sl.solve_triangular(r, q.T @ b)
        ne, ne1 = np.linalg.norm(A0.T @ (A0 @ x - b)), np.linalg.norm(A0.T @ (A0 @ x1 - b))
        if ne1 > 0:
            ratios.append(ne / ne1)
            if ne / ne1 >= 6.0:
                xo_ne = np.linalg.norm(A0.T @ (A0 @ xo - b))
                print(f"note: m={m} n={n} seed={seed}: statistic {ne:.3e} = {ne / ne1:.2f} x LAPACK's {ne1:.3e}; the oracle's x: {xo_ne / ne1:.2f} x", flush=True)
    if not ok:
        fail += 1
        print(f"FAIL m={m} n={n} seed={seed} dH={eH:.2e} da={ea:.2e} dx={ex:.2e}", flush=True)
r = np.array(ratios)
print(f"{ncases} cases, failures: {fail}, worst |dH|/max|H| = {worst:.2e}; acceptance statistic / LAPACK's: median {np.median(r):.2f}, "
      f"max {r.max():.2f}, above 8: {(r >= 8).sum()} of {len(r)}")
sys.exit(1 if fail else 0)
