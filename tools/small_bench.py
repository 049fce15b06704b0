"""qr!(A) \\ b on HOST arrays at the reference's test shapes (test/runtests.jl:42,87-89) beside LAPACK on the host cores:
seconds per call (best of N), the ratio the reference prints, and the two calls separately.
usage: small_bench.py [reps]   (DHQR_SMALL=0: the general drivers)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import scipy.linalg as sl  # noqa: E402

for m, n in ((110, 100), (220, 200), (440, 400), (880, 800), (1100, 1000)):
    A0 = np.asfortranarray(pkg.rand_colmajor(m, n, 0, "cuda:0").cpu().numpy())
    b0 = pkg.rand_vector_device(m, 1, "cuda:0").cpu().numpy()
    tq, tl, tt = [], [], []
    for _ in range(reps):
        A = A0.copy(order="F")
        t0 = time.perf_counter()
        H = pkg.qr_(A)
        t1 = time.perf_counter()
        x = pkg.ldiv(H, b0)
        t2 = time.perf_counter()
        tq.append(t1 - t0), tl.append(t2 - t1), tt.append(t2 - t0)
    tp = []
    from scipy.linalg import lapack
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=16):  # (oracle/lapack_ls_bench.py: the same three LAPACK calls, the same thread count)
        for _ in range(reps):
            A = A0.copy(order="F")
            bl = b0.copy()
            t0 = time.perf_counter()
            qr_, tau, _, _ = lapack.dgeqrf(A, overwrite_a=True)
            cq, _, _ = lapack.dormqr("L", "T", qr_, tau, bl.reshape(m, 1), lwork=64 * n)
            xl, _ = lapack.dtrtrs(qr_[:n, :n], cq[:n], lower=0)
            tp.append(time.perf_counter() - t0)
    xl = xl[:, 0]
    ne = float(np.linalg.norm(A0.T @ (A0 @ x - b0)))
    nel = float(np.linalg.norm(A0.T @ (A0 @ xl - b0)))
    print(json.dumps({"m": m, "n": n, "us_total": round(min(tt) * 1e6, 1), "us_qr": round(min(tq) * 1e6, 1),
                      "us_ldiv": round(min(tl) * 1e6, 1), "us_lapack": round(min(tp) * 1e6, 1),
                      "times_longer_than_lapack": round(min(tt) / min(tp), 3), "normal_eq": ne, "lapack_normal_eq": nel}))
