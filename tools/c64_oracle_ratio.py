"""The reference's ComplexF64 acceptance statistic (test/runtests.jl:49-62) for the ORACLE -- the CPU restatement of the
reference's algorithm -- on the draws tests/test_gpu_complex.py records for the GPU path: is `< 8 stdliberr` a property of
the reference's algorithm on these draws, or only of its Julia RNG stream?  CPU only (O(m n^2) per draw).

  python tools/c64_oracle_ratio.py [--seeds 0 2 4 6 8] [--shape 4400 4000] [--out profiles/r04_c64_oracle_ratio.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dhqr_oracle as orc  # noqa: E402


def ratios(m, n, seed, extra_x=None):
    """(oracle ratio in double as the reference writes it, the same with residuals in extended precision, stdliberr)"""
    A = orc.rand_matrix_c(m, n, seed)
    b = orc.rand_vector_c(m, seed + 1)
    q, r = np.linalg.qr(A)
    x1 = sl.solve_triangular(r, q.conj().T @ b)
    Ah = A.conj().T
    stdliberr = np.linalg.norm(Ah @ (A @ x1) - Ah @ b)
    H, al = orc.householder_c(A.copy(order="F"))
    x2 = orc.solve_c(H, al, b)
    err = np.linalg.norm(Ah @ (A @ x2) - Ah @ b)
    out = {"seed": seed, "stdliberr": float(stdliberr), "oracle_err": float(err), "oracle_ratio": float(err / stdliberr),
           "x_rel_diff_oracle_vs_lapack": float(np.abs(x2 - x1).max() / np.abs(x1).max())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 2, 4, 6, 8])
    ap.add_argument("--shape", type=int, nargs=2, default=[4400, 4000])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    orc.build()
    m, n = a.shape
    res = []
    for s in a.seeds:
        t0 = time.time()
        r = ratios(m, n, s)
        r["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(r), flush=True)
        res.append(r)
    doc = {"what": "test/runtests.jl:49-62 statistic for the ORACLE (reference restatement), ComplexF64, one draw per seed",
           "shape": [m, n], "threads": orc.num_threads(), "draws": res}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
