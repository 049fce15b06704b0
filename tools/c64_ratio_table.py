"""ComplexF64 acceptance statistic of test/runtests.jl:49-62 for LAPACK, the ORACLE (the reference restated on the CPU) and the GPU
paths (nb = 0: the reference's operation order, nb = 64: blocked), per draw, evaluated (a) in double exactly as the reference
writes it and (b) with the residual r = A x - b in extended precision (the evaluation's own rounding removed).  GPU box only.

  python tools/c64_ratio_table.py [--seeds 0 2 4 6 8] [--shape 4400 4000] [--out profiles/r04_c64_ratio_table.json]
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
import __graft_entry__ as g  # noqa: E402
from oracle import dhqr_oracle as orc  # noqa: E402


def stat_double(A, x, b):
    Ah = A.conj().T
    return float(np.linalg.norm(Ah @ (A @ x) - Ah @ b))


def stat_ext(A, x, b, Ar=None, Ai=None):
    Ar = A.real.astype(np.longdouble) if Ar is None else Ar
    Ai = A.imag.astype(np.longdouble) if Ai is None else Ai
    xr, xi = x.real.astype(np.longdouble), x.imag.astype(np.longdouble)
    rr = Ar @ xr - Ai @ xi - b.real
    ri = Ar @ xi + Ai @ xr - b.imag
    return float(np.linalg.norm(A.conj().T @ (rr.astype(float) + 1j * ri.astype(float))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 2, 4, 6, 8])
    ap.add_argument("--shape", type=int, nargs=2, default=[4400, 4000])
    ap.add_argument("--out", default=None)
    ap.add_argument("--real", action="store_true", help="Float64 instead of ComplexF64 (nb = 0 / 128)")
    a = ap.parse_args()
    pkg = g.import_package()
    orc.build()
    m, n = a.shape
    rows = []
    for s in a.seeds:
        t0 = time.time()
        A = orc.rand_matrix(m, n, s).astype(complex) if a.real else orc.rand_matrix_c(m, n, s)
        b = orc.rand_vector(m, s + 1).astype(complex) if a.real else orc.rand_vector_c(m, s + 1)
        Ar, Ai = A.real.astype(np.longdouble), A.imag.astype(np.longdouble)
        q, r = np.linalg.qr(A)
        xs = {"lapack": sl.solve_triangular(r, q.conj().T @ b)}
        if a.real:
            Ar_ = np.asfortranarray(A.real)
            Ho, ao = orc.householder(Ar_.copy(order="F"))
            xs["oracle"] = orc.solve(Ho, ao, b.real.copy()).astype(complex)
        else:
            Ho, ao = orc.householder_c(A.copy(order="F"))
            xs["oracle"] = orc.solve_c(Ho, ao, b)
        for nb in ((0, 128) if a.real else (0, 64)):
            if a.real:
                H = pkg.qr_(Ar_.copy(order="F"), nb=nb)
                xs[f"gpu_nb{nb}"] = np.asarray(pkg.ldiv(H, b.real.copy())).astype(complex)
                xs[f"gpu_nb{nb}_factor_oracle_solve"] = orc.solve(np.asfortranarray(np.asarray(H.A)), np.asarray(H.α), b.real.copy()).astype(complex)
                continue
            H = pkg.qr_(A.copy(order="F"), nb=nb)
            xs[f"gpu_nb{nb}"] = np.asarray(pkg.ldiv(H, b))
            # the oracle's solve on the GPU's factor: separates the factorisation's share from the solve's
            xs[f"gpu_nb{nb}_factor_oracle_solve"] = orc.solve_c(np.asfortranarray(np.asarray(H.A)), np.asarray(H.α), b)
        xt = xs["lapack"]
        row = {"seed": s}
        for k, x in xs.items():
            row[k] = {"double": stat_double(A, x, b), "ext": stat_ext(A, x, b, Ar, Ai),
                      "dx_vs_lapack": float(np.abs(x - xt).max() / np.abs(xt).max())}
        row["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(row), flush=True)
        rows.append(row)
    if a.out:
        json.dump({"what": __doc__, "shape": [m, n], "host_threads": orc.num_threads(), "draws": rows}, open(a.out, "w"), indent=1)
    print("\nratio to LAPACK (double | ext):")
    for row in rows:
        l = row["lapack"]
        print(f"seed {row['seed']}: " + "  ".join(f"{k} {row[k]['double'] / l['double']:.2f}|{row[k]['ext'] / l['ext']:.2f}"
                                                 for k in row if isinstance(row[k], dict) and k != "lapack"))


if __name__ == "__main__":
    main()
