"""Generate the ComplexF64 golden fixtures (tests/golden/zqr_*.npz) from the CPU oracle.

Same rules as make_golden.py: the reference cannot run here and stores no vectors, so the
fixtures come from oracle/dhqr_oracle_c64.c and every case is cross-checked against LAPACK zgeqrf
(scipy) before it is written: rows of R equal up to the unit phase of alpha_j (the reference does
not phase-normalise R: diag(R) = alpha_j = -exp(i arg a_jj) s), ||A - QR||/||A|| at round-off.
Inputs are regenerated from the portable generator (rand_matrix_c / rand_vector_c).

Run from the repo root:  python tests/golden/make_golden_c64.py
"""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dhqr_oracle as orc  # noqa: E402

CASES = [(8, 5, 0), (33, 17, 1), (64, 64, 2), (110, 100, 0), (300, 128, 4)]
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    orc.build()
    for m, n, seed in CASES:
        A = orc.rand_matrix_c(m, n, seed)
        b = orc.rand_vector_c(m, seed + 1)
        H, alpha = orc.householder_c(A)
        x = orc.solve_c(H, alpha, b)
        (qr_raw, _tau), _ = sl.qr(A, mode="raw")
        R = np.triu(H, 1)[:n] + np.diag(alpha)
        Rl = np.triu(qr_raw)[:n]
        k = n - 1 if m == n else n
        ph = np.diag(R)[:k] / np.diag(Rl)[:k]
        assert np.abs(np.abs(ph) - 1).max() < 1e-13 * n
        assert np.abs(R[:k] - ph[:, None] * Rl[:k]).max() < 1e-13 * n * np.abs(R).max()
        QR = orc.form_qr_c(H, alpha)
        assert np.linalg.norm(A - QR) / np.linalg.norm(A) < 1e-14
        np.savez(os.path.join(HERE, f"zqr_{m}x{n}_seed{seed}.npz"), m=m, n=n, seed=seed, H=H, alpha=alpha, x=x)
        print(f"wrote zqr_{m}x{n}_seed{seed}.npz")


if __name__ == "__main__":
    main()
