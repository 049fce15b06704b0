"""Generate the golden fixtures in tests/golden/ from the CPU oracle.

The reference is pure Julia and cannot run in this image, and its own tests hold no stored
vectors, so these fixtures are produced by oracle/dhqr_oracle.c (the line-by-line restatement of
src/DistributedHouseholderQR.jl) and every case is cross-checked here against LAPACK dgeqrf
(scipy) before it is written:  R equal, tau == v_jj^2, v_lapack == v / v_jj.
Inputs are not stored: A = rand_matrix(m, n, seed), b = rand_vector(m, seed+1) from the portable
generator, whose first values are pinned in generator.npz.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dhqr_oracle as orc  # noqa: E402

CASES = [(8, 5, 0), (33, 17, 1), (64, 64, 2), (110, 100, 0), (129, 129, 3), (300, 128, 4), (220, 200, 0)]
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    orc.build()
    np.savez(os.path.join(HERE, "generator.npz"),
             seed0_first8=orc.u01(0, np.arange(8, dtype=np.uint64)),
             seed12345_idx=np.array([0, 1, 2**32 + 7, 2**40 + 11], dtype=np.uint64),
             seed12345_val=orc.u01(12345, np.array([0, 1, 2**32 + 7, 2**40 + 11], dtype=np.uint64)))
    for m, n, seed in CASES:
        A = orc.rand_matrix(m, n, seed)
        b = orc.rand_vector(m, seed + 1)
        H, alpha = orc.householder(A)
        x = orc.solve(H, alpha, b)
        (qr_raw, tau), _ = sl.qr(A, mode="raw")
        R = np.triu(H, 1)[:n] + np.diag(alpha)
        dR = np.abs(R - np.triu(qr_raw)[:n])
        k = n - 1 if m == n else n  # LAPACK sets tau_n = 0 when m == n; the reference still reflects
        if m == n:
            dR[n - 1, n - 1] = abs(abs(R[n - 1, n - 1]) - abs(qr_raw[n - 1, n - 1]))
        vjj = np.diag(H)[:n]
        assert dR.max() < 1e-13 * np.abs(R).max() * n, (m, n, dR.max())
        assert np.abs(tau[:k] - vjj[:k] ** 2).max() < 1e-13 * n
        Vl = np.tril(qr_raw, -1)[:, :k]
        Vr = (np.tril(H, -1) / vjj)[:, :k]
        assert np.abs(Vl - Vr).max() < 1e-12 * n
        np.savez(os.path.join(HERE, f"qr_{m}x{n}_seed{seed}.npz"), m=m, n=n, seed=seed, H=H, alpha=alpha, x=x)
        print(f"wrote qr_{m}x{n}_seed{seed}.npz  max|dR_lapack|={dR.max():.2e}")


if __name__ == "__main__":
    main()
