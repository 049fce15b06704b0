"""Diagnostic: ComplexF64 blocked (nb = 64) vs unblocked (nb = 0) accuracy on one shape (argv: m n [seed])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.linalg as sl  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402
from oracle import dhqr_oracle as orc  # noqa: E402  (diagnostic tool, not the product path)


def main():
    m, n = int(sys.argv[1]), int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    pkg = g.import_package()
    A = orc.rand_matrix_c(m, n, seed)
    b = orc.rand_vector_c(m, seed + 1)
    Ah = A.conj().T
    q, r = np.linalg.qr(A)
    x1 = sl.solve_triangular(r, q.conj().T @ b)
    std = np.linalg.norm(Ah @ (A @ x1) - Ah @ b)
    xr = np.linalg.lstsq(A, b, rcond=None)[0]
    sv = np.linalg.svd(A, compute_uv=False)
    print(f"m={m} n={n} seed={seed} cond={sv[0] / sv[-1]:.3e} stdliberr={std:.3e}")
    out = {}
    for nb in (0, 64):
        H = pkg.qr_(A.copy(order="F"), nb=nb)
        x = pkg.ldiv(H, b)
        ne = np.linalg.norm(Ah @ (A @ x) - Ah @ b)
        QR = orc.form_qr_c(np.asfortranarray(H.A), H.α)
        v2 = (np.abs(np.tril(H.A)) ** 2).sum(axis=0)
        print(f"  nb={nb}: normal-eq residual {ne:.3e} = {ne / std:.2f} x stdlib; |x-lstsq|/|x| {np.abs(x - xr).max() / np.abs(xr).max():.2e};"
              f" ||A-QR||/||A|| {np.linalg.norm(A - QR) / np.linalg.norm(A):.2e}; max|v2-2| {np.abs(v2 - 2).max():.2e}")
        out[nb] = (H.A.copy(), H.α.copy())
    scale = np.abs(out[0][0]).max()
    d = np.abs(out[64][0] - out[0][0])
    print(f"  blocked vs unblocked: max|dH|/scale {d.max() / scale:.2e} at col {np.unravel_index(d.argmax(), d.shape)[1]};"
          f" per-64-col-block max: {[float(f'{d[:, k:k + 64].max() / scale:.1e}') for k in range(0, n, max(64, n // 16 // 64 * 64))]}")
    print(f"  |dalpha|/scale {np.abs(out[64][1] - out[0][1]).max() / scale:.2e}")


if __name__ == "__main__":
    main()
