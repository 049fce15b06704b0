"""Where does the normal-equation statistic of test/runtests.jl:49-62 (ComplexF64, 4400 x 4000) come from?  Factor with the GPU
(nb = 0, nb = 64) and with the oracle, solve each factorisation with the GPU solve and with the oracle's solve: 6 ratios."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sl
import __graft_entry__ as g
pkg = g.import_package()
from oracle import dhqr_oracle as orc
orc.build()
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4400, 4000)
for seed in (0, 2):
    A = orc.rand_matrix_c(m, n, seed); b = orc.rand_vector_c(m, seed + 1)
    q, r = np.linalg.qr(A); x1 = sl.solve_triangular(r, q.conj().T @ b)
    Ah = A.conj().T
    std = np.linalg.norm(Ah @ (A @ x1) - Ah @ b)
    ratio = lambda x: np.linalg.norm(Ah @ (A @ x) - Ah @ b) / std
    t = time.time(); Ho, ao = orc.householder_c(A.copy(order="F")); t = time.time() - t
    facs = {"oracle": (Ho, ao)}
    for nb in (0, 64):
        F = A.copy(order="F")
        H = pkg.qr_(F, nb=nb)
        facs[f"gpu nb={nb}"] = (F, np.asarray(H.α).copy())
    print(f"seed {seed}: oracle factorisation {t:.0f} s")
    for name, (F, al) in facs.items():
        H = pkg.DistributedHouseholderQRStruct(F, al)
        xg = np.asarray(pkg.ldiv(H, b))
        xo = orc.solve_c(np.asfortranarray(F), al, b)
        dH = np.abs(F - Ho).max() / np.abs(Ho).max()
        print(f"  factor {name:10s} |dH vs oracle| {dH:.1e}   ratio with GPU solve {ratio(xg):6.2f}   with oracle solve {ratio(xo):6.2f}"
              f"   |x_gpu - x_oracle_solve| {np.abs(xg - xo).max() / np.abs(xo).max():.1e}")
