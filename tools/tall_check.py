"""Developer probe: the unblocked path on columns taller than 8192 rows (k_rankk_tall) against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
from oracle import dhqr_oracle as orc  # noqa: E402  (checker only)

pkg = g.import_package()
shapes = [(12288, 64), (16390, 48), (9000, 40), (16384, 33), (12001, 37), (8200, 24), (20000, 20)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for m, n in shapes:
    A = pkg.rand_colmajor(m, n, 3, "cuda:0")
    A0 = A.clone()
    H = pkg.qr_(A, nb=0)
    torch.cuda.synchronize()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 3))
    scale = np.abs(Ho).max()
    eH = np.abs(H.A.cpu().numpy() - Ho).max() / scale
    ea = np.abs(H.α.cpu().numpy() - ao).max() / scale
    print(f"{m} x {n}: |dH|={eH:.2e} |dalpha|={ea:.2e} resid={pkg.residual(H, A0):.2e} tol={8 * max(m, n) * 2.2e-16:.1e}", flush=True)
