import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as g
pkg = g.import_package()
for m, n in ((440, 400), (880, 800), (1100, 1000), (2200, 2000), (4400, 4000)):
    A0d = pkg.rand_colmajor(m, n, 0, "cuda:0")
    A0 = np.asfortranarray(A0d.cpu().numpy())
    for nb in (128, 0):
        th, td = [], []
        for _ in range(8):
            A = A0.copy(order="F")
            t0 = time.perf_counter(); H = pkg.qr_(A, nb=nb); th.append(time.perf_counter() - t0)
            Ad = A0d.clone(); al = torch.zeros(n, dtype=torch.float64, device="cuda:0"); torch.cuda.synchronize()
            t0 = time.perf_counter(); pkg.householder_(Ad, al, nb=nb); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
        print(m, n, "nb", nb, "host us", round(min(th) * 1e6, 1), "device-resident us", round(min(td) * 1e6, 1), flush=True)
