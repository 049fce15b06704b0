"""device-resident blocked factorisations whose LAST panel is partial (n % 128 != 0) and taller than 256 rows. usage: partial_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.import_package()
for m, n in ((2200, 2000), (4400, 4000), (5000, 1000), (8192, 1000), (3000, 300), (1100, 1000)):
    A0 = pkg.rand_colmajor(m, n, 0, "cuda:0")
    ts = []
    for r in range(6):
        A = A0.clone(); al = torch.zeros(n, dtype=torch.float64, device="cuda:0"); torch.cuda.synchronize()
        t0 = time.perf_counter(); pkg.householder_(A, al, nb=128); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res = pkg.residual(pkg.DistributedHouseholderQRStruct(A, al), A0)
    print(m, n, "us", round(min(ts) * 1e6, 1), "resid", res, flush=True)
