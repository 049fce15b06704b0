"""device-resident blocked factorisations of a SHORT matrix under rocprofv3 --kernel-trace: which kernels fill the time.
usage: short_prof.py m n [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.import_package()
m, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
A0 = pkg.rand_colmajor(m, n, 0, "cuda:0")
for r in range(reps):
    A = A0.clone(); al = torch.zeros(n, dtype=torch.float64, device="cuda:0"); torch.cuda.synchronize()
    t0 = time.perf_counter(); pkg.householder_(A, al, nb=128); torch.cuda.synchronize()
    print(m, n, "us", round((time.perf_counter() - t0) * 1e6, 1), flush=True)
