"""blocked ComplexF64 factorisation on one GPU: best of 5, ms (env knobs are read when the context is created)"""
import sys, time, torch
sys.path.insert(0, ".")
import __graft_entry__
pkg = __graft_entry__.import_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A0 = pkg.rand_colmajor_c(n, n, 3, "cuda:0")
A = A0.clone()
ts = []
for i in range(6):
    A.copy_(A0); torch.cuda.synchronize(); t = time.perf_counter()
    pkg.qr_(A, nb=64); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(f"{n}^2 ComplexF64 blocked: {min(ts[1:]):.1f} ms", flush=True)
