#!/bin/bash
# A/B of DHQR_TUNE switches on the blocked sizes: ab_sizes.sh "<tune A>" "<tune B>" ... ; prints ms per factorisation (best of 3 warm runs)
cd $GRAFT_REPO_ROOT
for T in "$@"; do
  echo "== DHQR_TUNE=$T"
  DHQR_TUNE="$T" timeout 600 python - <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, __graft_entry__ as g
pkg = g.import_package()
ctx = pkg.get_context(0)
for n in (4096, 8192, 16384, 32768):
    A = pkg.empty_colmajor(n, n, "cuda:0"); al = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    import ctypes
    L = pkg._lib.lib()
    ts = []
    for r in range(4):
        ctx.use_torch_stream()
        pkg._lib.check(L.dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), n, n, n, 0, n, 0, pkg.NB, 1, 0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pkg.householder_(A, al, nb=pkg.NB)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res = pkg.residual(pkg.DistributedHouseholderQRStruct(A, al), pkg.rand_colmajor(n, n, 0, "cuda:0")) if n <= 16384 else -1
    print(n, "ms", round(min(ts[1:]), 2), "all", [round(t, 2) for t in ts], "resid", res, flush=True)
    del A, al; torch.cuda.empty_cache()
PY
done
