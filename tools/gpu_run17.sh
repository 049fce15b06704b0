#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "row_split" 2>&1 | tail -12
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/rowsplit_timing.txt
import time, torch, __graft_entry__ as g
pkg = g.import_package()
m, n = 262144, 4096
q = pkg.RowSplitQR(m, n)
for it in range(2):
    q.fill(0); torch.cuda.synchronize(); t0 = time.perf_counter(); q.factor(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fl = 2.0*m*n*n - 2.0/3.0*n**3
print(f"row-split driver, world size 1, {m}x{n}: {dt*1e3:.1f} ms, {fl/dt/1e12:.2f} TFLOP/s, retries {q.stats}")
print("residual", q.residual(0))
A = pkg.rand_colmajor(m, n, 0, "cuda:0"); al = torch.zeros(n, dtype=torch.float64, device="cuda:0")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pkg.householder_(A, al, nb=128); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
print(f"column driver (dhqr_factor_f64) same shape (2nd call refactors the factor): {dt2*1e3:.1f} ms, {fl/dt2/1e12:.2f} TFLOP/s")
PY
