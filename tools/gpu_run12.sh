#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "robustness or fast_panel" 2>&1 | grep -v amdgpu | tail -12 | tee gpurun_out/robustness.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/solve_timing.txt
import time, torch, __graft_entry__ as g
pkg = g.import_package()
for n in (8192, 16384):
    A = pkg.rand_colmajor(n, n, 0, "cuda:0"); H = pkg.qr_(A, nb=128)
    b = pkg.rand_vector_device(n, 1, "cuda:0"); x = pkg.ldiv(H, b); torch.cuda.synchronize()
    t0 = time.perf_counter(); x = pkg.ldiv(H, b); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    A0 = pkg.rand_colmajor(n, n, 0, "cuda:0")
    r = (A0 @ x - b).norm().item() / b.norm().item()
    print(f"solve n={n}: {dt*1e3:.1f} ms, ||Ax-b||/||b|| = {r:.2e}")
PY
