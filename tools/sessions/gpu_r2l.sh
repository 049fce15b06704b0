#!/bin/bash
# power / clock telemetry while (a) the pure MFMA loop, (b) the NN GEMM, (c) the TN2 GEMM, (d) the factorisation run
mkdir -p gpurun_out
O=gpurun_out/r2l_power.txt
: > $O
sample() {  # tag, seconds
  for i in $(seq 1 $2); do
    echo "$1 $(rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E 'Power|sclk|mclk|GPU use' | tr -s ' ' | tr '\n' ';')" >> $O
    sleep 0.5
  done
}
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30 >> $O
python - <<'PY' &
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.import_package(); L, _bh = pkg.bench_context(0); ctx = type("BenchCtx", (), {"handle": _bh})
out = (ctypes.c_double * 4)()
def mark(s):
    open("gpurun_out/r2l_marks.txt", "a").write(f"{time.time():.2f} {s}\n")
time.sleep(1.0)
mark("pure_mfma_2waves begin")
t0 = time.time()
while time.time() - t0 < 4: pkg.bench_check(L, L.dhqr_bench_mma_probe_f64(ctx.handle, 1, 512, out))
mark(f"pure_mfma_2waves end {out[0]:.1f} cyc {out[1]:.1f} TF")
time.sleep(1.0)
mark("nn begin")
pkg.bench_check(L, L.dhqr_bench_gemm_f64(ctx.handle, 0, 32768, 32768, 400, out))
mark(f"nn end {out[1]:.2f} TF {out[2]:.0f} MHz")
time.sleep(1.0)
mark("tn2 begin")
pkg.bench_check(L, L.dhqr_bench_gemm_f64(ctx.handle, 1, 32768, 32768, 400, out))
mark(f"tn2 end {out[1]:.2f} TF {out[2]:.0f} MHz")
time.sleep(1.0)
import torch
A = pkg.empty_colmajor(32768, 32768, torch.device("cuda", 0)); al = torch.zeros(32768, dtype=torch.float64, device="cuda:0")
mark("factor begin")
for _ in range(4):
    pkg.bench_check(L, L.dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), 32768, 32768, 32768, 0, 32768, 0, 128, 1, 0))
    pkg.householder_(A, al, nb=128)
torch.cuda.synchronize()
mark("factor end")
PY
PID=$!
while kill -0 $PID 2>/dev/null; do
  echo "$(date +%s.%N | cut -c1-13) $(rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E 'Power|sclk|GPU use' | tr -s ' ' | tr '\n' ';')" >> $O
done
cat gpurun_out/r2l_marks.txt
