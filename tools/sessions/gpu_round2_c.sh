#!/bin/bash
# fused two-panel TN kernel: bench + uncontended phases + parity subset
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err
DHQR_LOOKAHEAD=0 timeout 200 python tools/quick_bench.py 32768,128 > gpurun_out/r2c_phases.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu -x -k "parity" > gpurun_out/r2c_pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r2c_pytest.txt; tail -3 gpurun_out/r2c_pytest.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2c_bench.json").read().strip().splitlines()[-1])
print(f"bench: {d['value']:.0f} GFLOP/s, {d['ms_per_step']:.1f} ms/step, resid {d['residual']}, phases {d['phase_ms_per_step']}")
for r in d["roofline_all"]: print(r["kernel"][:30], round(r["achieved"],1), round(r["frac"],3), r["launches"], round(r["avg_launch_ms"],3))
print(open("gpurun_out/r2c_phases.txt").read().strip().splitlines()[-1][:500])
PY
