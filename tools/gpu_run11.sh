#!/bin/bash
mkdir -p gpurun_out
for CFG in "DHQR_SWIZZLE=0" "DHQR_SWIZZLE=1" "DHQR_RESERVE_CUS=8" "DHQR_RESERVE_CUS=16" "DHQR_RESERVE_CUS=32 DHQR_SWIZZLE=1"; do echo "== $CFG"; env $CFG timeout 600 python tools/quick_bench.py 32768,128 2>&1 | grep -v "amdgpu.ids\|ubench"; done > gpurun_out/quick_bench11.txt
DHQR_SWIZZLE=1 timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "blocked_vs_oracle or full_size or golden" 2>&1 | tail -3
python - <<'PY'
import json
for l in open('gpurun_out/quick_bench11.txt'):
    if l.startswith('=='): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    s=d['stats']; print(f"  n={d['n']} t0={d['t0']:.3f}s t1={d['t1']:.3f} {d['gflops']/1e3:.1f} TF/s panel={s.get('ms_panel',0):.0f} vta={s.get('ms_gemm_vta',0):.0f} tw={s.get('ms_gemm_tw',0):.0f} avw={s.get('ms_gemm_avw',0):.0f} resid={d.get('resid'):.1e}")
PY
