#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "blocked_vs_oracle or golden or robustness or fast_panel or full_size" 2>&1 | tail -15
for CFG in "DHQR_PAIR=1" "DHQR_PAIR=0"; do echo "== $CFG"; env $CFG timeout 600 python tools/quick_bench.py 16384,128 32768,128 2>&1 | grep -v "amdgpu.ids\|ubench"; done > gpurun_out/quick_bench15.txt
python - <<'PY'
import json
for l in open('gpurun_out/quick_bench15.txt'):
    if l.startswith('=='): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    s=d['stats']; print(f"  n={d['n']} t0={d['t0']:.3f}s t1={d['t1']:.3f}s panel={s.get('ms_panel',0):.0f} vta={s.get('ms_gemm_vta',0):.0f} tw={s.get('ms_gemm_tw',0):.0f} avw={s.get('ms_gemm_avw',0):.0f} resid={d.get('resid'):.1e}")
PY
