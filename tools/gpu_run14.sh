#!/bin/bash
mkdir -p gpurun_out
for D in python c; do timeout 900 python bench.py --driver $D --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$D', 'ms/step', round(d['ms_per_step'],1), 'TF/s', round(d['value']/1e3,2), 'resid', d['residual'], 'panels', d['panels_fast_fallback'], {k: round(v,1) for k,v in d['phase_ms_per_step'].items()}, 'roofline', round(d['roofline']['achieved'],1), d['roofline']['kernel'][:16])"; done | tee gpurun_out/bench14.txt
# world size 2 on ONE gpu (gloo): functional timing of the P>1 code path of bench.py itself
cat > /tmp/two.py <<'PY'
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
rank = int(os.environ["RANK"]); torch.cuda.set_device(0)
dist.init_process_group("gloo")
pkg = g.import_package()
n = 8192
q = pkg.ColumnCyclicQR(n, n)
for it in range(2):
    q.fill(0); torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    q.factor(); torch.cuda.synchronize(); dist.barrier(); dt = time.perf_counter() - t0
res = q.residual(0)
if rank == 0: print(f"2 ranks sharing one GPU (gloo), n={n}: {dt*1e3:.1f} ms, residual {res:.2e}")
dist.destroy_process_group()
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 /tmp/two.py 2>&1 | grep -v "amdgpu\|Gloo\|W0\|\*\*\*" | tail -3 | tee -a gpurun_out/bench14.txt
