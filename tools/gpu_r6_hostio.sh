#!/bin/bash
# round 6: host-in / host-out -- which engine carries the copies of the staged form, and does an SDMA setting change it?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6io; mkdir -p $O; cd $R
for V in "" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_SDMA=1" "GPU_MAX_HW_QUEUES=8"; do
  echo "== env: $V" >> $O/hostio_env.txt
  ( env $V timeout 400 python tools/hostio_bench.py 32768 2 2>&1 | grep '^{' ) >> $O/hostio_env.txt
done
cd /tmp && export TMPDIR=/tmp
DHQR_HOSTIO=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_io -o io -- python $R/tools/hostio_bench.py --inner 16384 1 > $O/prof_io.log 2>&1
cd $R
python - <<'PY' > $O/copy_engines.txt 2>&1
import sqlite3, glob, os
db = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6io/prof_io/**/*.db", recursive=True)
print("db", db)
con = sqlite3.connect(db[0])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if "memory_cop" in t.lower()]
print("memory copy tables:", mc)
for t in mc[:3]:
    cols = [r[1] for r in con.execute(f"pragma table_info('{t}')")]
    print(t, cols)
    for row in con.execute(f"select * from '{t}' limit 5"): print("  ", row)
    n = con.execute(f"select count(*) from '{t}'").fetchone()[0]
    print("  rows", n)
ks = [t for t in tabs if "kernel" in t.lower() and "dispatch" in t.lower()]
for t in tabs:
    if t.lower().startswith("kernels") or "top_kernels" in t.lower():
        try:
            for row in con.execute(f"select * from '{t}' where name like '%copyBuffer%' or name like '%rocclr%' limit 10"): print(t, row)
        except Exception as e: pass
PY
find $O -name "*.db" -size +40M -delete
cat $O/hostio_env.txt; tail -30 $O/copy_engines.txt
