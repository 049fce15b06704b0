#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
$R/tools/pmc_driver 2048 0 || exit 1
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_unblocked_$CNT -o pmc -- $R/tools/pmc_driver 8192 0 > $R/gpurun_out/pmc_unblocked_$CNT.log 2>&1
  DHQR_LOOKAHEAD=0 timeout 900 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_blocked_$CNT -o pmc -- $R/tools/pmc_driver 16384 128 > $R/gpurun_out/pmc_blocked_$CNT.log 2>&1
done
cd $R
tail -3 gpurun_out/pmc_unblocked_FETCH_SIZE.log | cut -c1-200
find gpurun_out/pmc_unblocked_FETCH_SIZE | head
python - <<'PY'
import csv, glob, collections, json
out = {}
for cfg in ("unblocked", "blocked"):
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"gpurun_out/pmc_{cfg}_{cnt}/**/*counter_collection.csv", recursive=True)
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in files:
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "")[:60]
                if row.get("Counter_Name") != cnt: continue
                agg[name][0] += 1; agg[name][1] += float(row["Counter_Value"])
        out[f"{cfg}_{cnt}"] = {k: {"dispatches": v[0], "sum": v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3500])
PY
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
