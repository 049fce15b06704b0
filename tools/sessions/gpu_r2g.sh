#!/bin/bash
# SQ counters of the two wide GEMM kernels (torch-free driver, ONE counter per pass): where do the MFMA cycles go?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$R/tools/pmc_driver
timeout 60 rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" $O/counters_list.txt | sort -u | tr '\n' ' ' | cut -c1-6000 > $O/counter_names.txt
for kind in 0 1; do
  for ctr in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES; do
    ( timeout 120 rocprofv3 --pmc $ctr --kernel-trace -d $O/k${kind}_$ctr -o out --output-format csv -- $D gemm $kind 32768 32768 2 > $O/k${kind}_$ctr.log 2>&1; echo "rc=$?" >> $O/k${kind}_$ctr.log )
    f=$(find $O/k${kind}_$ctr -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" $kind $ctr <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = "k_gemm_nn_sub" if sys.argv[2] == "0" else "k_gemm_tn2"
vals = [float(r["Counter_Value"]) for r in rows if want in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[3]]
print(f"kind {sys.argv[2]} {sys.argv[3]}: launches {len(vals)} per-launch {sum(vals) / max(1, len(vals)):.4g}")
PY
    else echo "kind $kind $ctr: no counter file ($(tail -1 $O/k${kind}_$ctr.log))"; fi
  done
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" -size +2M -delete
