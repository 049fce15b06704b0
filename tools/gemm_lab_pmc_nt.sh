#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab/pmc_nt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in glds512_mid glds512_mid_nt glds256_mid glds256_mid_nt; do
  LAB_WARM=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${v}_FETCH_SIZE -o out --output-format csv -- $R/tools/gemm_lab $v 32768 28672 1 > $O/$v.log 2>&1
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
python - <<'PY'
import csv,glob,os,collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/lab/pmc_nt'
for d in sorted(glob.glob(O+'/*_FETCH_SIZE')):
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: print(d,'no csv'); continue
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0][:60]
        a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])*2*1024
    print('==',os.path.basename(d))
    for k,a in agg.items():
        if a[1]/a[0]>1e9: print(f'  {k:62s} launches {a[0]:3d} read/launch {a[1]/a[0]/1e9:8.3f} GB')
PY
