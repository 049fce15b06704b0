cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zprof -o z -- python tools/zsplit_trace.py 2 > gpurun_out/zprof_run.txt 2>&1
find gpurun_out/zprof -name "*kernel_stats*" | head; f=$(find gpurun_out/zprof -name "*kernel_stats.csv" | head -1); head -30 "$f"; cp "$f" gpurun_out/z_kernel_stats.csv; rm -rf gpurun_out/zprof
tail -3 gpurun_out/zprof_run.txt
