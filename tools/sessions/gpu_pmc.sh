#!/bin/bash
# HBM byte counters: bounded attempts, one counter per pass, every tool under its own short timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=$R/tools/pmc_driver
timeout 30 $D stream 256 > $O/plain_stream.txt 2>&1; echo "plain rc=$?" >> $O/plain_stream.txt
timeout 60 rocprofv3-avail list > $O/avail.txt 2>&1; echo "rc=$?" >> $O/avail.txt
try() {  # tag, cmd...
  local tag=$1; shift
  ( timeout 90 "$@" > $O/$tag.log 2>&1; echo "rc=$?" >> $O/$tag.log ) 
  tail -2 $O/$tag.log | cut -c1-200
}
for ctr in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum; do
  try v3_stream_$ctr rocprofv3 --pmc $ctr --kernel-trace -d $O/v3_stream_$ctr -o out --output-format csv -- $D stream 256
done
# rocprofv2 / rocprof v1 fallbacks
echo "pmc: FETCH_SIZE WRITE_SIZE" > $O/in.txt
try v2_stream rocprofv2 -i $O/in.txt -d $O/v2_stream -o out $D stream 256
try v1_stream rocprof -i $O/in.txt -o $O/v1_stream.csv $D stream 256
# if something worked on the stream case, go for the real kernels
for ctr in FETCH_SIZE WRITE_SIZE; do
  if ls $O/v3_stream_$ctr/*counter_collection.csv >/dev/null 2>&1 || find $O/v3_stream_$ctr -name "*counter*" | grep -q .; then
    try v3_unblocked_$ctr rocprofv3 --pmc $ctr --kernel-trace -d $O/v3_unblocked_$ctr -o out --output-format csv -- $D unblocked 4096
    try v3_blocked_$ctr rocprofv3 --pmc $ctr --kernel-trace -d $O/v3_blocked_$ctr -o out --output-format csv -- $D blocked 8192
  fi
done
find $O -size +8M -delete
find $O -type f | head -50
