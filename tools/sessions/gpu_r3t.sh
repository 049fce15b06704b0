#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_complex.py -x -q 2>&1 | tail -3 ) > $O/pytest_complex.txt
{ for n in 8192 16384 12288; do timeout 300 python tools/c64_bench.py $n 64 2>&1 | grep -v amdgpu | tail -1; done; python tools/quick_bench.py 32768,128 2>&1 | grep '^{' | cut -c1-330; } > $O/c64.txt 2>&1
cat $O/pytest_complex.txt $O/c64.txt
