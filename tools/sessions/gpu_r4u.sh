#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/tall_check.py 32768,24 20000,64 24577,21 32790,40 16400,36 40000,20 > gpurun_out/r4u_xtall_check.txt 2>&1
tail -7 gpurun_out/r4u_xtall_check.txt
timeout 1500 python tools/xtall_bench.py > gpurun_out/r4u_xtall_bench.txt 2>&1
cat gpurun_out/r4u_xtall_bench.txt | grep -v amdgpu
