#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rfEs -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu8.txt
tail -25 gpurun_out/pytest_gpu8.txt
