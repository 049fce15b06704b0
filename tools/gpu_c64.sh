#!/bin/bash
# One-shot GPU check of the ComplexF64 path: torch-free first light, then the gpu-marked complex tests.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 150 python tools/gpu_c64_check.py > gpurun_out/c64_check.json 2> gpurun_out/c64_check.err
echo "check rc=$?" | tee gpurun_out/c64_rc.txt
tail -c 1500 gpurun_out/c64_check.json
timeout 230 python -m pytest tests/test_gpu_complex.py -q -m gpu -x --deselect "tests/test_gpu_complex.py::test_reference_acceptance_inequality_complex[4400-4000]" > gpurun_out/c64_pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/c64_rc.txt
tail -15 gpurun_out/c64_pytest.txt
