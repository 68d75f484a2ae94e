#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r06_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_pytest_gpu.txt
grep -a "passed\|failed\|rc=" gpurun_out/r06_pytest_gpu.txt | tail -5
