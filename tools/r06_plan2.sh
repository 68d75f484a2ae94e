#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_next.py tests/test_gpu_dropin.py -x -q > gpurun_out/r06_plan_pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06_plan_pytest.txt
