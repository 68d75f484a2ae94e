#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_gradients.py tests/test_gpu_fullsize.py -x -q -s > gpurun_out/r06_quick_pytest.txt 2>&1; echo "pytest rc=$?"; grep -a "ratio\|worst\|passed\|failed" gpurun_out/r06_quick_pytest.txt | tail -12
