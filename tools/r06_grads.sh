#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests/test_gpu_gradients.py -x -q -s -k "config5 or config4" > gpurun_out/r06_grads_pytest.txt 2>&1; echo "pytest rc=$?"; grep -a "tensors\|decisions\|passed\|failed\|Error\|assert" gpurun_out/r06_grads_pytest.txt | tail -20
