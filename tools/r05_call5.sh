#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
python tools/head_stage_probe.py 2>&1 | grep -a "fused\|l2norm" | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or l2norm" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_next.py tests/test_gpu_fullsize.py -x -q -s > gpurun_out/r05_c5_model.txt 2>&1; echo "model rc=$?"; grep -a "fused head\|passed\|failed\|Error" gpurun_out/r05_c5_model.txt | tail -8
