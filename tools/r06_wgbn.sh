#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "stem_weight_gradient" 2>&1 | tail -2
timeout 300 python tools/stem_wgrad_bn_probe.py 2>&1 | tail -5
bash tools/ab_bench.sh COCLR_WGRAD_BN "0 1" 3
