#!/bin/bash
# round 6: BatchNorm backward apply inside the stem weight gradient -- parity tier for the touched paths
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_gradients.py tests/test_abi.py tests/test_gpu_engine.py -x -q -k "stem_weight_gradient or small_cases or config1 or planned or abi or every_gradient or config2_backbone or config4 or s3d_backbone" > gpurun_out/r06_wgbn_pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r06_wgbn_pytest.txt
