#!/bin/bash
# First GPU visit of the next round: parity cases of the two-waves-per-SIMD spatial Winograd kernel
# (coclr_conv_desc.algo = 2), then the (1,3,3) layer table with the default kernel and with it.
# Everything under a short timeout: the kernel has never run on hardware.
set -u
mkdir -p gpurun_out
COCLR_TEST_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q \
    -k "spatial_winograd" > gpurun_out/hw8_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/hw8_tests.log
tail -4 gpurun_out/hw8_tests.log
timeout 90 python tools/bench_layers.py conv1 > gpurun_out/hw8_layers_algo1.txt 2>&1
COCLR_WINOGRAD_HW=2 timeout 90 python tools/bench_layers.py conv1 > gpurun_out/hw8_layers_algo2.txt 2>&1
paste -d'\n' <(grep conv1 gpurun_out/hw8_layers_algo1.txt) <(grep conv1 gpurun_out/hw8_layers_algo2.txt)
