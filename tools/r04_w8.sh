#!/bin/bash
# two-waves-per-SIMD spatial Winograd kernel (COCLR_WINO_W8=1): parity, then per-layer timing against the one-wave kernel
cd /root/repo; mkdir -p gpurun_out/w8
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "spatial_winograd and not weight_gradient" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -25 > gpurun_out/w8/pytest.txt
tail -3 gpurun_out/w8/pytest.txt
for r in 1 2; do for w in 0 1; do
  echo "== W8=$w run $r"; COCLR_WINO_W8=$w timeout 200 python tools/bench_layers.py Conv_2c.conv1 3b.b1.conv1 3c.b1.conv1 3b.b2.conv1 2>&1 | grep -v amdgpu.ids | cut -c1-100
done; done | tee gpurun_out/w8/layers.txt
