#!/bin/bash
# A/B of the spatial Winograd kernel's window staging (COCLR_WINO_X16=0: 4-byte pieces) + parity tests
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -k "wino or Wino or adjoint or conv" 2>&1 | tail -4
for v in 1 0 1 0; do
  echo "== COCLR_WINO_X16=$v"
  COCLR_WINO_X16=$v python tools/bench_layers.py Conv_2c.conv1 3c.b1.conv1 3b.b1.conv1 2>&1 | grep "conv1"
done
