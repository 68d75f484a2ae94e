#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
for modes in "1,0" "0,0" "1,1" "1,0"; do
  COCLR_TEST_HEAD_MODES=$modes timeout 600 python -m pytest tests/test_gpu_model.py -x -q -s -k "fused_head and coclr" 2>&1 | grep -a "fused head\|passed\|failed\|Error" | sed "s/^/[$modes] /" | cut -c1-400
done | tee gpurun_out/r05_c4_flaky.txt
