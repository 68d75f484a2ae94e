#!/bin/bash
# round 5, call 3: coclr_gemm_fused as product + fold-with-row-op (second form): probe, kernel tests, model-level
# equivalence, same-box A/B of the step with the fused and the module-by-module head
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/head_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_head_probe.txt; grep -A3 "^fc1 forward\|^logits backward K=16384\|average-pool" gpurun_out/r05_head_probe.txt | head -30
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or l2norm" > gpurun_out/r05_c3_kernels.txt 2>&1; echo "kernels rc=$?"; tail -2 gpurun_out/r05_c3_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -s -k "fused_head or small_cases or config1" > gpurun_out/r05_c3_model.txt 2>&1; echo "model rc=$?"; grep -a "fused head\|passed\|failed\|Error" gpurun_out/r05_c3_model.txt | tail -8
for i in 1 2 3; do
  for f in 1 0; do
    COCLR_FUSED_HEAD=$f timeout 600 python bench.py --steps 30 --warmup 8 --no-extra-legs --no-cpu-baseline --no-self-check 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('fused=$f', r['value'], r['ms_per_step'], 'calls', r['abi_calls_per_step'], 'host', r['host_enqueue_ms_per_step'])"
  done
done | tee gpurun_out/r05_c3_ab.txt
