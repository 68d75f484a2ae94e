#!/bin/bash
# round 5, call 2: the fused head (coclr_gemm_fused): kernel tests, model-level equivalence, parity tier of
# the model tests, and a same-box A/B of the step with the fused and the module-by-module head
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or l2norm" > gpurun_out/r05_c2_kernels.txt 2>&1; echo "kernels rc=$?"; tail -3 gpurun_out/r05_c2_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_gradients.py tests/test_gpu_next.py -x -q -s > gpurun_out/r05_c2_model.txt 2>&1; echo "model rc=$?"; grep -a "fused head\|passed\|failed\|Error" gpurun_out/r05_c2_model.txt | tail -8
for i in 1 2 3; do
  for f in 1 0; do
    COCLR_FUSED_HEAD=$f timeout 600 python bench.py --steps 30 --warmup 8 --no-extra-legs --no-cpu-baseline --no-self-check 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('fused=$f', r['value'], r['ms_per_step'], 'calls', r['abi_calls_per_step'], 'host', r['host_enqueue_ms_per_step'])"
  done
done | tee gpurun_out/r05_c2_ab.txt
