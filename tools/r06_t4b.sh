#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multi.py tests/test_gpu_bench_rehearsal.py -x -q -k "temporal_winograd or multi or pair or sums or rehearsal" > gpurun_out/r06_t4_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r06_t4_pytest.txt
for t4 in 0 1; do
  echo "== COCLR_WINO_T4=$t4 layers"
  COCLR_WINO_T4=$t4 timeout 300 python tools/bench_layers.py conv2 2>&1 | grep -v "^$\|pool\|bn unit\|pooled\|amdgpu.ids" | head -12
done
for rep in 1 2 3; do
for t4 in 0 1; do
  COCLR_WINO_T4=$t4 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r06_t4_ab_${t4}_$rep.txt 2> gpurun_out/r06_t4_ab_${t4}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_t4_ab_${t4}_$rep.txt").read().strip().splitlines()[-1])
    print("T4=$t4 rep=$rep value", d["value"], "ms", d["ms_per_step"], "loss", d["config"].get("final_loss"), "selfcheck", d["self_check"]["passed"])
except Exception as e:
    print("T4=$t4 rep=$rep FAILED", e); print(open("gpurun_out/r06_t4_ab_${t4}_$rep.err").read()[-2000:])
PY
done; done
