#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_gradients.py tests/test_gpu_fullsize.py -x -q -s > gpurun_out/r06_inaff2_pytest.txt 2>&1; echo "pytest rc=$?"; grep -a "ratio\|worst\|passed\|failed\|Error" gpurun_out/r06_inaff2_pytest.txt | tail -12
for rep in 1 2 3; do
for v in 0 1; do
  COCLR_IN_AFFINE_BWD=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r06_ib_ab_${v}_$rep.txt 2> gpurun_out/r06_ib_ab_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_ib_ab_${v}_$rep.txt").read().strip().splitlines()[-1])
    print("IN_AFFINE_BWD=$v rep=$rep value", d["value"], "ms", d["ms_per_step"], "loss", d["config"].get("final_loss"), "selfcheck", d["self_check"]["passed"])
except Exception as e:
    print("IN_AFFINE_BWD=$v rep=$rep FAILED", e); print(open("gpurun_out/r06_ib_ab_${v}_$rep.err").read()[-2000:])
PY
done; done
