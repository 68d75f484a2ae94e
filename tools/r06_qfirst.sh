#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r06_qf_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06_qf_pytest.txt
for rep in 1 2 3; do
for v in 0 1; do
  COCLR_QUERY_FIRST=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_qf_ab_${v}_$rep.txt 2> gpurun_out/r06_qf_ab_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_qf_ab_${v}_$rep.txt").read().strip().splitlines()[-1])
    print("QUERY_FIRST=$v rep=$rep value", d["value"], "ms", d["ms_per_step"], "unmodified", d["value_unmodified_caller"]["value"], "split", d["value_split_stages"]["value"], "k16", d["value_k16384"]["value"], "floor", d["host_floor_ms_per_step"], "selfcheck", d["self_check"]["passed"])
except Exception as e:
    print("QUERY_FIRST=$v rep=$rep FAILED", e); print(open("gpurun_out/r06_qf_ab_${v}_$rep.err").read()[-2000:])
PY
done; done
