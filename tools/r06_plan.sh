#!/bin/bash
# round 6: launch-plan replay -- parity test, then bench A/B on one box (COCLR_PLAN=0 / 1, alternating)
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "planned or graphed_query" > gpurun_out/r06_plan_pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06_plan_pytest.txt
for rep in 1 2; do
for plan in 0 1; do
  COCLR_PLAN=$plan timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_plan_ab_${plan}_$rep.txt 2> gpurun_out/r06_plan_ab_${plan}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_plan_ab_${plan}_$rep.txt").read().strip().splitlines()[-1])
    print("plan=$plan rep=$rep value", d["value"], "unmodified", d["value_unmodified_caller"]["value"], "split", d["value_split_stages"]["value"], "k16", d["value_k16384"]["value"], "host", d["host_enqueue_ms_per_step"], d["host_floor_ms_per_step"], "abi", d["abi_calls_per_step"], "plans", d["launch_plans"]["recorded"], d["launch_plans"]["replayed_passes"], d["launch_plans"]["disabled"], "selfcheck", d["self_check"]["passed"])
except Exception as e:
    print("plan=$plan rep=$rep FAILED", e)
    import subprocess; print(open("gpurun_out/r06_plan_ab_${plan}_$rep.err").read()[-3000:])
PY
done; done
