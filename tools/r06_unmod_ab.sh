#!/bin/bash
# does the fused stem backward change the unmodified caller's share of `value`?  full bench lines, one box, alternating
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for v in 0 1; do
COCLR_WGRAD_BN=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/unmod_ab.txt 2>/dev/null
python - <<PY
import json
r=json.loads(open("gpurun_out/unmod_ab.txt").read().strip().splitlines()[-1])
u=r["value_unmodified_caller"]
print("WGRAD_BN=$v rep=$rep value", r["value"], "unmod", u["value"], "ratio", round(u["value"]/r["value"],4), "ms", r["ms_per_step"], u["ms_per_step"])
PY
done; done
