#!/bin/bash
# Same-box alternating A/B of bench.py over the values of ONE environment switch (what the round-6 experiment
# loops did): tools/ab_bench.sh COCLR_WINO_T4 "0 1" [reps] [extra bench args]
# prints one line per run: <VAR>=<value> rep=<n> value <clips/s> ms <ms/step>
set -u
VAR=$1; VALS=$2; REPS=${3:-3}; shift; shift; shift || true
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs "$@" > gpurun_out/ab_$VAR.txt 2> gpurun_out/ab_$VAR.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_$VAR.txt").read().strip().splitlines()[-1])
    print("$VAR=$v rep=$rep value", d["value"], "ms", d["ms_per_step"], "self_check", (d.get("self_check") or {}).get("passed"))
except Exception as e:
    print("$VAR=$v rep=$rep FAILED", e); print(open("gpurun_out/ab_$VAR.err").read()[-800:])
PY
  done
done
