#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for fam in none bn_apply bn_bwd bn_multi pool wgrad dgrad none; do
  COCLR_ABLATE=$fam timeout 600 python tools/ablate_step.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-self-check > gpurun_out/r06_ablate_$fam.txt 2> gpurun_out/r06_ablate_$fam.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_ablate_$fam.txt").read().strip().splitlines()[-1])
    print("ablate=$fam ms_per_step", d["ms_per_step"])
except Exception as e:
    print("ablate=$fam FAILED", e); print(open("gpurun_out/r06_ablate_$fam.err").read()[-600:])
PY
done
