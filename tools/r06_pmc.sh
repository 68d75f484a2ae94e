#!/bin/bash
# round 6: counters on HEAD -- whole-step table (every kernel) and the dominant kernel's layer (traffic json)
set -u
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/pmc_step.sh r06 > gpurun_out/r06_pmc_step.log 2>&1; tail -45 gpurun_out/r06_pmc_step.log | cut -c1-200
bash tools/pmc_layers.sh Conv_2c > gpurun_out/r06_pmc_dominant.txt 2>&1
python tools/traffic_json.py gpurun_out/r06_pmc_dominant.txt gpurun_out/r06_traffic.json | cut -c1-400
rm -rf gpurun_out/pmc
