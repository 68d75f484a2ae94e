#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/prof_bench.sh r06mid --no-self-check > gpurun_out/r06_mid_prof.txt 2>&1; head -34 gpurun_out/r06_mid_prof.txt
COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 COCLR_PLAN=0 bash tools/prof_bench.sh r06serial --no-self-check > gpurun_out/r06_serial_summary.txt 2>&1
t=$(find gpurun_out/prof_r06serial -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $t gpurun_out/r06_step_sequence.txt; head -2 gpurun_out/r06_step_sequence.txt; grep -n "segment" gpurun_out/r06_step_sequence.txt | tail -20
