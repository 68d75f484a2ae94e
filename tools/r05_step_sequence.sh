#!/bin/bash
# serialised trace of one step on HEAD: launch count and per-segment kernel time (tools/step_sequence.py)
set -u
cd /root/repo; mkdir -p gpurun_out
COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 bash tools/prof_bench.sh r05serial --no-self-check > gpurun_out/r05_serial_summary.txt 2>&1
t=$(find gpurun_out/prof_r05serial -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $t gpurun_out/r05_step_sequence.txt; head -2 gpurun_out/r05_step_sequence.txt; grep -n "segment" gpurun_out/r05_step_sequence.txt | tail -22
cp gpurun_out/r05serial_kernel_stats.csv gpurun_out/r05_serial_kernel_stats.csv
