#!/bin/bash
# round 6, call 1: bench line on the round-5 tree (this box's baseline) + in-step kernel trace
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_base_bench.txt 2> gpurun_out/r06_base_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r06_base_bench.txt | cut -c1-400
bash tools/prof_bench.sh r06base --no-self-check > gpurun_out/r06_base_prof.txt 2>&1; head -30 gpurun_out/r06_base_prof.txt
