#!/bin/bash
# Round-3 evidence run on one MI355X: parity tier, bench line, kernel-trace stats (three-stream and
# single-stream), PMC passes on the dominant kernel.  Outputs under gpurun_out/ (copied to profiles/).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu.txt
grep -E "passed|failed|rc=" gpurun_out/r03_pytest_gpu.txt | tail -3
timeout 400 python bench.py > gpurun_out/r03_bench_stdout.txt 2> gpurun_out/r03_bench_stderr.txt
tail -1 gpurun_out/r03_bench_stdout.txt | cut -c1-300
bash tools/prof_bench.sh r03_bench > gpurun_out/r03_prof_summary.txt 2>&1
COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 bash tools/prof_bench.sh r03_serial > gpurun_out/r03_serial_summary.txt 2>&1
head -3 gpurun_out/r03_serial_summary.txt
bash tools/pmc_layers.sh Conv_2c.conv1 > gpurun_out/r03_pmc_dominant.txt 2>&1
head -24 gpurun_out/r03_pmc_dominant.txt
