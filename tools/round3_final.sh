#!/bin/bash
# final evidence of the round on one box: full parity tier, bench line, traces, PMC, other configs
bash tools/round3_measure.sh
python bench.py --steps 10 --moco-k 16384 --no-cpu-baseline > gpurun_out/r03_bench_cfg3_k16384.txt 2>/dev/null; tail -1 gpurun_out/r03_bench_cfg3_k16384.txt | cut -c1-140
python bench.py --steps 10 --model coclr --no-cpu-baseline > gpurun_out/r03_bench_cfg4_coclr.txt 2>/dev/null; tail -1 gpurun_out/r03_bench_cfg4_coclr.txt | cut -c1-140
python bench.py --steps 10 --net r50 --moco-k 16384 --no-cpu-baseline > gpurun_out/r03_bench_cfg5_r50.txt 2>/dev/null; tail -1 gpurun_out/r03_bench_cfg5_r50.txt | cut -c1-140
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
