#!/bin/bash
# round 4 evidence on one box: full parity tier, bench line, kernel traces (three streams / one), PMC on the dominant
# kernel and the NCE logits kernel, other configs, smoke
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest_gpu.txt; tail -3 gpurun_out/r04_pytest_gpu.txt | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_stdout.txt 2>gpurun_out/r04_bench.err; tail -1 gpurun_out/r04_bench_stdout.txt | cut -c1-400
bash tools/prof_bench.sh r04_bench > gpurun_out/r04_prof_bench.log 2>&1; tail -3 gpurun_out/r04_bench_timeline.txt
OUT=$R/gpurun_out/prof_serial; rm -rf $OUT
(cd /tmp && COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o trace -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof_serial.log 2>&1)
t=$(find $OUT -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $t gpurun_out/r04_step_sequence.txt; tail -20 gpurun_out/r04_step_sequence.txt; head -1 gpurun_out/r04_step_sequence.txt
cp $(find $OUT -name '*kernel_stats.csv' | head -1) gpurun_out/r04_serial_kernel_stats.csv
find $OUT -name '*kernel_trace.csv' -size +30M -delete
bash tools/pmc_layers.sh Conv_2c.conv1 > gpurun_out/r04_pmc_dominant.txt 2>&1
python tools/traffic_json.py gpurun_out/r04_pmc_dominant.txt gpurun_out/r04_traffic.json | cut -c1-300
mv gpurun_out/pmc gpurun_out/pmc_dominant
KS=16384 PMC_SCRIPT=tools/bench_nce.py bash tools/pmc_layers.sh > gpurun_out/r04_pmc_nce_head.txt 2>&1
python tools/nce_pmc_json.py gpurun_out/r04_pmc_nce_head.txt gpurun_out/pmc gpurun_out/r04_nce_pmc.json | cut -c1-300
python bench.py --steps 10 --moco-k 16384 --no-cpu-baseline > gpurun_out/r04_bench_cfg3_k16384.txt 2>/dev/null; tail -1 gpurun_out/r04_bench_cfg3_k16384.txt | cut -c1-140
python bench.py --steps 10 --model coclr --no-cpu-baseline > gpurun_out/r04_bench_cfg4_coclr.txt 2>/dev/null; tail -1 gpurun_out/r04_bench_cfg4_coclr.txt | cut -c1-140
python bench.py --steps 10 --net r50 --moco-k 16384 --no-cpu-baseline > gpurun_out/r04_bench_cfg5_r50.txt 2>/dev/null; tail -1 gpurun_out/r04_bench_cfg5_r50.txt | cut -c1-140
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
rm -rf gpurun_out/pmc gpurun_out/pmc_dominant gpurun_out/prof_serial gpurun_out/prof_r04_bench
