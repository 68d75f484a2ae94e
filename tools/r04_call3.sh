#!/bin/bash
# round 4, GPU visit 3: multi-problem launches -- bit-identity tests, engine regression, A/B, serial step sequence
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/multi_tests.log 2>&1
echo "rc=$?" >> gpurun_out/multi_tests.log; tail -15 gpurun_out/multi_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_next.py -x -q > gpurun_out/engine_tests.log 2>&1
echo "rc=$?" >> gpurun_out/engine_tests.log; tail -6 gpurun_out/engine_tests.log | cut -c1-300
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['abi_calls_per_step'])"; }
for i in 1 2 3; do
  COCLR_PAIR_UNITS=0 timeout 200 $B > gpurun_out/ab_unpaired$i.log 2>/dev/null; val gpurun_out/ab_unpaired$i.log unpaired
  COCLR_PAIR_UNITS=1 timeout 200 $B > gpurun_out/ab_paired$i.log 2>/dev/null; val gpurun_out/ab_paired$i.log paired
done
cd /tmp
OUT=$R/gpurun_out/prof_serial
rm -rf $OUT
COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o trace -- python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof_serial.log 2>&1
t=$(find $OUT -name '*kernel_trace.csv' | head -1)
python $R/tools/step_sequence.py $t $R/gpurun_out/r04_step_sequence_paired.txt; tail -22 $R/gpurun_out/r04_step_sequence_paired.txt
cp $(find $OUT -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04_serial_kernel_stats_paired.csv
find $OUT -name '*kernel_trace.csv' -size +30M -delete
