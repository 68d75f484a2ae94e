#!/bin/bash
# round 4, GPU visit 2: gradient tests again (timings), weight-gradient traffic per order (PMC), serial step sequence,
# PMC table over the long kernels
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_gradients.py -q -s > gpurun_out/grad_tests.log 2>&1
echo "rc=$?" >> gpurun_out/grad_tests.log; grep -v Warn gpurun_out/grad_tests.log | grep "times\|decisions\|tensors;\|B=32\|passed\|failed\|product:\|oracle fp32:" | cut -c1-400
cd /tmp
for order in split tile; do
  OUT=$R/gpurun_out/pmc_wgrad_$order
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    COCLR_WGRAD_ORDER=$order timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/p$i -o pmc -- python $R/tools/bench_layers.py Conv_1a.conv2 Conv_2c.conv1 Conv_2c.conv2 > $OUT/p$i.log 2>&1
  done
  echo "== COCLR_WGRAD_ORDER=$order" >> $R/gpurun_out/r04_pmc_wgrad.txt
  python $R/tools/pmc_summary.py $OUT >> $R/gpurun_out/r04_pmc_wgrad.txt 2>&1
done
grep -A3 "wgrad\|ORDER" $R/gpurun_out/r04_pmc_wgrad.txt | head -60
OUT=$R/gpurun_out/prof_serial
rm -rf $OUT
COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o trace -- python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof_serial.log 2>&1
t=$(find $OUT -name '*kernel_trace.csv' | head -1)
python $R/tools/step_sequence.py $t $R/gpurun_out/r04_step_sequence_start.txt; tail -14 $R/gpurun_out/r04_step_sequence_start.txt
cp $(find $OUT -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04_serial_kernel_stats_start.csv
find $OUT -name '*kernel_trace.csv' -size +30M -delete
cd $R
PMC_SCRIPT=tools/bench_layers.py bash tools/pmc_layers.sh Conv_1a Conv_2c 3c.b1 4f.b1 > gpurun_out/r04_pmc_layers.txt 2>&1; tail -5 gpurun_out/r04_pmc_layers.txt
