#!/bin/bash
# round 4, GPU visit 5: widened lockstep pairing + re-boxed Winograd-domain wgrad: parity, then mode 1 vs 2
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q -k "multi or pairs or paired or batchnorm_multi or adjoint or winograd_weight or config2" > gpurun_out/call5_tests.log 2>&1
echo "rc=$?" >> gpurun_out/call5_tests.log; tail -8 gpurun_out/call5_tests.log | cut -c1-400
for m in 1 2; do echo "== COCLR_WGRAD_WINO2=$m"; COCLR_WGRAD_WINO2=$m timeout 300 python tools/bench_layers.py Conv_2c.conv1 3c.b1.conv1 4f.b1.conv1 4c.b1.conv1 5c.b1.conv1 2>/dev/null | grep "conv1"; done > gpurun_out/r04_wino2_modes.txt 2>&1; cat gpurun_out/r04_wino2_modes.txt
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['abi_calls_per_step'])"; }
for i in 1 2 3; do
  COCLR_WGRAD_WINO2=1 timeout 200 $B > gpurun_out/ab_m1_$i.log 2>/dev/null; val gpurun_out/ab_m1_$i.log mode1
  COCLR_WGRAD_WINO2=2 timeout 200 $B > gpurun_out/ab_m2_$i.log 2>/dev/null; val gpurun_out/ab_m2_$i.log mode2
done
