#!/bin/bash
# round 4, GPU visit 4: F(2x2,3x3)-domain weight gradient -- parity, per-layer timing, A/B; widened lockstep pairing
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "spatial_winograd_weight_gradient or temporal_winograd or wgrad" > gpurun_out/wino2_tests.log 2>&1
echo "rc=$?" >> gpurun_out/wino2_tests.log; tail -12 gpurun_out/wino2_tests.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_fullsize.py -x -q > gpurun_out/multi_tests.log 2>&1
echo "rc=$?" >> gpurun_out/multi_tests.log; tail -6 gpurun_out/multi_tests.log | cut -c1-300
for m in 0 1; do echo "== COCLR_WGRAD_WINO2=$m"; COCLR_WGRAD_WINO2=$m timeout 300 python tools/bench_layers.py Conv_2c.conv1 3b.b1.conv1 3c.b1.conv1 2>/dev/null | grep "conv1"; done > gpurun_out/r04_wino2_layers.txt 2>&1; cat gpurun_out/r04_wino2_layers.txt
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['abi_calls_per_step'])"; }
for i in 1 2 3; do
  COCLR_WGRAD_WINO2=0 timeout 200 $B > gpurun_out/ab_w2off$i.log 2>/dev/null; val gpurun_out/ab_w2off$i.log direct
  COCLR_WGRAD_WINO2=1 timeout 200 $B > gpurun_out/ab_w2on$i.log 2>/dev/null; val gpurun_out/ab_w2on$i.log wino2
done
