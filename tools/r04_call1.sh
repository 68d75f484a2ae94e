#!/bin/bash
# round 4, GPU visit 1: new gradient-parity tests, weight-gradient order A/B (per launch and whole step), per-layer table
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(nproc; free -g) > gpurun_out/host.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_gradients.py -x -q -s > gpurun_out/grad_tests.log 2>&1
echo "rc=$?" >> gpurun_out/grad_tests.log; tail -12 gpurun_out/grad_tests.log
timeout 400 python tools/wgrad_order_ab.py 3 > gpurun_out/wgrad_order.txt 2>gpurun_out/wgrad_order.err; tail -5 gpurun_out/wgrad_order.txt
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  COCLR_WGRAD_ORDER=split timeout 200 $B > gpurun_out/ab_split$i.log 2>/dev/null; val gpurun_out/ab_split$i.log split
  COCLR_WGRAD_ORDER=tile timeout 200 $B > gpurun_out/ab_tile$i.log 2>/dev/null; val gpurun_out/ab_tile$i.log tile
done
timeout 400 python tools/bench_layers.py > gpurun_out/layers.txt 2>&1; tail -3 gpurun_out/layers.txt
