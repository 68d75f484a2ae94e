#!/bin/bash
# round 4, GPU visit 6: instruction-level software pipelining of the Winograd-domain weight gradients
set -u
mkdir -p gpurun_out
cp coclr_amd/csrc/build/lib_new.so coclr_amd/libcoclr_hip.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd or wgrad" > gpurun_out/call6_tests.log 2>&1; echo "rc=$?" >> gpurun_out/call6_tests.log; tail -4 gpurun_out/call6_tests.log | cut -c1-300
for l in head new; do cp coclr_amd/csrc/build/lib_$l.so coclr_amd/libcoclr_hip.so; echo "== $l"; python tools/bench_layers.py Conv_2c.conv 3b.b1.conv 3c.b1.conv 4f.b1.conv2 5c.b1.conv2 2>/dev/null | grep "conv"; done > gpurun_out/r04_pipe_layers.txt; cat gpurun_out/r04_pipe_layers.txt
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  cp coclr_amd/csrc/build/lib_head.so coclr_amd/libcoclr_hip.so; timeout 200 $B > gpurun_out/ab_ph$i.log 2>/dev/null; val gpurun_out/ab_ph$i.log head
  cp coclr_amd/csrc/build/lib_new.so coclr_amd/libcoclr_hip.so; timeout 200 $B > gpurun_out/ab_pn$i.log 2>/dev/null; val gpurun_out/ab_pn$i.log new
done
