#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_engine.py -x -q 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "== COCLR_CONV_XG=$v"
  COCLR_CONV_XG=$v python tools/bench_layers.py 4f.b1.conv1 4b.b2.conv1 4c.b1.conv1 5c.b1.conv1 2>&1 | grep "conv1"
done
for v in 1 0; do COCLR_CONV_XG=$v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xg=$v', d['value'], d['ms_per_step'])"; done
