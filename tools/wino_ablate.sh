#!/bin/bash
# compile-time timing ablations of conv_wino_hw_kernel: build with FLAGS="... -DCOCLR_WINO_ABLATE" (csrc/build.sh),
# 1 no window DMA, 2 no weight DMA, 4 no output transform/stores/statistics, 8 no statistics,
# 16 no patch LDS reads + input transform, 32 no weight LDS reads
for d in 0 1 2 3 8 16 32 48 0; do
  echo "== COCLR_WINO_ABL=$d"
  COCLR_WINO_ABL=$d python tools/bench_layers.py Conv_2c.conv1 3c.b1.conv1 2>&1 | grep "conv1"
done
