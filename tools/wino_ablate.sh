#!/bin/bash
# timing ablations of conv_wino_hw_kernel (COCLR_WINO_DEBUG bits: 1 no window DMA, 2 no weight DMA,
# 4 no output transform/stores, 8 no statistics) on Conv_2c.conv1 and Mixed_3c.b1.conv1
for d in 0 1 2 3 4 8 12 15; do
  echo "== COCLR_WINO_DEBUG=$d"
  COCLR_WINO_DEBUG=$d python tools/bench_layers.py Conv_2c.conv1 3c.b1.conv1 2>&1 | grep "conv1"
done
