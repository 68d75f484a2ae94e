#!/bin/bash
# timing ablations of conv_wino_hw8_kernel (wrong results by design): library built with -DCOCLR_WINO_ABLATE
# swapped in for the run.  1 no DMA, 2 no output stores, 4 no LDS operand reads
cd /root/repo
cp coclr_amd/libcoclr_hip.so /tmp/lib_keep.so
cp coclr_amd/csrc/build_abl/libcoclr_hip_abl.so coclr_amd/libcoclr_hip.so
for d in 0 1 2 3 4 7 0; do
  echo "== COCLR_W8_ABL=$d"
  COCLR_WINO_W8=1 COCLR_W8_ABL=$d timeout 120 python tools/bench_layers.py Conv_2c.conv1 3c.b1.conv1 2>&1 | grep "conv1" | cut -c1-70
done | tee gpurun_out/w8/ablate.txt
cp /tmp/lib_keep.so coclr_amd/libcoclr_hip.so
