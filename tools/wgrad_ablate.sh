#!/bin/bash
# Is a weight-gradient kernel paced by its loader waves?  Same box: the shipped library against a build with
# -DCOCLR_WGRAD_ABLATE=1 (no window DMA after the first two boxes: wrong results, timing only).
for l in head abl; do cp coclr_amd/csrc/build/lib_$l.so coclr_amd/libcoclr_hip.so; echo "== $l"; python tools/bench_layers.py Conv_1a.conv2 Conv_2c.conv1 Conv_2c.conv2 3c.b1.conv 4f.b1.conv 2>/dev/null | grep "conv"; done
cp coclr_amd/csrc/build/lib_head.so coclr_amd/libcoclr_hip.so
