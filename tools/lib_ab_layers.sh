#!/bin/bash
# A/B of two builds of the library on one box: coclr_amd/csrc/build_old/libcoclr_hip_old.so against the in-tree one
cd /root/repo; mkdir -p gpurun_out/libab
cp coclr_amd/libcoclr_hip.so /tmp/lib_new.so
LAYERS="Conv_1a.conv1"
for r in 1 2; do for which in old new; do
  if [ $which = old ]; then cp coclr_amd/csrc/build_old/libcoclr_hip_old.so coclr_amd/libcoclr_hip.so; else cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so; fi
  echo "== $which run $r"; timeout 300 python tools/bench_layers.py $LAYERS 2>&1 | grep "^Conv\|^3\|^4\|^5" | cut -c1-75
done; done | tee gpurun_out/libab/layers.txt
cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so
for r in 1 2 3; do for which in old new; do
  if [ $which = old ]; then cp coclr_amd/csrc/build_old/libcoclr_hip_old.so coclr_amd/libcoclr_hip.so; else cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | grep '^{' > gpurun_out/libab/bench_${which}_r${r}.json
  python - <<PY
import json
r = json.load(open("gpurun_out/libab/bench_${which}_r${r}.json"))
print("${which} run ${r}: value", r["value"], "ms", r["ms_per_step"])
PY
done; done | tee gpurun_out/libab/step.txt
cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so
