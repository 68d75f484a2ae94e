#!/bin/bash
# pooled backward with all arg-max / gradient loads in flight at once: parity, then old library against new
cd /root/repo; mkdir -p gpurun_out/libab
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "pool" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -2
cp coclr_amd/libcoclr_hip.so /tmp/lib_new.so
for r in 1 2; do for which in old new; do
  if [ $which = old ]; then cp coclr_amd/csrc/build_old/libcoclr_hip_old.so coclr_amd/libcoclr_hip.so; else cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so; fi
  echo "== $which run $r"; timeout 300 python tools/bench_layers.py pool Conv_1a.bn Conv_2c.bn 2>&1 | grep "pool\|bn" | cut -c1-80
done; done | tee gpurun_out/libab/pool.txt
cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so
