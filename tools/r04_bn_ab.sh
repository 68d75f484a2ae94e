#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/libab
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "bn or batchnorm or norm" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -1
cp coclr_amd/libcoclr_hip.so /tmp/lib_new.so
for r in 1 2; do for which in old new; do
  if [ $which = old ]; then cp coclr_amd/csrc/build_old/libcoclr_hip_old.so coclr_amd/libcoclr_hip.so; else cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so; fi
  echo "== $which run $r"; timeout 300 python tools/bench_layers.py Conv_1a.bn Conv_2c.bn 3b.bn 3c.bn 2>&1 | grep "bn" | cut -c1-80
done; done | tee gpurun_out/libab/bn.txt
cp /tmp/lib_new.so coclr_amd/libcoclr_hip.so
