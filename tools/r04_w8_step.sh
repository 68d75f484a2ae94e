#!/bin/bash
# in-step effect of the two-waves-per-SIMD spatial Winograd kernel: model-level parity, then alternating bench runs
cd /root/repo; mkdir -p gpurun_out/w8
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_multi.py -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|ProcessGroupNCCL\|amdgpu.ids" | tail -5 > gpurun_out/w8/pytest_model.txt
tail -2 gpurun_out/w8/pytest_model.txt
for r in 1 2 3; do for d in 0 1; do
  COCLR_WINO_W8=$d timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | grep '^{' > gpurun_out/w8/bench_w${d}_r${r}.json
  python - <<PY
import json
r = json.load(open("gpurun_out/w8/bench_w${d}_r${r}.json"))
print("w8=${d} run ${r}: value", r["value"], "ms", r["ms_per_step"], "dominant frac", r["roofline"]["frac"])
PY
done; done | tee gpurun_out/w8/step_ab.txt
