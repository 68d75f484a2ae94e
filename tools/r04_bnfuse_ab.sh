#!/bin/bash
# BatchNorm backward sums in the data gradient's epilogue (COCLR_FUSE_BN_REDUCE): parity tests, then an
# alternating A/B of bench.py on one box.
cd /root/repo; mkdir -p gpurun_out/bnfuse
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "sums or phases or pair_launches" 2>&1 | tail -15 > gpurun_out/bnfuse/pytest_kernels.txt
cat gpurun_out/bnfuse/pytest_kernels.txt | tail -3
timeout 1200 python -m pytest tests/test_gpu_gradients.py tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_next.py::test_gradients_in_ddp_buckets_bit_identical_and_deterministic tests/test_gpu_multirank.py tests/test_gpu_bench_rehearsal.py -x -q 2>&1 | tail -15 > gpurun_out/bnfuse/pytest_model.txt
tail -3 gpurun_out/bnfuse/pytest_model.txt
for r in 1 2 3; do for d in 0 1; do
  COCLR_FUSE_BN_REDUCE=$d timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | grep '^{' > gpurun_out/bnfuse/bench_f${d}_r${r}.json
  python - <<PY
import json
r = json.load(open("gpurun_out/bnfuse/bench_f${d}_r${r}.json"))
print("fuse=${d} run ${r}: value", r["value"], "ms", r["ms_per_step"])
PY
done; done | tee gpurun_out/bnfuse/ab.txt
