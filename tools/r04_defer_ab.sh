#!/bin/bash
# A/B on one box: the per-stage autograd structure (world > 1) with and without the deferred join of the
# weight-gradient stream.  bench.py reports value_split_stages; alternate the switch three times.
cd /root/repo; mkdir -p gpurun_out/defer
timeout 900 python -m pytest tests/test_gpu_next.py::test_gradients_in_ddp_buckets_bit_identical_and_deterministic \
    tests/test_gpu_multirank.py tests/test_gpu_bench_rehearsal.py -x -q 2>&1 | tail -5 | tee gpurun_out/defer/pytest.txt
for r in 1 2 3; do for d in 0 1; do
  COCLR_DEFER_JOIN=$d timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/defer/bench_d${d}_r${r}.json
  python - <<PY
import json
r = json.load(open("gpurun_out/defer/bench_d${d}_r${r}.json"))
print("defer=${d} run ${r}: value", r["value"], "split", r.get("value_split_stages"))
PY
done; done | tee gpurun_out/defer/ab.txt
