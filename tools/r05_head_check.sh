#!/bin/bash
# HEAD sanity: the streams / engine tests, rehearsal, bench line, smoke
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_next.py tests/test_gpu_model.py tests/test_gpu_bench_rehearsal.py -q 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r05_bench_head.txt 2>/dev/null; echo "bench rc=$?"; tail -1 gpurun_out/r05_bench_head.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
