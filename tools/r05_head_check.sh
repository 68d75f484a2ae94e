#!/bin/bash
# HEAD sanity after the last bench / test edits: rehearsal cases (incl. CoCLR), bench line, smoke
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_bench_rehearsal.py tests/test_gpu_multirank.py -q -s 2>&1 | grep -a "N=2 rehearsal\|passed\|failed" | tail -8
timeout 600 python bench.py > gpurun_out/r05_bench_head.txt 2>/dev/null; echo "bench rc=$?"; tail -1 gpurun_out/r05_bench_head.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
