#!/bin/bash
# round 5, call 6: the new B=32 parity cases; what costs the unmodified caller its 5-10 %; a traced step on HEAD
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest "tests/test_gpu_gradients.py::test_config2_backbone_gradients_at_benchmarked_size" "tests/test_gpu_fullsize.py::test_ubernce_training_step_at_config2_size" -x -q -s > gpurun_out/r05_c6_tests.txt 2>&1; echo "tests rc=$?"; grep -a "B=32\|passed\|failed\|Error" gpurun_out/r05_c6_tests.txt | cut -c1-400 | tail -8
show() { tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('$1', 'value', r['value'], 'unmodified', (r.get('value_unmodified_caller') or {}).get('value'), 'split', (r.get('value_split_stages') or {}).get('value'), 'k16', (r.get('value_k16384') or {}).get('value'), 'host', r['host_enqueue_ms_per_step'], r['host_floor_ms_per_step'])"; }
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | show default
timeout 600 python bench.py --no-cpu-baseline --no-self-check 2>/dev/null | show no-self-check
COCLR_GRAPH_QUERY=late timeout 600 python bench.py --no-cpu-baseline --no-self-check 2>/dev/null | show graph-late
COCLR_GRAPH_QUERY=1 timeout 600 python bench.py --no-cpu-baseline --no-self-check 2>/dev/null | show graph-all
bash tools/prof_bench.sh r05 > gpurun_out/r05_prof_summary.txt 2>&1; head -3 gpurun_out/r05_prof_summary.txt | cut -c1-200; head -12 gpurun_out/r05_timeline.txt
