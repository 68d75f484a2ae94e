#!/bin/bash
# round 6, last call: the parity tier and the driver-style bench line on the final tree, smoke
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r06_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_pytest_gpu.txt
grep -a "passed\|failed\|rc=" gpurun_out/r06_pytest_gpu.txt | tail -3
t0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
tail -1 gpurun_out/r06_bench_stdout.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read()); sc=r.get('self_check') or {}
print('value',r['value'],'ms',r['ms_per_step'],'unmod',r['value_unmodified_caller']['value'],'split',r['value_split_stages']['value'],'k16',r['value_k16384']['value'],'caller_opt',r['value_caller_optimizer']['value'])
print('self_check',sc.get('passed'),'floor',r['host_floor_ms_per_step'],'roof',r['roofline']['frac'],r['roofline']['isolated']['frac'],'traffic' , (r['roofline']['traffic'] or {}).get('bytes_per_launch'), 'issued', r['step_roofline']['mfma_issued_frac'])
"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
