#!/bin/bash
# round 5, call 1: the new N>1 bench machinery with the real kernels (two ranks on one GPU), the exchange
# scheme chosen at run time, and the N=1 line with the self-check and the K=16384 leg
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_bench_rehearsal.py tests/test_gpu_multirank.py -x -q -s > gpurun_out/r05_c1_pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_c1_pytest.txt
grep -a "N=2 rehearsal\|passed\|failed\|rc=" gpurun_out/r05_c1_pytest.txt | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_c1_bench.txt 2> gpurun_out/r05_c1_bench.err
echo "bench rc=$?"
tail -1 gpurun_out/r05_c1_bench.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read())
sc=r.get('self_check') or {}
print('value',r['value'],'ms',r['ms_per_step'],'unmod',(r.get('value_unmodified_caller') or {}).get('value'),'split',(r.get('value_split_stages') or {}).get('value'), (r.get('value_split_stages') or {}).get('host_floor_ms_per_step'),'k16',(r.get('value_k16384') or {}).get('value'))
print('self_check',sc.get('passed'),sc.get('rung'),[ (t['rung'],t['bit_identical_to_serial_on_every_rank'],t['this_rank']['first_mismatch'],t['this_rank']['max_abs_diff']) for t in sc.get('trials',[])])
print('host_enqueue',r['host_enqueue_ms_per_step'],'floor',r['host_floor_ms_per_step'],'calls',r['abi_calls_per_step'])
print('roof',r['roofline']['frac'],r['roofline']['avg_launch_ms'],r['roofline']['kernel'][:60])
print('cpu',r['cpu_baseline']['value'])
"
tail -5 gpurun_out/r05_c1_bench.err
