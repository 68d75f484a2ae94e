#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
tail -1 gpurun_out/r06_bench_stdout.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read()); sc=r.get('self_check') or {}
print('value',r['value'],'ms',r['ms_per_step'],'unmod',r['value_unmodified_caller']['value'],'split',r['value_split_stages']['value'],'k16',r['value_k16384']['value'],'caller_opt',r['value_caller_optimizer']['value'])
print('self_check',sc.get('passed'),'host_enqueue',r['host_enqueue_ms_per_step'],'floor',r['host_floor_ms_per_step'],'calls',r['abi_calls_per_step'],'plans',r['launch_plans']['recorded'],r['launch_plans']['disabled'])
print('roof',r['roofline']['frac'],r['roofline']['avg_launch_ms'],r['roofline'].get('traffic'),r['roofline'].get('traffic_note'))
print('step',{k:v for k,v in r['step_roofline'].items() if k!='what'})
print('hbm',r['roofline_hbm']['frac'],r['roofline_hbm']['isolated'])
print('cpu',r['cpu_baseline']['value'],r['cpu_baseline']['cores'],r.get('cpu_baseline_all_cores'))
"
t0=$(date +%s); timeout 900 python bench.py > gpurun_out/r06_bench_default.txt 2>/dev/null; echo "default bench rc=$? in $(( $(date +%s) - t0 )) s"; tail -1 gpurun_out/r06_bench_default.txt | cut -c1-200
