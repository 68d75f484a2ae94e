#!/bin/bash
# round 6, end: counters on HEAD (whole-step table + the dominant kernel's layer), the bench line with every leg
# and both CPU baselines (AFTER the counter files are in place, so that it can price them), BASELINE configs 3/4/5
# on one GPU, the serialised step sequence, smoke, and the N=2 one-GPU rehearsal line
set -u
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/pmc_step.sh r06 > gpurun_out/r06_pmc_step.log 2>&1; tail -3 gpurun_out/r06_pmc_step.log | cut -c1-300
bash tools/pmc_layers.sh Conv_2c > gpurun_out/r06_pmc_dominant.txt 2>&1
python tools/traffic_json.py gpurun_out/r06_pmc_dominant.txt gpurun_out/r06_traffic.json | cut -c1-300
rm -rf gpurun_out/pmc gpurun_out/pmc_step
cp gpurun_out/r06_step_pmc.json gpurun_out/r06_traffic.json profiles/      # this copy of the tree only: bench.py reads profiles/
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
tail -1 gpurun_out/r06_bench_stdout.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read()); sc=r.get('self_check') or {}
print('value',r['value'],'ms',r['ms_per_step'],'unmod',r['value_unmodified_caller']['value'],'split',r['value_split_stages']['value'],'k16',r['value_k16384']['value'],'caller_opt',r['value_caller_optimizer']['value'])
print('self_check',sc.get('passed'),'host_enqueue',r['host_enqueue_ms_per_step'],'floor',r['host_floor_ms_per_step'],'calls',r['abi_calls_per_step'],'plans',r['launch_plans']['recorded'],r['launch_plans']['disabled'])
print('roof',r['roofline']['frac'],r['roofline']['avg_launch_ms'],r['roofline'].get('traffic'),r['roofline'].get('traffic_note'))
print('step',r['step_roofline'])
print('cpu',r['cpu_baseline']['value'],r['cpu_baseline']['cores'],(r.get('cpu_baseline_all_cores') or {}).get('value'),(r.get('cpu_baseline_all_cores') or {}).get('cores'))
"
show() { tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); sc=r.get('self_check') or {}
mg=r.get('multi_gpu') or {}
print('$1', r['metric'][:60], 'value', r['value'], 'ms', r['ms_per_step'], 'self_check', sc.get('passed'), 'tensors', sc.get('tensors_compared'), 'rung', mg.get('rung'), mg.get('shuffle_mode'), 'plans', (r.get('launch_plans') or {}).get('recorded'), (r.get('launch_plans') or {}).get('disabled'))"; }
timeout 600 python bench.py --moco-k 16384 --no-cpu-baseline --no-extra-legs 2>/dev/null | tee gpurun_out/r06_bench_cfg3_k16384.txt | show cfg3
timeout 600 python bench.py --model coclr --no-cpu-baseline --no-extra-legs 2>/dev/null | tee gpurun_out/r06_bench_cfg4_coclr.txt | show cfg4
timeout 600 python bench.py --net r50 --moco-k 16384 --no-cpu-baseline --no-extra-legs 2>/dev/null | tee gpurun_out/r06_bench_cfg5_r50.txt | show cfg5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29790 tests/bench_rehearse_gpu.py --gpus 2 --steps 10 --warmup 3 --batch 16 --moco-k 2048 2> gpurun_out/r06_rehearsal_n2.err | tail -1 > gpurun_out/r06_rehearsal_n2.json; cat gpurun_out/r06_rehearsal_n2.json | show n2
COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 COCLR_PLAN=0 bash tools/prof_bench.sh r06serial --no-self-check > gpurun_out/r06_serial_summary.txt 2>&1
t=$(find gpurun_out/prof_r06serial -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $t gpurun_out/r06_step_sequence.txt; head -1 gpurun_out/r06_step_sequence.txt
cp gpurun_out/r06serial_kernel_stats.csv gpurun_out/r06_serial_kernel_stats.csv
bash tools/prof_bench.sh r06bench --no-self-check > gpurun_out/r06_bench_prof.txt 2>&1; sed -n 1,6p gpurun_out/r06bench_timeline.txt | cut -c1-200
find gpurun_out -name '*kernel_trace.csv' -size +5M -delete
