#!/bin/bash
# BASELINE configs 3 / 4 / 5 on one GPU on HEAD, each with its self-check; CoCLR through the N=2 one-GPU rehearsal
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); sc=r.get('self_check') or {}
mg=r.get('multi_gpu') or {}
print('$1', r['metric'][:60], 'value', r['value'], 'ms', r['ms_per_step'], 'self_check', sc.get('passed'), [(t['rung'], t['bit_identical_to_serial_on_every_rank'], t['this_rank']['first_mismatch']) for t in sc.get('trials', [])], 'tensors', sc.get('tensors_compared'), 'rung', mg.get('rung'), mg.get('shuffle_mode'))"; }

timeout 600 python bench.py --model coclr --no-cpu-baseline --no-extra-legs 2>/dev/null | tee gpurun_out/r05_bench_cfg4_coclr.txt | show cfg4

timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29791 tests/bench_rehearse_gpu.py --gpus 2 --steps 5 --warmup 2 --batch 8 --moco-k 2048 --model coclr 2>gpurun_out/r05_rehearsal_coclr.err | tee gpurun_out/r05_rehearsal_n2_coclr.json | show n2-coclr
tail -3 gpurun_out/r05_rehearsal_coclr.err | cut -c1-300
