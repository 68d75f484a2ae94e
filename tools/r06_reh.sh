#!/bin/bash
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1 0 1; do
COCLR_WGRAD_BN=$v timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2979$v tests/bench_rehearse_gpu.py --gpus 2 --steps 10 --warmup 3 --batch 16 --moco-k 2048 2> gpurun_out/reh_$v.err | tail -1 > gpurun_out/reh_$v.json
python -c "
import json; r=json.loads(open('gpurun_out/reh_$v.json').read()); print('WGRAD_BN=$v', r['value'], r['ms_per_step'], r['self_check']['passed'], r['launch_plans']['settle_steps'])"
done
