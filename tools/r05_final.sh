#!/bin/bash
# round 5, end: the whole parity tier with its slowest tests, the bench line, smoke, and the N=2 one-GPU rehearsal line
set -u
cd /root/repo; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r05_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_pytest_gpu.txt
grep -a "passed\|failed\|rc=" gpurun_out/r05_pytest_gpu.txt | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_stdout.txt 2> gpurun_out/r05_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r05_bench_stdout.txt | cut -c1-260
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29790 tests/bench_rehearse_gpu.py --gpus 2 --steps 10 --warmup 3 --batch 16 --moco-k 2048 2> gpurun_out/r05_rehearsal_n2.err | tail -1 > gpurun_out/r05_rehearsal_n2.json; cut -c1-200 gpurun_out/r05_rehearsal_n2.json
