#!/bin/bash
# last call of the round: full parity tier and the bench line on HEAD
set -u
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest_gpu.txt
grep "passed\|failed" gpurun_out/r04_pytest_gpu.txt | tail -2; tail -1 gpurun_out/r04_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_stdout.txt 2>gpurun_out/r04_bench.err; tail -1 gpurun_out/r04_bench_stdout.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
