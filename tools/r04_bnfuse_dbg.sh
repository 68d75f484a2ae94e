#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/bnfuse
timeout 1500 python -m pytest tests/test_gpu_gradients.py tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_next.py::test_gradients_in_ddp_buckets_bit_identical_and_deterministic tests/test_gpu_multirank.py tests/test_gpu_bench_rehearsal.py -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|ProcessGroupNCCL" > gpurun_out/bnfuse/pytest_model_full.txt
grep -n "FAILED\|passed\|failed" gpurun_out/bnfuse/pytest_model_full.txt | tail
