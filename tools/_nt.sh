cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -k "pool or bn or batchnorm or norm or basic" > gpurun_out/nt_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/nt_pytest.txt
bash tools/ab_bench.sh COCLR_BN_NT_MB "-1 0" 4 --no-self-check
