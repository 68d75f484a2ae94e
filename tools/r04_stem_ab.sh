cd /root/repo
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "stem or gather or n_index" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -2
bash tools/lib_ab_layers.sh
