# same-box A/B of the fused pool + BatchNorm backward (COCLR_POOLED_BACKWARD), alternating
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "pool or batchnorm or bn" > gpurun_out/pooled_k.log 2>&1; tail -5 gpurun_out/pooled_k.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py -x -q > gpurun_out/pooled_m.log 2>&1; tail -5 gpurun_out/pooled_m.log
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  COCLR_POOLED_BACKWARD=0 $B > gpurun_out/pab_off$i.log 2>/dev/null; val gpurun_out/pab_off$i.log off
  $B > gpurun_out/pab_on$i.log 2>/dev/null; val gpurun_out/pab_on$i.log on
done
