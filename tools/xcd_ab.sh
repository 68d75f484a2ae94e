# XCD-aware tile ids (direct / temporal Winograd kernels; tile-fastest (split, tile) ids in the weight-gradient
# kernels) against plain ids (COCLR_XCD_MAP=0), same box, alternating
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or wgrad" > gpurun_out/xcd_k.log 2>&1; tail -1 gpurun_out/xcd_k.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -k "adjoint and s3d" > gpurun_out/xcd_f.log 2>&1; tail -1 gpurun_out/xcd_f.log
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  COCLR_XCD_MAP=0 $B > gpurun_out/xc_off$i.log 2>/dev/null; val gpurun_out/xc_off$i.log plain
  $B > gpurun_out/xc_on$i.log 2>/dev/null; val gpurun_out/xc_on$i.log xcd
done
