# XCD-aware tile ids in the direct / temporal Winograd kernels and split counts that are multiples of 8 in
# the weight-gradient kernels (COCLR_XCD_MAP=0: plain ids), same box, alternating
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or wgrad" > gpurun_out/xcd_k.log 2>&1; tail -2 gpurun_out/xcd_k.log
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  COCLR_XCD_MAP=0 $B > gpurun_out/xc_off$i.log 2>/dev/null; val gpurun_out/xc_off$i.log plain
  $B > gpurun_out/xc_on$i.log 2>/dev/null; val gpurun_out/xc_on$i.log xcd
done
