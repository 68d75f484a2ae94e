mkdir -p gpurun_out
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  $B > gpurun_out/g_on$i.log 2>/dev/null; val gpurun_out/g_on$i.log key_graph
  COCLR_GRAPHS=0 $B > gpurun_out/g_off$i.log 2>/dev/null; val gpurun_out/g_off$i.log key_eager
done
COCLR_GRAPHS=0 COCLR_OVERLAP_KEYS=0 $B > gpurun_out/g_ser.log 2>/dev/null; val gpurun_out/g_ser.log key_eager_same_stream
