# same-box A/B of COCLR_GRAPH_QUERY=late (hipGraph replay of Mixed_4b..5c only), alternating
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "graphed" > gpurun_out/late_t.log 2>&1; tail -5 gpurun_out/late_t.log
B="python bench.py --steps 15 --warmup 6 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  $B > gpurun_out/lab_off$i.log 2>/dev/null; val gpurun_out/lab_off$i.log off
  COCLR_GRAPH_QUERY=late $B > gpurun_out/lab_on$i.log 2>gpurun_out/lab_on$i.err; val gpurun_out/lab_on$i.log late
done
COCLR_GRAPH_QUERY=1 $B > gpurun_out/lab_all.log 2>/dev/null; val gpurun_out/lab_all.log all
