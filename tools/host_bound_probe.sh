# Is the host on the critical path?  Same box, alternating: extra host microseconds per launch.
mkdir -p gpurun_out
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for d in 0 5 10 20 0; do
  COCLR_HOST_DELAY_US=$d $B > gpurun_out/hb_$d.log 2>/dev/null; val gpurun_out/hb_$d.log delay_us=$d
done
