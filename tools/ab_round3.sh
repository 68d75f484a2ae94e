mkdir -p gpurun_out
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
$B > gpurun_out/ab_base.log 2>/dev/null; val gpurun_out/ab_base.log base
COCLR_SIDE_WINDOW=0 $B > gpurun_out/ab_nowin.log 2>/dev/null; val gpurun_out/ab_nowin.log side_window_off
COCLR_LANES=small $B > gpurun_out/ab_lsmall.log 2>/dev/null; val gpurun_out/ab_lsmall.log lanes_small
COCLR_LANES=graph $B > gpurun_out/ab_lgraph.log 2>/dev/null; val gpurun_out/ab_lgraph.log lanes_graph
COCLR_LANES=small+graph $B > gpurun_out/ab_lsg.log 2>/dev/null; val gpurun_out/ab_lsg.log lanes_small+graph
$B > gpurun_out/ab_base2.log 2>/dev/null; val gpurun_out/ab_base2.log base_again
timeout 300 python tools/find_copies.py > gpurun_out/find_copies.txt 2>&1; tail -80 gpurun_out/find_copies.txt
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -x -q > gpurun_out/pytest_r03b.log 2>&1; tail -5 gpurun_out/pytest_r03b.log
