mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "pool" > gpurun_out/pool_k.log 2>&1; tail -3 gpurun_out/pool_k.log
echo "== generic (one round trip per class)"; COCLR_POOL_PRELOAD=0 python tools/bench_layers.py pool Pool bn2 2>/dev/null | grep -v "^$" | grep -v "^layer\|^bn unit"
echo "== preload"; python tools/bench_layers.py pool Pool bn2 2>/dev/null | grep -v "^$" | grep -v "^layer\|^bn unit"
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  COCLR_POOL_PRELOAD=0 $B > gpurun_out/pl_off$i.log 2>/dev/null; val gpurun_out/pl_off$i.log generic
  $B > gpurun_out/pl_on$i.log 2>/dev/null; val gpurun_out/pl_on$i.log preload
done
