# same-box A/B of two builds of the library (coclr_amd/csrc/build/lib_head.so vs lib_new.so), alternating.
# usage: tools/lib_ab.sh "<pytest -k expression>" "<bench_layers.py filters>"
mkdir -p gpurun_out
K=${1:-pool}; L=${2:-pool Pool}
cp coclr_amd/csrc/build/lib_new.so coclr_amd/libcoclr_hip.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "$K" > gpurun_out/lab_k.log 2>&1; tail -2 gpurun_out/lab_k.log
for l in head new; do cp coclr_amd/csrc/build/lib_$l.so coclr_amd/libcoclr_hip.so; echo "== $l"; python tools/bench_layers.py $L 2>/dev/null | grep -v "^$\|^layer\|^bn unit\|^pooled"; done
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  cp coclr_amd/csrc/build/lib_head.so coclr_amd/libcoclr_hip.so; $B > gpurun_out/lab_h$i.log 2>/dev/null; val gpurun_out/lab_h$i.log head
  cp coclr_amd/csrc/build/lib_new.so coclr_amd/libcoclr_hip.so; $B > gpurun_out/lab_n$i.log 2>/dev/null; val gpurun_out/lab_n$i.log new
done
