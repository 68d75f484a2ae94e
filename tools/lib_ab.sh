# same-box A/B of two builds of the library (coclr_amd/csrc/build/lib_head.so vs lib_new.so), alternating
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "bn or batchnorm" > gpurun_out/lab_k.log 2>&1; tail -2 gpurun_out/lab_k.log
for l in head new; do cp coclr_amd/csrc/build/lib_$l.so coclr_amd/libcoclr_hip.so; echo "== $l"; python tools/bench_layers.py Mixed_4f.b1 Mixed_5c.b1 2>/dev/null | grep "Mixed_"; done
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  cp coclr_amd/csrc/build/lib_head.so coclr_amd/libcoclr_hip.so; $B > gpurun_out/lab_h$i.log 2>/dev/null; val gpurun_out/lab_h$i.log head
  cp coclr_amd/csrc/build/lib_new.so coclr_amd/libcoclr_hip.so; $B > gpurun_out/lab_n$i.log 2>/dev/null; val gpurun_out/lab_n$i.log new
done
