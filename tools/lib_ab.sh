# same-box A/B of two builds of the library (coclr_amd/csrc/build/lib_head.so vs lib_new.so), alternating
mkdir -p gpurun_out
B="python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extra-legs"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  cp coclr_amd/csrc/build/lib_head.so coclr_amd/libcoclr_hip.so; $B > gpurun_out/lab_h$i.log 2>/dev/null; val gpurun_out/lab_h$i.log head
  cp coclr_amd/csrc/build/lib_new.so coclr_amd/libcoclr_hip.so; $B > gpurun_out/lab_n$i.log 2>/dev/null; val gpurun_out/lab_n$i.log new
done
