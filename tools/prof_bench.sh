#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run; summary CSV copied to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_bench.sh <tag> [bench args...]   (env passes through, e.g. COCLR_OVERLAP_KEYS=0)
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o trace -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-legs "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log | cut -c1-300
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total per step: %.2f ms" % (tot/5/1e6/ (8/5)))   # 3 warmup + 5 timed = 8 steps traced
for r in rows[:28]:
    print("%8.3f ms/step %6d calls  %s" % (float(r["TotalDurationNs"])/8/1e6, int(r["Calls"]), r["Name"][:110]))
PY
t=$(find $OUT -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $t > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt 2>&1
find $OUT -name '*kernel_trace.csv' -size +20M -delete
