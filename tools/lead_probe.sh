#!/bin/bash
# How far ahead of the GPU is the host?  HIP runtime trace + kernel trace of a short bench run, joined on
# the correlation id: lead = kernel start - end of the hipLaunchKernel / hipGraphLaunch call that queued it.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lead
rm -rf $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace -f csv -d $OUT -o trace -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/lead.log 2>&1)
tail -1 $GRAFT_REPO_ROOT/gpurun_out/lead.log | cut -c1-200
ls -la $OUT
python $GRAFT_REPO_ROOT/tools/lead_analyse.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/lead_summary.txt 2>&1
head -80 $GRAFT_REPO_ROOT/gpurun_out/lead_summary.txt
find $OUT -name '*.csv' -size +8M -delete
