#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace of the same bench command,
# per-layer kernel table.  Everything lands under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT=${1:-all}
if [[ $WHAT == all || $WHAT == tests ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -3 gpurun_out/pytest_gpu.log
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>gpurun_out/bench.err
  tail -1 gpurun_out/bench.log
fi
if [[ $WHAT == all || $WHAT == prof ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1)
  find gpurun_out/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -40
  # the raw trace is large; keep only the stats
  find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
fi
if [[ $WHAT == all || $WHAT == layers ]]; then
  timeout 600 python tools/bench_layers.py > gpurun_out/layers.log 2>&1
  cat gpurun_out/layers.log
fi
