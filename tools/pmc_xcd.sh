#!/bin/bash
# HBM fetch bytes of the temporal Winograd kernel on Conv_2c.conv2 with plain and XCD-aware tile ids
# (one --pmc pass per counter and mode; kernel trace only beside it)
set -u
export TMPDIR=/tmp
cd /tmp
for mode in 0 1; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_xcd$mode
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    COCLR_XCD_MAP=$mode timeout 200 rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_layers.py Conv_2c.conv2 > $OUT/p$i.log 2>&1
  done
  echo "== COCLR_XCD_MAP=$mode"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT 2>&1 | grep -A3 "wino_t\|wgrad2" | head -12
done
