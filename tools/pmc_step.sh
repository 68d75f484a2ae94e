#!/bin/bash
# PMC passes over WHOLE training steps of bench.py (one stream, no graphs / plans: the counters serialise the
# dispatches anyway), folded per kernel over ONE steady-state step by tools/pmc_step_table.py.
# usage: tools/pmc_step.sh <tag>     -> gpurun_out/<tag>_pmc_step_table.txt, gpurun_out/<tag>_step_pmc.json
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0 COCLR_PLAN=0 timeout 900 rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/p$i -o pmc -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-self-check > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python $ROOT/tools/pmc_step_table.py $OUT $ROOT/gpurun_out/${TAG}_pmc_step_table.txt $ROOT/gpurun_out/${TAG}_step_pmc.json
find $OUT -name '*.csv' -size +30M -delete
