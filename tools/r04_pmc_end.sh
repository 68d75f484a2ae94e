#!/bin/bash
# end-of-round PMC table over the long kernels (separate --pmc passes) and the three-stream kernel trace of the bench
set -u
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/pmc_layers.sh Conv_1a Conv_2c 3c.b1 4f.b1 > gpurun_out/r04_pmc_layers_end.txt 2>&1
tail -5 gpurun_out/r04_pmc_layers_end.txt | cut -c1-150
rm -rf gpurun_out/pmc
bash tools/prof_bench.sh r04_bench > gpurun_out/r04_prof_bench.log 2>&1; tail -3 gpurun_out/r04_bench_timeline.txt
rm -rf gpurun_out/prof_r04_bench
