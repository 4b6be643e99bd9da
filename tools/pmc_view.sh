#!/bin/bash
# Dynamic instruction mix and issue cycles of every kernel of one view (eager launches, one view at a time):
#   tools/pmc_view.sh <tag>   -> gpurun_out/pmc_<tag>_insts.txt, gpurun_out/pmc_<tag>_cycles.txt
tag=${1:-v}
R=$GRAFT_REPO_ROOT
tools/pmc_insts.sh $tag $R/bench.py --launch eager --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision || exit 1
cd $R
f=$(find gpurun_out/pmc_$tag -name "i*counter_collection.csv" | head -1)
g=$(find gpurun_out/pmc_$tag -name "j*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f > gpurun_out/pmc_${tag}_insts.txt
python tools/pmc_summary.py $g > gpurun_out/pmc_${tag}_cycles.txt
wc -l gpurun_out/pmc_${tag}_insts.txt gpurun_out/pmc_${tag}_cycles.txt
