#!/bin/bash
# HBM traffic of every kernel of the bench workload: two separate counter passes (no tracing domains), then the summary
# that bench.py reads (profiles/pmc_traffic_<workload>.json).  usage (on the GPU box): tools/pmc_traffic.sh cfg3
wl=${1:-cfg3}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
args="$R/bench.py --workload $wl --in-flight 1 --steps 10 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_traffic -o fetch -- python3 $args > $R/gpurun_out/pmc_traffic_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_traffic -o write -- python3 $args > $R/gpurun_out/pmc_traffic_write.log 2>&1 &&
cd $R && python3 tools/pmc_traffic.py gpurun_out/pmc_traffic/fetch_counter_collection.csv gpurun_out/pmc_traffic/write_counter_collection.csv gpurun_out/pmc_traffic_$wl.json
