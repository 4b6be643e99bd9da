#!/bin/bash
# round-3 acceptance bundle on the GPU box: SR tests, cfg5 (small set: 1 rank, 2-rank gloo rehearsal), default-ish bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_sr.py -x -q 2>&1 | tail -2
python bench.py --workload cfg5 --cfg5-scans 2 --cfg5-images 12 > $O/cfg5_small.json 2> $O/cfg5_small.err; tail -3 $O/cfg5_small.err; cat $O/cfg5_small.json
python bench.py --workload cfg5 --cfg5-scans 2 --cfg5-images 12 --gpus 2 --backend gloo > $O/cfg5_2rank_gloo.json 2> $O/cfg5_2rank.err; tail -3 $O/cfg5_2rank.err; cat $O/cfg5_2rank_gloo.json
python bench.py --steps 30 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value",round(d["value"],1),"single",round(d["single_stream"]["ms_per_view"],3), {k:round(v,3) for k,v in d["ms_per_cost_volume_stage"].items()}, "differing", d["single_stream"]["timed_in_flight_views_differing_from_single_stream"])
PY
