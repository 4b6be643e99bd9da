#!/bin/bash
# same-box sweep: views in flight x graph form (linear / with the pass's side stream), 20 and 60 timed steps
R=$GRAFT_REPO_ROOT; cd $R
for steps in 20 60; do
for cfg in "1 3" "0 3" "0 4" "1 4" "0 5" "0 6" "0 4" "1 3"; do
  set -- $cfg
  python bench.py --steps $steps --graph-branches $1 --in-flight $2 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > gpurun_out/ab_if.json 2> gpurun_out/ab_if.err || { tail -5 gpurun_out/ab_if.err; exit 1; }
  python - <<PY
import json
r = json.load(open("gpurun_out/ab_if.json"))
print("steps $steps branches $1 in-flight $2:", round(r["value"], 1), "views/s;", round(r["config"].get("single_stream_ms"), 3), "ms single-stream")
PY
done
done
