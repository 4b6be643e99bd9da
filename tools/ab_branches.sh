#!/bin/bash
# same-box A/B: views in flight on linear graphs vs graphs captured with the pass's side stream; 20 and 60 timed steps
R=$GRAFT_REPO_ROOT; cd $R
for steps in 20 60; do
for br in 0 1 0 1; do
  python bench.py --steps $steps --graph-branches $br --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > gpurun_out/ab_br.json 2> gpurun_out/ab_br.err || { tail -5 gpurun_out/ab_br.err; exit 1; }
  python - <<PY
import json
r = json.load(open("gpurun_out/ab_br.json"))
print("steps $steps branches $br:", round(r["value"], 1), "views/s in flight;", r["config"].get("single_stream_ms"), "ms single-stream;", r["config"]["launch"][:90])
PY
done
done
