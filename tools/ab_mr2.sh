#!/bin/bash
# same-box A/B of the rows-per-wave threshold EFFI_MR2_MIN (workgroup count from which the split 3x3 convolutions take 2 rows per wave)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for v in 400 100 200 50 400 100 200 50; do
  EFFI_MR2_MIN=$v python bench.py --steps ${STEPS:-20} --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > gpurun_out/t.json 2> gpurun_out/t.err || { tail -3 gpurun_out/t.err; continue; }
  python - <<PY
import json
r = json.load(open("gpurun_out/t.json"))
print("EFFI_MR2_MIN=$v steps ${STEPS:-20}:", round(r["value"], 1), "views/s in flight;", round(r["config"]["single_stream_ms"], 3), "ms single-stream;", {k: round(x, 3) for k, x in r["ms_per_cost_volume_stage"].items()})
PY
done
