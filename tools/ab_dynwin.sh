#!/bin/bash
# same-box A/B of the stage-2/3 warp kernel forms: LDS-window kernel (default) vs gather kernel (EFFI_DYN_WIN=-1); bench line figures
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for v in "" -1; do
  EFFI_DYN_WIN=$v python bench.py --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > gpurun_out/ab_dynwin_${v:-win}_$rep.json 2> gpurun_out/ab_dynwin.err || exit 1
  python - <<PY
import json
r = json.load(open("gpurun_out/ab_dynwin_${v:-win}_$rep.json"))
print("EFFI_DYN_WIN=${v:-unset} rep $rep:", round(r["value"], 1), "views/s in flight;", r["config"].get("single_stream_ms"), "ms single-stream; stages", {k: round(x, 4) for k, x in r["ms_per_cost_volume_stage"].items()})
PY
done
done
