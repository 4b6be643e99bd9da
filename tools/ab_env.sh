#!/bin/bash
# A/B runs of bench.py under environment settings, on one box:  tools/ab_env.sh <tag> "VAR=a" "VAR=b" ...  (each setting runs twice)
#   prints: setting, views/s in flight, ms per view in flight / single-stream, differing views, the warp kernels' ms per view
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    out=gpurun_out/ab_${tag}_${i}_$rep.json
    env $setting python bench.py --steps 40 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > $out 2> ${out%.json}.err || { echo "$setting FAILED"; tail -3 ${out%.json}.err; exit 1; }
    python - "$out" "$setting" <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); ss = r["single_stream"]; kb = r["kernel_breakdown_ms"]
print("%-34s %6.1f views/s  %.3f ms in flight  %.3f ms single  differing %s  %s" % (
    sys.argv[2], r["value"], r["ms_per_step"], ss["ms_per_view"], ss["timed_in_flight_views_differing_from_single_stream"],
    {k: v for k, v in kb.items() if "warpcorr" in k}))
PY
  done
done
