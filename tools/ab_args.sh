#!/bin/bash
# A/B runs of bench.py under different ARGUMENTS on one box:  tools/ab_args.sh <tag> "<args a>" "<args b>" ...  (each twice)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  i=0
  for a in "$@"; do
    i=$((i+1))
    out=gpurun_out/ab_${tag}_${i}_$rep.json
    timeout -k 10 240 python bench.py $a --steps 40 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > $out 2> ${out%.json}.err || { echo "[$a] FAILED"; tail -3 ${out%.json}.err; exit 1; }
    python - "$out" "$a" <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); ss = r.get("single_stream", {})
print("%-40s %6.1f views/s  %.3f ms per view  single %.3f ms  differing %s" % (
    sys.argv[2], r["value"], r["ms_per_step"], ss.get("ms_per_view", float("nan")), ss.get("timed_in_flight_views_differing_from_single_stream")))
PY
  done
done
