#!/bin/bash
# A/B runs of the rows-per-wave rule of the split-precision 3x3 convolutions (EFFI_MR4_MIN / EFFI_MR4_NT2_MAX / EFFI_MR2_MIN,
# csrc/conv2d.hip) on ONE box: headline (3 views in flight) and single-stream ms per view for each setting.
#   tools/sweep_tiles.sh [workload]      outputs under gpurun_out/tiles_*.json, summary on stdout
wl=${1:-cfg3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
XARGS=""
run() {
  tag=$1; shift
  env "$@" python bench.py $XARGS --workload $wl --steps 40 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision \
      > $O/tiles_${wl}_$tag.json 2> $O/tiles_${wl}_$tag.err || { echo "$tag FAILED"; tail -3 $O/tiles_${wl}_$tag.err; return 1; }
  python - <<PY
import json
r = json.load(open("$O/tiles_${wl}_$tag.json"))
ss = r.get("single_stream", {})
print("%-28s in-flight %.1f views/s (%.3f ms)  single %.3f ms  differing %s  stages %s" % (
    "$tag", r["value"], r["ms_per_step"], ss.get("ms_per_view", float("nan")),
    ss.get("timed_in_flight_views_differing_from_single_stream"),
    {k: round(v, 3) for k, v in r.get("ms_per_cost_volume_stage", {}).items()}))
PY
}
run default EFFI_DUMMY=0 &&
run mr4_always EFFI_MR4_MIN=1 EFFI_MR4_NT2_MAX=1000000000 &&
run mr4_from_100 EFFI_MR4_MIN=100 EFFI_MR4_NT2_MAX=1000000000 &&
run mr4_nt2_too EFFI_MR4_NT2_MAX=1000000000 &&
run mr4_from_100_nt2_rule EFFI_MR4_MIN=100 &&
run mr2_from_100 EFFI_MR2_MIN=100 &&
run default_again EFFI_DUMMY=0 &&
XARGS="--in-flight 2" run in_flight_2 EFFI_DUMMY=0 &&
XARGS="--in-flight 4" run in_flight_4 EFFI_DUMMY=0 &&
XARGS="--in-flight 6" run in_flight_6 EFFI_DUMMY=0
