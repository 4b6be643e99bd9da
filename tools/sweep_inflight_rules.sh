#!/bin/bash
# tile rules that favour THROUGHPUT (views in flight: another view's kernels fill what a launch leaves idle, so fewer, larger workgroups
# and less weight traffic win) against the defaults, which favour one view's latency.  Same box, 40 timed steps, in-flight value and
# single-stream ms (both captured under the same options here).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
run() {
  tag=$1; shift
  env "$@" python bench.py --steps 40 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > gpurun_out/t.json 2> gpurun_out/t.err || { echo "$tag FAILED"; tail -2 gpurun_out/t.err; return; }
  python - "$tag" <<'PY'
import json, sys
r = json.load(open("gpurun_out/t.json"))
print("%-34s in flight %6.1f views/s   single %.3f ms   stages %s" % (sys.argv[1], r["value"], r["config"]["single_stream_ms"], {k: round(v, 3) for k, v in r["ms_per_cost_volume_stage"].items()}))
PY
}
run default EFFI_DUMMY=0
run mr2_100 EFFI_MR2_MIN=100
run mr2_100_mr4_200 EFFI_MR2_MIN=100 EFFI_MR4_MIN=200
run mr2_100_mr4_100 EFFI_MR2_MIN=100 EFFI_MR4_MIN=100
run mr2_100_srw8 EFFI_MR2_MIN=100 EFFI_SR_WAVES=8
run mr2_50 EFFI_MR2_MIN=50
run mr4_200 EFFI_MR4_MIN=200
run default_again EFFI_DUMMY=0
run mr2_100_again EFFI_MR2_MIN=100
