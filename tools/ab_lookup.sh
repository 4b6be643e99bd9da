#!/bin/bash
# same-box A/B: generated-input kernel with refined-reciprocal look-up divisions (in-tree) vs IEEE divisions (abl_libs/libeffimvs_exactlk.so,
# tools/build_variant.sh exactlk "-DEFFI_EXACT_LOOKUP"); eager per-kernel times of the bench's discovery pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do
for lib in "" abl_libs/libeffimvs_exactlk.so; do
  EFFI_MVS_LIB=${lib:+$R/$lib} python bench.py --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > gpurun_out/ab_lk.json 2> gpurun_out/ab_lk.err || { tail -3 gpurun_out/ab_lk.err; exit 1; }
  python - "${lib:-in-tree}" <<'PY'
import json, sys
r = json.load(open("gpurun_out/ab_lk.json")); kb = r["kernel_breakdown_ms"]
print("%-36s %6.1f views/s  single %.3f ms  encgen nt1/nt2/nt3 %.4f %.4f %.4f ms per view  parity mean %.3e" % (
    sys.argv[1], r["value"], r["config"]["single_stream_ms"], kb.get("encgen_pair_nt1", 0), kb.get("encgen_pair_nt2", 0), kb.get("encgen_pair_nt3", 0),
    r.get("parity_vs_oracle", {}).get("worst_depth_mean_norm", float("nan"))))
PY
done
done
