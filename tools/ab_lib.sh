#!/bin/bash
# A/B runs of bench.py with two builds of the library on one box:  tools/ab_lib.sh <tag> <lib a> <lib b>   (each twice)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  i=0
  for lib in "$@"; do
    i=$((i+1))
    out=gpurun_out/ab_${tag}_${i}_$rep.json
    EFFI_MVS_LIB=$R/$lib timeout -k 10 300 python bench.py --steps 40 --cpu-views 1 --torch-baseline-views 0 --no-whole-forward --no-other-precision > $out 2> ${out%.json}.err || { echo "[$lib] FAILED"; tail -3 ${out%.json}.err; exit 1; }
    python - "$out" "$lib" <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); ss = r.get("single_stream", {}); kb = r["kernel_breakdown_ms"]; po = r.get("parity_vs_oracle", {})
g = lambda *ks: round(sum(v for k, v in kb.items() if any(k.startswith(p) for p in ks)), 4)
print("%-44s %6.1f views/s  %.3f ms  single %.3f ms  differing %s  zr %.4f q %.4f mask %.4f  parity mean %.3e p99 %.3e" % (
    sys.argv[2].split("/")[-1], r["value"], r["ms_per_step"], ss.get("ms_per_view", float("nan")), ss.get("timed_in_flight_views_differing_from_single_stream"),
    g("conv2d_k3x3_nt2_epi1", "conv2d_k3x3_nt4_epi1", "conv2d_k3x3_nt6_epi1"), g("conv2d_k3x3_nt1_epi2", "conv2d_k3x3_nt2_epi2", "conv2d_k3x3_nt3_epi2"),
    g("conv2d_k3k1up"), po.get("worst_depth_mean_norm", float("nan")), po.get("worst_depth_p99_norm", float("nan"))))
PY
  done
done
