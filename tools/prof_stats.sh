#!/bin/bash
# Kernel-trace statistics of a short single-stream graph-replay run:  tools/prof_stats.sh <tag> [grep pattern]
tag=${1:-q}; pat=${2:-.}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision --steps 10 > $R/gpurun_out/prof_$tag.log 2>&1 || exit 1
cd $R
python - "$tag" "$pat" <<'PY'
import csv, re, sys
tag, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f"gpurun_out/prof_{tag}/p_kernel_stats.csv")))
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if re.search(pat, n):
        print("%-70s calls %5s avg %9.1f us  %5.2f %%" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
