#!/bin/bash
# Kernel-trace statistics of the training step (tools/bench_train.py, HIP path only):  tools/prof_train.sh <tag>
tag=${1:-t}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && export BENCH_TRAIN_HIP_ONLY=1 && export BENCH_TRAIN_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$tag -o p -- python3 $R/tools/bench_train.py > $R/gpurun_out/prof_train_$tag.log 2>&1 || { tail -20 $R/gpurun_out/prof_train_$tag.log; exit 1; }
cd $R
grep "HIP training step" gpurun_out/prof_train_$tag.log
python3 - "$tag" <<'PY' > gpurun_out/prof_train_$1.txt
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/prof_train_{tag}/p_kernel_stats.csv")))
steps = 6                     # bench_train: 1 warm-up + 5 timed steps
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / steps / 1e6:.2f} ms, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
for r in rows[:70]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-110s calls/step %7.1f avg %8.1f us  %6.2f ms/step %5.2f %%" % (n[:110], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                                   float(r["TotalDurationNs"]) / steps / 1e6, float(r["Percentage"])))
PY
rm -f gpurun_out/prof_train_$tag/*kernel_trace.csv gpurun_out/prof_train_$tag/*.db
head -45 gpurun_out/prof_train_$tag.txt
