#!/bin/bash
# usage (GPU box): tools/graph_ab.sh <tag> "<ENV=.. for A>" "<ENV=.. for B>"  -- rocprofv3 kernel trace of linear-graph replays (one view
# after the other) under two environments; timeline of one replay each -> gpurun_out/<tag>_{a,b}_timeline.txt and a per-kernel diff
tag=$1; ea=$2; eb=$3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  e=$ea; [ $v = b ] && e=$eb
  env $e rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_prof_$v -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision --steps 10 > $O/${tag}_prof_$v.log 2>&1 || exit 1
  python3 $R/tools/graph_timeline.py $(find $O/${tag}_prof_$v -name "p_kernel_trace.csv" | head -1) 14 > $O/${tag}_${v}_timeline.txt
  rm -rf $O/${tag}_prof_$v
  tail -1 $O/${tag}_${v}_timeline.txt
done
python3 - <<PY
import collections
def load(p):
    d=collections.OrderedDict()
    for l in open(p):
        if " us  +" not in l: continue
        parts=l.split()
        dur=float(parts[2].lstrip("+")); name=parts[-1]
        d.setdefault(name,[0,0.0]); d[name][0]+=1; d[name][1]+=dur
    return d
a=load("$O/${tag}_a_timeline.txt"); b=load("$O/${tag}_b_timeline.txt")
ks=sorted(set(a)|set(b), key=lambda k:-max(a.get(k,[0,0])[1], b.get(k,[0,0])[1]))
for k in ks[:40]:
    x=a.get(k,[0,0.0]); y=b.get(k,[0,0.0])
    print(f"{k:62s} {x[0]:3d} {x[1]:8.1f} | {y[0]:3d} {y[1]:8.1f}  {y[1]-x[1]:+7.1f}")
PY
