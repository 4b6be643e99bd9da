#!/bin/bash
# Everything the round's profiles/ directory is built from, in one call on the GPU box:  tools/measure_round.sh <tag>
#   bench (default run, with the CPU baseline), PMC traffic passes, kernel-trace stats of the graph replay and of a serial
#   eager run bracketing one key, the graph timeline, the other workloads, the fp32 mode.  Outputs under gpurun_out/<tag>_*.
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
# PMC traffic first: bench.py's roofline.traffic / traffic_per_view read profiles/pmc_traffic_<workload>.json of THIS build
tools/pmc_traffic.sh cfg3 > $O/${tag}_pmc_traffic_summary.txt 2>&1 || exit 1
cp $O/pmc_traffic_cfg3.json $R/profiles/pmc_traffic_cfg3.json
echo "pmc done"
python bench.py > $O/${tag}_bench_cfg3.json 2> $O/${tag}_bench.err || exit 1
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_graph -o p -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision --steps 10 > $O/${tag}_prof_graph.log 2>&1 || exit 1
EFFI_MVS_BRANCHES=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_serial -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision --steps 20 > $O/${tag}_bench_cfg3_eager_serial.json 2> $O/${tag}_prof_serial.log || exit 1
echo "traces done"
cd $R
python tools/graph_timeline.py $O/${tag}_prof_graph/p_kernel_trace.csv 14 > $O/${tag}_graph_timeline.txt
tail -1 $O/${tag}_graph_timeline.txt
for wl in cfg2 cfg3b cfg4; do
  python bench.py --workload $wl --no-cpu-baseline --torch-baseline-views 2 > $O/${tag}_bench_$wl.json 2>> $O/${tag}_bench.err
  echo "$wl done"
done
python bench.py --precision fp32 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > $O/${tag}_bench_cfg3_fp32.json 2>> $O/${tag}_bench.err
python bench.py --precision bf16 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > $O/${tag}_bench_cfg3_bf16.json 2>> $O/${tag}_bench.err
python bench.py --gpus 2 --backend gloo --workload cfg2 --steps 10 --no-cpu-baseline --torch-baseline-views 0 --no-whole-forward --no-other-precision > $O/${tag}_bench_2rank_gloo_rehearsal_cfg2.json 2>> $O/${tag}_bench.err
python bench.py --fusion-torch-baseline --no-cpu-baseline --torch-baseline-views 0 --no-other-precision --steps 5 > $O/${tag}_bench_cfg3_fusion_torch_baseline.json 2>> $O/${tag}_bench.err
bash tools/pmc_calibrate.sh > $O/${tag}_fetch_calibration.log 2>&1
python - <<PY
import json
for wl in ("cfg3", "cfg2", "cfg3b", "cfg4", "cfg3_fp32", "cfg3_bf16", "cfg3_eager_serial"):
    try:
        r = json.load(open("$O/${tag}_bench_%s.json" % wl))
        print(wl, round(r["value"], 1), r["unit"], round(r["ms_per_step"], 3), "ms", r.get("roofline", {}).get("kernel"), r.get("roofline", {}).get("frac"))
    except Exception as e:
        print(wl, "unreadable:", e)
PY
