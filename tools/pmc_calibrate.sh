#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/microbench/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
R=$GRAFT_REPO_ROOT
cd $R/tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_calib -o c -- /tmp/fetch_calib > $R/gpurun_out/fetch_calib.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, collections, glob, re
true = {}
for line in open("gpurun_out/fetch_calib.log"):
    if line.startswith("true_bytes"):
        t = line.split()
        true = {t[i]: int(t[i + 1]) for i in range(1, len(t), 2)}
f = glob.glob("gpurun_out/fetch_calib/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    n = r["Kernel_Name"]
    k = "seg" + re.search(r"seg_kernel<(\d+)", n).group(1) if "seg_kernel<" in n else n.split("_kernel")[0].split()[-1]
    acc[k].append(float(r["Counter_Value"]) * 1024.0)
out = open("gpurun_out/fetch_calib.txt", "w")
for k, v in acc.items():
    v = sorted(v)[len(v) // 2]
    line = f"{k:10s} true {true.get(k, 0) / 1e6:9.1f} MB   FETCH_SIZE {v / 1e6:9.1f} MB   factor (true / counted) {true.get(k, 0) / v:.3f}"
    print(line); out.write(line + "\n")
PY
