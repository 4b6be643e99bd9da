#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile): one line per kernel with VGPRs, AGPRs,
scratch, occupancy and LDS.  Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt; kernel_resources.py res.txt"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
blocks = t.split("Function Name: ")[1:]
names = [b.split("\n")[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n") if names else []
keys = {"vgpr": r"VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
        "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)", "sgpr": r"SGPRs: (\d+)"}
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b, d in zip(blocks, dem):
    d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
    if flt and flt not in d:
        continue
    vals = {k: int(re.search(p, b).group(1)) for k, p in keys.items()}
    print(f"{d:78s} " + " ".join(f"{k} {v:>5d}" for k, v in vals.items()))
