#!/bin/bash
# Run tools/bench_sr.py once per A/B library (GPU box).  Usage: tools/ab_libs.sh "default old late ..." [stages] [extra bench_sr args]
cd "$(dirname "$0")/.."
for v in $1; do
  if [ "$v" = default ]; then lib=""; else lib=$PWD/abl_libs/libeffimvs_$v.so; fi
  echo "=== $v"
  EFFI_MVS_LIB=$lib python tools/bench_sr.py --stages ${2:-0,1,2} --n 40 $3 2>&1 | grep -v amdgpu.ids
done
