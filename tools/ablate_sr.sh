#!/bin/bash
# Ablation builds of the split-resident 3x3 convolution tile (conv2d_x3.hpp, -DEFFI_ABL=<bits>): one extra library per variant under
# gpurun_out/abl/ (built HERE, cross-compiled; they travel with the snapshot), then tools/bench_sr.py once per library on the GPU box.
# Usage (build container):  tools/ablate_sr.sh build "1 2 4 8 16 3 24 31"      (GPU box):  tools/ablate_sr.sh run "0 1 2 ..." [stages]
set -e
cd "$(dirname "$0")/.."
mode=$1; variants=$2; stages=${3:-2}
CS=effi_mvs_plus_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
if [ "$mode" = build ]; then
  mkdir -p abl_libs
  for v in $variants; do
    /opt/rocm/bin/hipcc $FLAGS -DEFFI_ABL=$v -c $CS/conv2d_sr.hip -o /tmp/conv2d_sr_abl$v.o
    objs=$(ls $CS/*.o | grep -v "conv2d_sr.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl_libs/libeffimvs_abl$v.so $objs /tmp/conv2d_sr_abl$v.o
    echo built $v
  done
else
  for v in $variants; do
    if [ "$v" = 0 ]; then lib=""; else lib=$PWD/abl_libs/libeffimvs_abl$v.so; fi
    echo "=== EFFI_ABL=$v"
    EFFI_MVS_LIB=$lib python tools/bench_sr.py --stages $stages --n 40 ${ROWS:+--rows "$ROWS"} | awk '{print $0}'
  done
fi
