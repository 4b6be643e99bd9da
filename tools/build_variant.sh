#!/bin/bash
# A/B library with extra -D flags on the split-resident convolution unit only (conv2d_sr.hip); the other objects are the in-tree ones.
# Usage: tools/build_variant.sh <name> "<flags>"   ->   abl_libs/libeffimvs_<name>.so   (select with EFFI_MVS_LIB on the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
CS=effi_mvs_plus_amd/csrc
mkdir -p abl_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function $flags -c $CS/conv2d_sr.hip -o /tmp/conv2d_sr_$name.o
objs=$(ls $CS/*.o | grep -v "conv2d_sr.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl_libs/libeffimvs_$name.so $objs /tmp/conv2d_sr_$name.o
echo built abl_libs/libeffimvs_$name.so
