#!/bin/bash
# sweep of the rolling-window conv3d launch parameters at cfg3 shapes (GPU): EFFI_ROLL_ZT x EFFI_ROLL_MR (x 4-slot variant)
for mr in 2 4; do for zt in 4 6 8 12 16 24 48; do
  echo "base MR=$mr ZT=$zt"; EFFI_ROLL_MR=$mr EFFI_ROLL_ZT=$zt python tools/bench_conv3d.py 2>/dev/null | cut -c1-44
done; done
for cfg in "8 1" "4 2" "8 2"; do set -- $cfg; for zt in 4 6 8 12 16 24 48; do
  echo "v2 NW=$1 MR=$2 ZT=$zt"; EFFI_ROLL_V2=1 EFFI_ROLL_NW=$1 EFFI_ROLL_MR=$2 EFFI_ROLL_ZT=$zt python tools/bench_conv3d.py 2>/dev/null | cut -c1-44
done; done
