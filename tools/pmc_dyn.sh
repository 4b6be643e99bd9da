#!/bin/bash
# SQ counters of the stage-2/3 warp kernels run stand-alone by tools/probe_dyn_windows.py (three rocprofv3 --pmc passes):
#   tools/pmc_dyn.sh <tag>   -> gpurun_out/pmc_<tag>_{insts,cycles,waits}.txt
tag=${1:-dyn}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PROBE_SKIP_BOXES=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_$tag -o i -- python3 $R/tools/probe_dyn_windows.py cfg3 > $R/gpurun_out/pmc_${tag}_i.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_$tag -o j -- python3 $R/tools/probe_dyn_windows.py cfg3 > $R/gpurun_out/pmc_${tag}_j.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_$tag -o k -- python3 $R/tools/probe_dyn_windows.py cfg3 > $R/gpurun_out/pmc_${tag}_k.log 2>&1
cd $R
for p in i:insts j:cycles k:waits; do
  f=$(find gpurun_out/pmc_$tag -name "${p%%:*}*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f warpcorr_dyn > gpurun_out/pmc_${tag}_${p##*:}.txt
done
rm -rf gpurun_out/pmc_$tag
cat gpurun_out/pmc_${tag}_*.txt
