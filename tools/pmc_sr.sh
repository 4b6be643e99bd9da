#!/bin/bash
# usage (on the GPU box): tools/pmc_sr.sh <tag> [bench_sr.py args]   -- SQ counters of the per-layer benchmark (tools/bench_sr.py), three
# rocprofv3 --pmc passes (instruction mix; issue cycles; waits + LDS) -> gpurun_out/pmc_<tag>_{insts,cycles,waits}.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_$tag -o i -- python3 $R/tools/bench_sr.py "$@" > $R/gpurun_out/pmc_${tag}_i.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_$tag -o j -- python3 $R/tools/bench_sr.py "$@" > $R/gpurun_out/pmc_${tag}_j.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_$tag -o k -- python3 $R/tools/bench_sr.py "$@" > $R/gpurun_out/pmc_${tag}_k.log 2>&1
cd $R
for p in i:insts j:cycles k:waits; do
  f=$(find gpurun_out/pmc_$tag -name "${p%%:*}*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f conv2d_k3 > gpurun_out/pmc_${tag}_${p##*:}.txt
done
rm -rf gpurun_out/pmc_$tag
wc -l gpurun_out/pmc_${tag}_*.txt
