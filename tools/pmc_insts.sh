#!/bin/bash
# usage: tools/pmc_insts.sh <tag> <python args...>   -- dynamic instruction mix of one command (SQ counters only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_$tag -o i -- python3 "$@" > $R/gpurun_out/pmc_${tag}_i.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_$tag -o j -- python3 "$@" > $R/gpurun_out/pmc_${tag}_j.log 2>&1
