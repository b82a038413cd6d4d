#!/bin/bash
# SQ counter passes of the bench command (GPU box).  Usage: tools/pmc_sq_run.sh <tag> [bench flags...]
# Two --pmc passes (8 SQ slots each), kernel-trace/stats NOT combined with them (gpurun refuses that mix).
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
tag=$1; shift
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"
rocprofv3 --pmc $A --output-format csv -d $R/gpurun_out/sq_${tag}_a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-secondary --no-module-api "$@" > /dev/null 2> $R/gpurun_out/sq_${tag}_a.log
rocprofv3 --pmc $B --output-format csv -d $R/gpurun_out/sq_${tag}_b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-secondary --no-module-api "$@" > /dev/null 2> $R/gpurun_out/sq_${tag}_b.log
python3 $R/tools/pmc_sq.py $R/gpurun_out/sq_${tag}_a $R/gpurun_out/sq_${tag}_b > $R/gpurun_out/sq_${tag}.txt
