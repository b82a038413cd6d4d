#!/bin/bash
# Round profile set (GPU box): kernel-trace statistics, FETCH/WRITE traffic passes and SQ counter passes of the bench command
# for both storage modes.  Counter passes never combine --pmc with a trace option (gpurun refuses that mix); the program
# itself follows "--" (no env / shell hop).   Usage: tools/profile_round.sh r03
set -e
tag=${1:-r03}
R=/root/repo
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-profile --no-secondary --no-module-api"
for mode in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$mode -- $BENCH --steps 20 --warmup 5 --dtype $mode > $OUT/stats_$mode.json 2> $OUT/stats_$mode.log
  cp $(ls $OUT/stats_$mode/*/*kernel_stats.csv | head -1) $OUT/${tag}_kernel_stats_$mode.csv
  python3 $R/tools/trace_by_grid.py $(ls $OUT/stats_$mode/*/*kernel_trace.csv | head -1) 25 60 > $OUT/${tag}_kernel_trace_by_grid_$mode.txt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$mode -- $BENCH --steps 2 --warmup 1 --dtype $mode > /dev/null 2> $OUT/fetch_$mode.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$mode -- $BENCH --steps 2 --warmup 1 --dtype $mode > /dev/null 2> $OUT/write_$mode.log
  python3 $R/tools/pmc_traffic.py $OUT/fetch_$mode $OUT/write_$mode $OUT/${tag}_pmc_traffic_$mode.json > $OUT/traffic_$mode.txt
  A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
  B="SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"
  rocprofv3 --pmc $A --output-format csv -d $OUT/sqa_$mode -- $BENCH --steps 2 --warmup 1 --dtype $mode > /dev/null 2> $OUT/sqa_$mode.log
  rocprofv3 --pmc $B --output-format csv -d $OUT/sqb_$mode -- $BENCH --steps 2 --warmup 1 --dtype $mode > /dev/null 2> $OUT/sqb_$mode.log
  python3 $R/tools/pmc_sq.py --json $OUT/${tag}_pmc_sq_$mode.json $OUT/sqa_$mode $OUT/sqb_$mode > $OUT/sq_$mode.txt
  rm -rf $OUT/stats_$mode $OUT/fetch_$mode $OUT/write_$mode $OUT/sqa_$mode $OUT/sqb_$mode
  echo "mode $mode done"
done
