#!/bin/bash
# One kernel-trace pass of the bench command (fp32 storage unless $2 says bf16), aggregated by (kernel, grid).
# Usage: tools/trace_quick.sh <tag> [f32|bf16] [rows]
set -e
tag=${1:-q}
mode=${2:-f32}
rows=${3:-90}
R=/root/repo
OUT=$R/gpurun_out/trace_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-secondary --no-module-api --no-inference --steps 20 --warmup 5 --dtype $mode > $OUT/bench.json 2> $OUT/bench.log
python3 $R/tools/trace_by_grid.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) 25 $rows > $OUT/by_grid_$mode.txt
python3 $R/tools/trace_gaps.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) 25 > $OUT/gaps_$mode.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$mode.csv
rm -rf $OUT/stats
