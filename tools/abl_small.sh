#!/bin/bash
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 7; do
  OUT=$R/gpurun_out/abl_small_$d
  mkdir -p $OUT
  RLN_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-secondary --no-module-api --no-inference --steps 4 --warmup 2 > $OUT/bench.json 2> $OUT/bench.log
  echo "dbg=$d" >> $R/gpurun_out/abl_small.txt
  python3 $R/tools/trace_by_grid.py $(ls $OUT/stats/*/*kernel_trace.csv | head -1) 6 200 | grep -E "igemm_k<3, 1, 1, 0, (8|4), 32|splitk" >> $R/gpurun_out/abl_small.txt
  rm -rf $OUT
done
