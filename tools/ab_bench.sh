#!/bin/bash
# A/B of two builds of librln.so on ONE box (clock and box-to-box spread make separate gpurun calls incomparable):
# usage: tools/ab_bench.sh <a.so> <b.so> [rounds]   -- alternates the two libraries, prints ms/step of every run
R=/root/repo
A=$1; B=$2; N=${3:-3}
cp $R/sim2real_lane_segment_amd/csrc/librln.so /tmp/librln.keep.so
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then cp $A $R/sim2real_lane_segment_amd/csrc/librln.so; else cp $B $R/sim2real_lane_segment_amd/csrc/librln.so; fi
    python3 $R/bench.py --no-cpu-baseline --no-inference --no-module-api --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['storage_bf16']['ms_per_step'])"
  done
done
cp /tmp/librln.keep.so $R/sim2real_lane_segment_amd/csrc/librln.so
