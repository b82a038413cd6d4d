#!/bin/bash
# ablation of the TransitionDown forward kernel on a diagnostic build (run on the GPU box from the repo root)
set -e
bash sim2real_lane_segment_amd/csrc/build.sh -DRLN_DIAG > /dev/null
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 6 3; do
  RLN_P1_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1abl_$dbg -o t -- python3 /root/repo/tools/p1_bench.py "$@" > /dev/null 2>&1
  f=$(ls /tmp/p1abl_$dbg/*kernel_stats.csv | head -1)
  echo "dbg=$dbg $(grep p1_fwd_k $f | cut -d, -f1-4 | tr -d '"' | sed 's/void rln:://')"
done
