#!/bin/bash
# Ablation of the dense3 kernels in a diagnostic build on the GPU box: rebuilds dense3/net with -DRLN_DIAG and times the
# bench's kernel classes with RLN_D3_DBG = 0 / 1 (no loads) / 2 (no commit) / 4 (no MFMA) / combinations.
cd sim2real_lane_segment_amd/csrc && rm -f build/dense3.o build/net.o && bash build.sh -DRLN_DIAG > /dev/null 2>&1; cd ../..
for d in "$@"; do
  RLN_D3_DBG=$d python bench.py --no-cpu-baseline --no-module-api --steps 4 --warmup 2 --fwd-arith f16x2 --bwd-arith bf16x2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kc={k['name']:k['ms_per_step'] for k in d['kernel_classes']}
print('dbg=$d', 'd3_wgrad', kc.get('dense3_wgrad'), 'd3_fwd', kc.get('dense3_fwd'), 'd3_pull', kc.get('dense3_dgrad_pull'))"
done
