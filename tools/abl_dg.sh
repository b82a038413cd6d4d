# diag build: level-0 data-gradient kernel with parts removed (timing only)
for v in 0 1 2 4 7; do
  echo "== RLN_DG_ABL=$v (1 no S/G loads, 2 no G stores, 4 no MFMA)"
  RLN_DBG=64 RLN_DG_ABL=$v timeout -k 10 150 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={x['name']:x for x in d['kernel_classes']}
print('step',d['ms_per_step'],'dgrad',k['dense_conv3x3_dgrad']['ms_per_step'])"
  RLN_DBG=64 RLN_DG_ABL=$v timeout -k 10 100 python tools/stamps_dgrad.py 2>/dev/null | tail -6
done
