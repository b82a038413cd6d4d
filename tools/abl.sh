for v in 0 1; do
  if [ $v = 1 ]; then export RLN_NO_SPLITK=1; fi
  timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={x['name']:x for x in d['kernel_classes']}
print('no_splitk',$v,'img/s',d['value'],'step',d['ms_per_step'],'fwd',k['dense_conv3x3_fwd']['ms_per_step'],'dgrad',k['dense_conv3x3_dgrad']['ms_per_step'],'wgrad',k['dense_conv3x3_wgrad']['ms_per_step'],'bn',k['bn_stats_affine']['ms_per_step'])"
done
