for v in 0 1 2; do
  RLN_STRIP_MODE=$v timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={x['name']:x for x in d['kernel_classes']}
print('strip_mode',$v,'img/s',d['value'],'step',d['ms_per_step'],'fwd',k['dense_conv3x3_fwd']['ms_per_step'],k['dense_conv3x3_fwd']['tflops'])"
done
