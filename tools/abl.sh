# A/B of one env toggle on the default bench: usage tools/abl.sh VAR v1 v2 ...
var=${1:-RLN_DBG}; shift
for v in "$@"; do
  env $var=$v timeout -k 10 150 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={x['name']:x for x in d['kernel_classes']}
print('$var=$v','img/s',d['value'],'step',d['ms_per_step'],' '.join('%s %.2f'%(n.replace('dense_conv3x3_','d_'),k[n]['ms_per_step']) for n in k))"
done
