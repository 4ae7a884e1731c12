#!/bin/bash
# r05z2: per-layer: the commit before (never) / by size (the new default) / always
set -o pipefail
O=gpurun_out/r05z2; mkdir -p $O
for rep in 1 2 3; do
  ESCOIN_LIB=$PWD/tools/ab/libescoin_prev.so timeout -k 10 300 python bench.py --no-cpu --workload googlenet > $O/goog_prev_$rep.json 2> /dev/null
  timeout -k 10 300 python bench.py --no-cpu --workload googlenet > $O/goog_size_$rep.json 2> /dev/null
  timeout -k 10 300 python bench.py --no-cpu --workload googlenet --stream-stores always > $O/goog_always_$rep.json 2> /dev/null
  timeout -k 10 300 python bench.py --no-cpu --workload googlenet --stream-stores never > $O/goog_never_$rep.json 2> /dev/null
done
python - <<'PY'
import json
def L(n): return [json.load(open('gpurun_out/r05z2/goog_%s_%d.json'%(n,r))) for r in (1,2,3)]
S={n:L(n) for n in ('prev','never','size','always')}
for i,l in enumerate(S['prev'][0]['roofline']['per_layer']):
    v={n:min(x['roofline']['per_layer'][i]['us'] for x in S[n]) for n in S}
    print('%-28s prev %6.1f never %6.1f size %6.1f always %6.1f'%(l['layer'],v['prev'],v['never'],v['size'],v['always']))
print({n:min(x['ms_per_step'] for x in S[n]) for n in S})
PY
