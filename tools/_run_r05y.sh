#!/bin/bash
# r05y: ordinary vs non-temporal stores of the pointwise top blob: as producer / consumer pairs and per GoogLeNet layer (three runs each)
set -o pipefail
O=gpurun_out/r05y; mkdir -p $O
timeout -k 10 600 python tools/producer_consumer.py | tee $O/pairs.txt
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --no-cpu --workload googlenet > $O/goog_plain_$rep.json 2> /dev/null
  timeout -k 10 300 python bench.py --no-cpu --workload googlenet --stream-stores > $O/goog_nt_$rep.json 2> /dev/null
done
python - <<'PY'
import json
P=[json.load(open('gpurun_out/r05y/goog_plain_%d.json'%r)) for r in (1,2,3)]
N=[json.load(open('gpurun_out/r05y/goog_nt_%d.json'%r)) for r in (1,2,3)]
for i,l in enumerate(P[0]['roofline']['per_layer']):
    p=min(x['roofline']['per_layer'][i]['us'] for x in P); n=min(x['roofline']['per_layer'][i]['us'] for x in N)
    print('%-28s %6.1f %6.1f %+5.1f%%'%(l['layer'],p,n,(n/p-1)*100))
print(min(x['ms_per_step'] for x in P), min(x['ms_per_step'] for x in N))
PY
