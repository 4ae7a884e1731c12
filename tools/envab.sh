#!/bin/bash
# env-variable A/B on the GPU box, on the EXPERIMENTS flavour (tools/mkabl.sh exp; the product build reads no tuning
# switches):  bash tools/envab.sh googlenet "ESCOIN_NBUF=3" "ESCOIN_LDS_KB=48" ...
WL=$1; shift
for rep in 1 2; do
for e in "" "$@"; do
  env ESCOIN_LIB=${ESCOIN_LIB:-$PWD/tools/ab/libescoin_exp.so} $e python bench.py --workload $WL --no-cpu 2> /tmp/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-28s %-10s ms/step %.4f  ' % ('$e', '$WL', d['ms_per_step']) + ' '.join('%.0f' % l['us'] for l in d['roofline']['per_layer'][:40]), 'parity %.1e' % d['parity_max_rel_err'])
"
done; done
