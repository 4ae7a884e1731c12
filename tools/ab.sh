#!/bin/bash
# A/B on the GPU box: bench of several builds of the library back to back on the same GPU.
#   bash tools/ab.sh resnet50 tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so ...   (paths from the repo root)
WL=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  ESCOIN_LIB=$PWD/$lib python bench.py --workload $WL --no-cpu 2> /tmp/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-22s %-10s ms/step %.4f  ' % ('$lib', '$WL', d['ms_per_step']) + ' '.join('%s:%.1f' % (l['layer'].split('_')[0][:12], l['us']) for l in d['roofline']['per_layer'][:8]), 'parity %.1e' % d['parity_max_rel_err'])
"
done; done
