#!/bin/bash
# r05x: counted wait at a workgroup's first tile (three plane buffers): parity, then on / off through the switch (experiments flavour)
set -o pipefail
O=gpurun_out/r05x; mkdir -p $O; : > $O/knobs.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
run() { WL=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu --workload $WL 2> $O/err.txt > $O/last.json; python -c "
import sys, json
d = json.load(open('$O/last.json'))
print('%-9s %-40s ms/step %.4f ' % ('$WL', '$*', d['ms_per_step']) + ' '.join('%.1f' % l['us'] for l in d['roofline']['per_layer'][:41]) + ' parity %.1e' % d['parity_max_rel_err'])
" | tee -a $O/knobs.txt; }
for rep in 1 2 3; do
  run googlenet ESCOIN_FIRST_TILE_COUNTED=0
  run googlenet ESCOIN_FIRST_TILE_COUNTED=1
done
run resnet50 ESCOIN_FIRST_TILE_COUNTED=0
run resnet50 ESCOIN_FIRST_TILE_COUNTED=1
