#!/bin/bash
# r05gk: pointwise switches once more on the final code (GoogLeNet set, experiments flavour)
set -o pipefail
O=gpurun_out/r05gk; mkdir -p $O; : > $O/knobs.txt
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu --workload googlenet 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-48s ms/step %.4f ' % ('$*', d['ms_per_step']) + ' '.join('%.1f' % l['us'] for l in d['roofline']['per_layer'][:41]))
" | tee -a $O/knobs.txt; }
for rep in 1 2; do
  run X=0
  run ESCOIN_JIT_DEPTH1=3
  run ESCOIN_JIT_DEPTH1=8
  run ESCOIN_JIT_DEPTH1=12
  run ESCOIN_JIT_PRIO_ROWS=0
  run ESCOIN_JIT_PRIO_ROWS=8
  run ESCOIN_JIT_DMA_SPREAD=50
  run ESCOIN_JIT_DMA_SPREAD=85
  run ESCOIN_JIT_YOUNG_PRIO=0
  run ESCOIN_JIT_SELF_ZERO=0
  run ESCOIN_HALF_WG=0
done
