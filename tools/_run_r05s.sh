#!/bin/bash
# r05s: the switches that moved in r05r, combined and on the other sets (experiments flavour)
set -o pipefail
O=gpurun_out/r05s; mkdir -p $O; : > $O/knobs.txt
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
run() { WL=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu --workload $WL 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-9s %-72s ms/step %.4f ' % ('$WL', '$*', d['ms_per_step']) + ' '.join('%.1f' % l['us'] for l in d['roofline']['per_layer'][:12]) + ' parity %.1e' % d['parity_max_rel_err'])
" | tee -a $O/knobs.txt; }
for rep in 1 2; do
  run resnet50 X=0
  run resnet50 ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4
  run resnet50 ESCOIN_JIT_PRIO_ROWS=8 ESCOIN_JIT_PRIO_WAVES=4
  run resnet50 ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4 ESCOIN_JIT_DMA_SPREAD=70
  run resnet50 ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4 ESCOIN_JIT_DMA_SPREAD=70 ESCOIN_XCD_MAP=1
  run alexnet X=0
  run alexnet ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4
  run alexnet ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4 ESCOIN_JIT_DMA_SPREAD=70
  run alexnet ESCOIN_JIT_PREFETCH=0
  run googlenet X=0
  run googlenet ESCOIN_JIT_PREFETCH=0
  run googlenet ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4
  run googlenet ESCOIN_JIT_DMA_SPREAD=70
done
