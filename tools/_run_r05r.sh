#!/bin/bash
# r05r: generator / tiling switches re-swept on the ResNet set now that the walk carries no literal moves (experiments flavour)
set -o pipefail
O=gpurun_out/r05r; mkdir -p $O; : > $O/knobs.txt
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
for rep in 1 2; do
for V in "" "ESCOIN_JIT_DEPTH=1" "ESCOIN_JIT_YOUNG_PRIO=0" "ESCOIN_JIT_YOUNG_PRIO=2" "ESCOIN_JIT_PRIO_ROWS=2 ESCOIN_JIT_PRIO_WAVES=4" "ESCOIN_JIT_PRIO_ROWS=4 ESCOIN_JIT_PRIO_WAVES=4" "ESCOIN_JIT_DMA_SPREAD=70" "ESCOIN_JIT_DMA_SPREAD=50" "ESCOIN_LDS_KB=48" "ESCOIN_LDS_KB=64" "ESCOIN_LDS_KB=24" "ESCOIN_JIT_PREFETCH=0" "ESCOIN_BALANCE=0" "ESCOIN_XCD_MAP=0" "ESCOIN_XCD_MAP=1"; do
  env $V timeout -k 10 300 python bench.py --no-cpu 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-48s ms/step %.4f ' % ('$V', d['ms_per_step']) + ' '.join('%s:%.1f' % (l['layer'][:4], l['us']) for l in d['roofline']['per_layer']) + ' parity %.1e' % d['parity_max_rel_err'])
" | tee -a $O/knobs.txt
done; done
