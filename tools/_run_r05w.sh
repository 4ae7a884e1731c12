#!/bin/bash
# r05w: res2 (one workgroup column, seven tiles per workgroup): plane buffer count / size, non-temporal loads (experiments flavour)
set -o pipefail
O=gpurun_out/r05w; mkdir -p $O; : > $O/knobs.txt
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
run() { WL=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu --workload $WL 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-9s %-60s ms/step %.4f ' % ('$WL', '$*', d['ms_per_step']) + ' '.join('%.1f' % l['us'] for l in d['roofline']['per_layer'][:12]) + ' parity %.1e' % d['parity_max_rel_err'])
" | tee -a $O/knobs.txt; }
for rep in 1 2; do
  run resnet50 X=0
  run resnet50 ESCOIN_JIT_NBUF=3 ESCOIN_LDS_KB=36
  run resnet50 ESCOIN_JIT_NBUF=3 ESCOIN_LDS_KB=48
  run resnet50 ESCOIN_JIT_NBUF=3 ESCOIN_LDS_KB=24
  run resnet50 ESCOIN_NT=0
  run resnet50 ESCOIN_JIT_PRIO_ROWS=8
  run resnet50 ESCOIN_JIT_PRIO_ROWS=2
  run resnet50 ESCOIN_JIT_PRIO_WAVES=8
done
