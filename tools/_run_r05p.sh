#!/bin/bash
# r05p: weights through the scalar cache on the 3x3 / 5x5 layers: GPU suite, A/B through the switch (experiments flavour), product vs round 4
set -o pipefail
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
EXP=$PWD/tools/ab/libescoin_exp.so
for WL in resnet50 alexnet; do for rep in 1 2; do for V in 0 1; do
  ESCOIN_LIB=$EXP ESCOIN_JIT_SWEIGHTS=$V timeout -k 10 300 python bench.py --no-cpu --workload $WL 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$WL SWEIGHTS=$V ms/step %.4f ' % d['ms_per_step'] + ' '.join('%s:%.1f' % (l['layer'][:9], l['us']) for l in d['roofline']['per_layer']) + ' code MB %.1f parity %.1e' % (d['generated_code_bytes'] / 1e6, d['parity_max_rel_err']))
" | tee -a $O/sweights_ab.txt
done; done; done
for WL in resnet50 alexnet googlenet; do bash tools/ab.sh $WL tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_$WL.txt 2>&1; cut -c1-200 $O/ab_$WL.txt; done
