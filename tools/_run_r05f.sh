#!/bin/bash
# r05f: the global (LPT) start of the channel deal: A/B through the experiments flavour, then the product's skew runs
set -o pipefail
O=gpurun_out/r05f; mkdir -p $O
EXP=$PWD/tools/ab/libescoin_exp.so
for WL in resnet50 alexnet; do for D in uniform i iii; do for rep in 1 2; do for V in 0 1; do
  ESCOIN_LIB=$EXP ESCOIN_DEAL_LPT=$V timeout -k 10 300 python bench.py --no-cpu --workload $WL --sparsity-dist $D 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$WL $D LPT=$V ms/step %.4f ' % d['ms_per_step'] + ' '.join('%s:%.1f' % (l['layer'][:9], l['us']) for l in d['roofline']['per_layer']) + ' | deal ' + ' '.join('%.3f' % c['slowest_over_mean'] for c in d.get('channel_deal', [])) + ' parity %.1e' % d['parity_max_rel_err'])
" | tee -a $O/deal_ab.txt
done; done; done; done
for D in uniform i ii iii; do
  timeout -k 10 300 python bench.py --no-cpu --sparsity-dist $D > $O/bench_resnet50_dist_$D.json 2> $O/bench_resnet50_dist_$D.err || echo "skew $D failed"
  timeout -k 10 300 python bench.py --no-cpu --workload alexnet --sparsity-dist $D > $O/bench_alexnet_dist_$D.json 2> $O/bench_alexnet_dist_$D.err || echo "skew alex $D failed"
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
