#!/bin/bash
# r05h: batch sweep with the unchained-tiling penalty; config-batch A/B against round 4's library
set -o pipefail
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 900 python tools/batch_sweep.py --batches 96,100,192,200,250,255,257,293,300,341,384,512 res2 res3 res4 res5 goog0 goog5 goog25 goog33 alex3 > $O/batch_sweep_line.md 2> $O/batch_sweep_line.err || { echo sweep failed; tail -5 $O/batch_sweep_line.err; }
cat $O/batch_sweep_line.md | cut -c1-400
for WL in resnet50 alexnet googlenet; do bash tools/ab.sh $WL tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_$WL.txt 2>&1; cut -c1-260 $O/ab_$WL.txt; done
