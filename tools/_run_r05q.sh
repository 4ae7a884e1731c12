#!/bin/bash
# r05q: workgroup lifetimes of the layers with several tiles per workgroup (what an atomic tile queue could balance)
O=gpurun_out/r05q; mkdir -p $O
for L in res2 res3 res4 goog0 goog5; do
  ESCOIN_LIB=$PWD/tools/ab/libescoin_abl.so ESCOIN_PROF=1 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 2 > $O/stamp_$L.log 2>&1
  echo "== $L"; grep -i "life\|xcd\|percentile\|start" $O/stamp_$L.log | head -12
done
