#!/bin/bash
# r05q2: stamps of the tile top after the accumulator clearing went (where the time of a tile boundary is now)
O=gpurun_out/r05q2; mkdir -p $O
for L in res2 res3 goog0; do
  ESCOIN_LIB=$PWD/tools/ab/libescoin_abl.so ESCOIN_PROF=1 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 2 > $O/stamp_$L.log 2>&1
  echo "== $L"; grep "wave0 cycles" $O/stamp_$L.log | tail -1
done
