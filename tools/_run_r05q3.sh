#!/bin/bash
# r05q3: what the next tile's quad table costs a launch (timing only: ESCOIN_DBG bit 12 on the ablations flavour leaves it unbuilt)
O=gpurun_out/r05q3; mkdir -p $O; : > $O/tab.txt
for rep in 1 2; do for L in res2 res3 res4 goog0 goog5; do for d in 0 4096; do
  echo "$L DBG=$d $(ESCOIN_LIB=$PWD/tools/ab/libescoin_abl.so ESCOIN_DBG=$d ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 100 2>&1 | tail -1)" | tee -a $O/tab.txt
done; done; done
