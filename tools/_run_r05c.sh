#!/bin/bash
# r05c: two 4-wave workgroups per CU on the HBM-bound pointwise layers (experiments flavour)
set -o pipefail
O=gpurun_out/r05c; mkdir -p $O
EXP=$PWD/tools/ab/libescoin_exp.so
: > $O/waves4.txt
for L in goog0 goog1 goog2 goog3 goog4 goog5 goog7 goog8 goog9 goog12; do
  for V in "" "ESCOIN_WAVES=4 ESCOIN_JIT_NBUF=2 ESCOIN_LDS_KB=32" "ESCOIN_WAVES=4 ESCOIN_JIT_NBUF=3 ESCOIN_LDS_KB=24" "ESCOIN_WAVES=4 ESCOIN_JIT_NBUF=2 ESCOIN_LDS_KB=24"; do
    echo -n "$L [$V] : " >> $O/waves4.txt
    env ESCOIN_LIB=$EXP ESCOIN_VERBOSE=1 $V ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 > $O/tmp.log 2>&1 || { echo FAILED >> $O/waves4.txt; tail -3 $O/tmp.log >> $O/waves4.txt; continue; }
    grep -o "jit: .*" $O/tmp.log | head -1 | cut -c1-150 >> $O/waves4.txt
    tail -1 $O/tmp.log >> $O/waves4.txt
  done
done
cat $O/waves4.txt
