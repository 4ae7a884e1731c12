#!/bin/bash
# r05k: two 4-wave workgroups per CU on the 3x3 layers (experiments flavour)
set -o pipefail
O=gpurun_out/r05k; mkdir -p $O
EXP=$PWD/tools/ab/libescoin_exp.so
: > $O/waves4_3x3.txt
for L in res2 res3 res4 res5 alex3; do
  for V in "" "ESCOIN_WAVES=4 ESCOIN_LDS_KB=32" "ESCOIN_WAVES=4 ESCOIN_LDS_KB=24" "ESCOIN_WAVES=4 ESCOIN_LDS_KB=16"; do
    echo -n "$L [$V] : " >> $O/waves4_3x3.txt
    env ESCOIN_LIB=$EXP ESCOIN_VERBOSE=1 $V ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 > $O/tmp.log 2>&1 || { echo FAILED >> $O/waves4_3x3.txt; tail -3 $O/tmp.log >> $O/waves4_3x3.txt; continue; }
    grep -o "jit: .*" $O/tmp.log | head -1 | cut -c1-170 >> $O/waves4_3x3.txt
    tail -1 $O/tmp.log | sed 's/.*launches): //' >> $O/waves4_3x3.txt
  done
done
cat $O/waves4_3x3.txt
