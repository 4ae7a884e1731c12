#!/bin/bash
# GPU box: upper bounds for what the ENDS of an HBM-bound pointwise launch could give back.
# Per layer (conv2/3x3_reduce, inception_3b/1x1, 4e/1x1, 5b/1x1 -- one per image size), HBM-cold (four rotating pairs):
#   product build; ablation build as it is; ablation build with ESCOIN_DBG=128 (NO STORES: what a perfectly hidden
#   epilogue could give); ESCOIN_DBG=2 (no walk); stamp profile (ESCOIN_PROF=1: workgroup lifetimes min / mean / max --
#   launch time minus the MEAN lifetime is what perfect balance + free dispatch could give).
#   bash tools/pointwise_ends.sh <outdir> [layers...]
set -u
O=${1:-gpurun_out/pointwise_ends}; shift || true
ABL=$PWD/tools/ab/libescoin_abl.so
LAYERS=${*:-goog0 goog5 goog25 goog33}
mkdir -p $O
: > $O/times.txt
for L in $LAYERS; do
  echo -n "$L product : " >> $O/times.txt
  ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 2>&1 | tail -1 >> $O/times.txt || exit 1
  for D in 0 128 2 130; do
    echo -n "$L abl ESCOIN_DBG=$D : " >> $O/times.txt
    ESCOIN_LIB=$ABL ESCOIN_DBG=$D ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 2>&1 | tail -1 >> $O/times.txt || exit 1
  done
  ESCOIN_LIB=$ABL ESCOIN_PROF=1 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 2 > $O/stamp_$L.log 2>&1 || exit 1
  ESCOIN_LIB=$ABL ESCOIN_PROF=1 ESCOIN_DBG=128 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 2 > $O/stamp_nostore_$L.log 2>&1 || exit 1
  echo "$L done"
done
cat $O/times.txt
