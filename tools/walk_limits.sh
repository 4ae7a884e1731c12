#!/bin/bash
# GPU box: what bounds the generated-code walk -- SQ counters of one launch and timing-only ablations of the code
# generator / kernel body on single layers (tools/one_layer.py, four rotating blob pairs: HBM-cold).
#   bash tools/walk_limits.sh <outdir>        -> <outdir>/pmc_<layer>.txt, <outdir>/abl.txt
set -o pipefail
O=${1:-gpurun_out/walk_limits}
ABL=${2:-$PWD/tools/ab/libescoin_abl.so}      # (the wrong-result switches exist in the ablation build only)
mkdir -p $O
G="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM;SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU;SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU;SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES"
for L in res3 res5 goog25; do
  ONE_LAYER_BUFS=4 bash tools/pmc_pass.sh $O/pmc_$L "$G" tools/one_layer.py $L 10 > $O/pmc_$L.txt 2>&1 || exit 1
  rm -rf $O/pmc_$L
done
: > $O/abl.txt
for L in res2 res3 res4 res5; do
  for V in "0 0" "16 0" "32 0" "64 0" "112 0" "4 0" "16388 0" "512 0" "0 1" "0 1025" "0 2049"; do
    set -- $V
    echo -n "$L ESCOIN_JIT_ABL=$1 ESCOIN_DBG=$2 : " >> $O/abl.txt
    ESCOIN_LIB=$ABL ESCOIN_JIT_ABL=$1 ESCOIN_DBG=$2 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 100 2>&1 | tail -1 | sed 's/.*launches): //' >> $O/abl.txt || exit 1
  done
done
for L in goog0 goog5 goog9 goog25 goog33; do
  for V in "0 0" "16 0" "64 0" "112 0"; do
    set -- $V
    echo -n "$L ESCOIN_JIT_ABL=$1 ESCOIN_DBG=$2 : " >> $O/abl.txt
    ESCOIN_LIB=$ABL ESCOIN_JIT_ABL=$1 ESCOIN_DBG=$2 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 100 2>&1 | tail -1 | sed 's/.*launches): //' >> $O/abl.txt || exit 1
  done
done
cat $O/abl.txt
