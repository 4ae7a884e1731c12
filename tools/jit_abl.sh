#!/bin/bash
# GPU box: timing-only ablations of the generated-code kernel, one ResNet layer shape back to back.
#   bash tools/jit_abl.sh res4 "0 1 2 3 4 8"   (ESCOIN_JIT_ABL values; results are WRONG for != 0)
L=${1:-res4}; shift
for a in ${1:-0 1 2 4 8}; do
  echo -n "ABL=$a  "
  ESCOIN_JIT_ABL=$a python tools/one_layer.py $L 100 2>&1 | tail -1
done
