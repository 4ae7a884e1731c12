#!/bin/bash
# r05g: what the tiling cost model could have chosen at the batch sweep's cliffs
set -o pipefail
O=gpurun_out/r05g; mkdir -p $O
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
( python tools/tiling_oracle.py res4 96 192 257 293 256
  python tools/tiling_oracle.py res5 192 293 341 256
  python tools/tiling_oracle.py res3 200 257
  python tools/tiling_oracle.py goog5 96 257
  python tools/tiling_oracle.py goog33 96
  python tools/tiling_oracle.py alex3 96 257 ) > $O/tiling_oracle.txt 2>&1
cat $O/tiling_oracle.txt
