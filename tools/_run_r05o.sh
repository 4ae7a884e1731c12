#!/bin/bash
# r05o: dense kernel, second half of the grid started late (experiments flavour)
set -o pipefail
O=gpurun_out/r05o; mkdir -p $O; rm -f $O/chain_dense.txt
EXP=$PWD/tools/ab/libescoin_exp.so
for V in 0 1 2 4 0 1 2; do
  echo "== ESCOIN_DENSE_STAGGER=$V" >> $O/chain_dense.txt
  ESCOIN_LIB=$EXP ESCOIN_DENSE_STAGGER=$V timeout -k 10 300 python tools/chain_dense.py 256 >> $O/chain_dense.txt 2>&1 || echo "chain_dense failed"
done
grep "==\|total" $O/chain_dense.txt
