#!/bin/bash
# r05i: dense strided-pointwise path through registers: parity test, chain A/B through the switch
set -o pipefail
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense" > $O/pytest_dense.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dense.log
EXP=$PWD/tools/ab/libescoin_exp.so
for rep in 1 2; do for V in 0 1; do
  echo "== ESCOIN_DENSE_S2=$V" >> $O/chain_dense.txt
  ESCOIN_LIB=$EXP ESCOIN_DENSE_S2=$V timeout -k 10 300 python tools/chain_dense.py 256 >> $O/chain_dense.txt 2>&1 || echo "chain_dense failed"
done; done
cat $O/chain_dense.txt
