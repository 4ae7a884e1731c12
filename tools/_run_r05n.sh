#!/bin/bash
# r05n: dense kernel, pixel-tile-major tile order + B tiles kept: tests, chain A/B through the switch
set -o pipefail
O=gpurun_out/r05n; mkdir -p $O; rm -f $O/chain_dense.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_tools_gpu.py -m gpu -x -q -k "dense or chain or conv_mode or crossover or auto" > $O/pytest_dense.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dense.log
EXP=$PWD/tools/ab/libescoin_exp.so
for rep in 1 2; do for V in 0 1; do
  echo "== ESCOIN_DENSE_BY_PTILE=$V" >> $O/chain_dense.txt
  ESCOIN_LIB=$EXP ESCOIN_DENSE_BY_PTILE=$V timeout -k 10 300 python tools/chain_dense.py 256 >> $O/chain_dense.txt 2>&1 || echo "chain_dense failed"
done; done
grep "==\|total" $O/chain_dense.txt
