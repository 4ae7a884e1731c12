#!/bin/bash
# r05l: dense kernel with the next step's operands settled in front of the epilogue's stores: tests, chain A/B against round 4's library
set -o pipefail
O=gpurun_out/r05l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_tools_gpu.py -m gpu -x -q -k "dense or chain or conv_mode or crossover or auto" > $O/pytest_dense.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dense.log
for rep in 1 2; do for LIB in tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so; do
  echo "== $LIB" >> $O/chain_dense.txt
  ESCOIN_LIB=$PWD/$LIB timeout -k 10 300 python tools/chain_dense.py 256 >> $O/chain_dense.txt 2>&1 || echo "chain_dense failed"
done; done
grep "==\|total" $O/chain_dense.txt
python tools/caffe_test.py --model resnet50_chain --batch 256 2>&1 | tail -12 > $O/caffe_test_chain.txt; cat $O/caffe_test_chain.txt
