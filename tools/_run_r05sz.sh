#!/bin/bash
# r05sz: accumulators initialised by the generated code (no clearing in the kernel body): GPU suite, then against the commit before
set -o pipefail
O=gpurun_out/r05sz; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for wl in resnet50 alexnet googlenet lenet; do
  bash tools/ab.sh $wl tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so | tee -a $O/ab.txt
done
