#!/bin/bash
# r05z: non-temporal stores by size as the default: GPU suite, then against the commit before
set -o pipefail
O=gpurun_out/r05z; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for wl in googlenet resnet50 alexnet lenet; do
  bash tools/ab.sh $wl tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so | tee -a $O/ab.txt
done
timeout -k 10 600 python tools/producer_consumer.py | tee $O/pairs.txt
