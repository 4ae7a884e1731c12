#!/bin/bash
# r05u: the re-tuned defaults (s_setprio rows, DMA spread, XCD threshold, code touches) as the product against the commit before
set -o pipefail
O=gpurun_out/r05u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for wl in resnet50 alexnet googlenet; do
  bash tools/ab.sh $wl tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so | tee -a $O/ab.txt
done
for rep in 1 2; do for lib in tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so; do
  ESCOIN_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu --workload googlenet > $O/goog_$(basename $lib .so)_$rep.json 2> /dev/null
done; done
