#!/bin/bash
# r05v: second pass of the re-tuned defaults (no touches for any small code incl. res2; 4 x 4 layers keep them); tiling dump
set -o pipefail
O=gpurun_out/r05v; mkdir -p $O
for wl in resnet50 googlenet alexnet; do
  bash tools/ab.sh $wl tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so | tee -a $O/ab.txt
done
ESCOIN_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu --workload resnet50 2>&1 > /dev/null | grep "jit:" | sort -u > $O/verbose_resnet.txt
ESCOIN_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu --workload googlenet 2>&1 > /dev/null | grep "jit:" > $O/verbose_goog.txt
for rep in 1 2; do for lib in tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so; do
  ESCOIN_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu --workload googlenet > $O/goog_$(basename $lib .so)_$rep.json 2> /dev/null
done; done
