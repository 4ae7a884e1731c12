#!/bin/bash
# usage: sweep.sh model
M=$1
for w in 1 2 4 8; do for k in 16 32 64; do
  ESCOIN_WAVES=$w ESCOIN_LDS_KB=$k python tools/caffe_test.py --model $M --iterations 3 > gpurun_out/sw_${M}_${w}_${k}.txt 2>&1
done; done
