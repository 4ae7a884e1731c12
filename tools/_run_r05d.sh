#!/bin/bash
# r05d: half-workgroup rule: GPU suite, GoogLeNet set A/B against the round-4 library, per-layer A/B through the knob
set -o pipefail
O=gpurun_out/r05d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt; tail -3 $O/pytest_gpu.log
bash tools/ab.sh googlenet tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_googlenet.txt 2>&1; cat $O/ab_googlenet.txt | cut -c1-400
bash tools/ab.sh resnet50 tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_resnet50.txt 2>&1; cat $O/ab_resnet50.txt | cut -c1-200
EXP=$PWD/tools/ab/libescoin_exp.so
: > $O/half.txt
for L in goog1 goog2 goog3 goog4 goog5 goog6 goog7 goog8; do
  for V in "ESCOIN_HALF_WG=0" "ESCOIN_HALF_WG=-1"; do
    echo -n "$L [$V] : " >> $O/half.txt
    env ESCOIN_LIB=$EXP ESCOIN_VERBOSE=1 $V ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 > $O/tmp.log 2>&1 || { echo FAILED >> $O/half.txt; tail -3 $O/tmp.log >> $O/half.txt; continue; }
    grep -o "jit: .*" $O/tmp.log | head -1 | cut -c1-130 >> $O/half.txt
    tail -1 $O/tmp.log | sed 's/.*launches): //' >> $O/half.txt
  done
done
cat $O/half.txt
