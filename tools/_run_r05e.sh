#!/bin/bash
# r05e: new GPU tests, GoogLeNet / ResNet A/B against round 4's library, MALL probe
set -o pipefail
O=gpurun_out/r05e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "skewed or off_the_tilings or small_launch or fresh_processes" > $O/pytest_new.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt; tail -3 $O/pytest_new.log
timeout -k 10 120 tools/probes/probe_mall_share > $O/probe_mall_share.txt 2>&1 || echo "mall probe failed"; cat $O/probe_mall_share.txt
bash tools/ab.sh googlenet tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_googlenet.txt 2>&1; cut -c1-330 $O/ab_googlenet.txt
bash tools/ab.sh resnet50 tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_resnet50.txt 2>&1; cut -c1-200 $O/ab_resnet50.txt
bash tools/ab.sh alexnet tools/ab/libescoin_prev.so caffe-escoin_amd/libescoin_hip.so > $O/ab_alexnet.txt 2>&1; cut -c1-200 $O/ab_alexnet.txt
