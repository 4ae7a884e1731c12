#!/bin/bash
# GPU box: the raw evidence behind DESIGN.md's "what binds it" claims, one file per claim under
# gpurun_out/<tag>/ (tools/make_evidence.py condenses them into profiles/).
#   bash tools/evidence.sh r04 [abl-lib]
#  * probe_valu_rate / probe_mixload outputs
#  * in-kernel stamp profile (ESCOIN_PROF=1 on the -DESCOIN_ABLATIONS build) of the four ResNet shapes and
#    three GoogLeNet sizes, HBM-cold (four rotating blob pairs)
#  * ESCOIN_JIT_ABL ablation table per ResNet shape (timing only; results are wrong for != 0)
#   EVIDENCE_ONLY=stamp: the stamp profile only
set -u
TAG=${1:-r04}
ABL=${2:-$PWD/tools/ab/libescoin_abl.so}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "${EVIDENCE_ONLY:-}" != "stamp" ]; then
( cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o probe_valu_rate probe_valu_rate.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o probe_mixload probe_mixload.hip ) > $OUT/probe_build.log 2>&1
timeout -k 10 120 tools/probes/probe_valu_rate > $OUT/probe_valu_rate.txt 2>&1
timeout -k 10 180 tools/probes/probe_mixload > $OUT/probe_mixload.txt 2>&1
echo probes done
fi
for L in res2 res3 res4 res5 goog0 goog5 goog13 goog25 goog33 goog37; do
  ESCOIN_LIB=$ABL ESCOIN_PROF=1 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 2 > $OUT/stamp_$L.log 2>&1
  ESCOIN_LIB=$ABL ESCOIN_VERBOSE=1 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 > $OUT/time_abl_$L.log 2>&1
  ESCOIN_VERBOSE=1 ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 200 > $OUT/time_$L.log 2>&1
  echo "stamp $L done"
done
[ "${EVIDENCE_ONLY:-}" = "stamp" ] && exit 0
for L in res2 res3 res4 res5; do
  for a in 0 1 2 4 8 3 7; do
    echo "ABL=$a $(ESCOIN_LIB=$ABL ESCOIN_JIT_ABL=$a ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 100 2>&1 | tail -1)" >> $OUT/jit_abl_$L.txt
  done
  for d in 1 2 3 4 128 64; do
    echo "DBG=$d $(ESCOIN_LIB=$ABL ESCOIN_DBG=$d ONE_LAYER_BUFS=4 timeout -k 10 120 python tools/one_layer.py $L 100 2>&1 | tail -1)" >> $OUT/jit_abl_$L.txt
  done
  echo "abl $L done"
done
python bench.py --no-cpu > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err
python bench.py --no-cpu --workload googlenet > $OUT/bench_googlenet.json 2> $OUT/bench_googlenet.err
echo all done
