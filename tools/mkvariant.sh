#!/bin/bash
# Builds a VARIANT of the library for a same-box A/B (tools/ab.sh): the stream loop regenerated with
# generator knobs from the environment, everything else as in the tree.
#   tools/mkvariant.sh <tag> [GEN_VAR=value ...]      ->  tools/ab/libescoin_<tag>.so
# e.g. tools/mkvariant.sh al0 ESC_GEN_ALIGN=0 ; tools/mkvariant.sh al4 ESC_GEN_ALIGN=4
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
CS=$PWD/caffe-escoin_amd/csrc
make -C "$CS" -j4 > /dev/null          # the tree's own objects (everything but the tiled kernel is shared)
mkdir -p tools/ab; D=/tmp/var_$TAG; rm -rf $D; mkdir -p $D
env "$@" python3 $CS/gen_stream_loop.py > $D/stream_loop_asm.inc
cp $CS/sconv_tiled.hip $D/           # (a quoted #include looks beside the including file first)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$PWD/include -I$CS -Wno-unused-result \
  -Wno-inline-asm -fvisibility=hidden -DESCOIN_BUILD ${VARIANT_CFLAGS:-} -c -o $D/sconv_tiled.o $D/sconv_tiled.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libescoin_$TAG.so \
  $CS/escoin_capi.o $CS/sconv_generic.o $D/sconv_tiled.o $CS/dense_mfma.o $CS/sconv_lowered.o $CS/stream_builder.o $CS/jit_codegen.o $CS/jit_module.o -lamd_comgr
echo "built tools/ab/libescoin_$TAG.so ($*)"
