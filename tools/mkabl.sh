#!/bin/bash
# Builds caffe-escoin_amd/libescoin_abl.so: the library with -DESCOIN_ABLATIONS (in-kernel stamp
# profile under ESCOIN_PROF=1, timing-only ablations under ESCOIN_DBG).  Not a product build.
set -e
cd "$(dirname "$0")/../caffe-escoin_amd/csrc"
make stream_loop_asm.inc
mkdir -p /tmp/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -Wno-inline-asm -fvisibility=hidden -DESCOIN_BUILD -DESCOIN_ABLATIONS ${ABL_CFLAGS:-}"
pids=()
for s in escoin_capi sconv_generic sconv_tiled dense_mfma sconv_lowered; do
  /opt/rocm/bin/hipcc $F -c -o /tmp/abl/$s.o $s.hip & pids+=($!)
done
for s in stream_builder jit_codegen jit_module; do
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include -c -o /tmp/abl/$s.o $s.cpp & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done     # a failed compile aborts the script (set -e)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libescoin_${ABL_NAME:-abl}.so /tmp/abl/*.o -lamd_comgr
ls -la ../libescoin_${ABL_NAME:-abl}.so
