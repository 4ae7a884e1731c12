#!/bin/bash
# Builds a NON-PRODUCT flavour of the library into tools/ab/ (never into the package directory; csrc/knobs.h):
#   tools/mkabl.sh          -> tools/ab/libescoin_abl.so   -DESCOIN_ABLATIONS: in-kernel stamp profile (ESCOIN_PROF=1),
#                                                          wrong-result timing switches (ESCOIN_DBG, ESCOIN_JIT_ABL,
#                                                          ESCOIN_DENSE_ABL) and every tuning switch
#   tools/mkabl.sh exp      -> tools/ab/libescoin_exp.so   -DESCOIN_EXPERIMENTS: the tuning switches only (tilings,
#                                                          buffers, kernel selection ...); results stay right
# ABL_CFLAGS adds flags (e.g. -DESCOIN_PROF_STARTUP), ABL_NAME overrides the output tag.  Select a flavour at run
# time with ESCOIN_LIB=$PWD/tools/ab/libescoin_<tag>.so (the Python binding; the product never reads it).
set -e
cd "$(dirname "$0")/../caffe-escoin_amd/csrc"
FLAVOUR=${1:-abl}
case $FLAVOUR in
  abl) DEF=-DESCOIN_ABLATIONS ;;
  exp) DEF=-DESCOIN_EXPERIMENTS ;;
  *) echo "usage: $0 [abl|exp]"; exit 2 ;;
esac
TAG=${ABL_NAME:-$FLAVOUR}
make stream_loop_asm.inc
O=/tmp/abl_$TAG
mkdir -p $O ../../tools/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -Wno-inline-asm -fvisibility=hidden -DESCOIN_BUILD $DEF ${ABL_CFLAGS:-}"
pids=()
for s in escoin_capi sconv_generic sconv_tiled dense_mfma sconv_lowered code_memory; do
  /opt/rocm/bin/hipcc $F -c -o $O/$s.o $s.hip & pids+=($!)
done
for s in stream_builder jit_codegen jit_module; do
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include $DEF ${ABL_CFLAGS:-} -c -o $O/$s.o $s.cpp & pids+=($!)
done
# Caffe::CPU mode: plain host translation units (the same flags as the product Makefile)
CPUF="-x c++ -O3 -std=c++17 -fPIC -fvisibility=hidden -DESCOIN_BUILD -ffp-contract=off -I. -I../../include"
/opt/rocm/bin/hipcc $CPUF -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $DEF -c -o $O/sconv_cpu.o sconv_cpu.cpp & pids+=($!)
/opt/rocm/bin/hipcc $CPUF -DESC_CPU_ISA=2 -mavx2 -mfma -c -o $O/sconv_cpu_kernel_avx2.o sconv_cpu_kernel.cpp & pids+=($!)
/opt/rocm/bin/hipcc $CPUF -DESC_CPU_ISA=512 -mavx512f -mavx512vl -mavx512dq -mavx2 -mfma -c -o $O/sconv_cpu_kernel_avx512.o sconv_cpu_kernel.cpp & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done     # a failed compile aborts the script (set -e)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libescoin_$TAG.so $O/*.o -lamd_comgr -lhsa-runtime64 -lpthread
ls -la ../../tools/ab/libescoin_$TAG.so
