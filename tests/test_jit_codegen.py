"""WeightAlign's code generator (csrc/jit_codegen.{h,cpp}) without a GPU:
  * every instruction form it encodes, with random operands, against the assembler (llvm-mc);
  * the generated code of whole layers is interpreted by tests/cpp/emulate_tiled.cpp (LDS reads
    landing only at counted waits) and compared with a dense convolution -- that runs in
    tests/test_stream_builder.py next to the LDS-staged stream's emulation."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "caffe-escoin_amd", "csrc")
LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"


@pytest.mark.skipif(not os.path.exists(LLVM_MC), reason="llvm-mc not installed")
def test_encoders_agree_with_the_assembler(tmp_path):
    exe = str(tmp_path / "jit_encode_dump")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + CSRC, "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "jit_encode_dump.cpp")])
    lines = subprocess.run([exe], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    texts = [l.split("|")[0].strip() for l in lines]
    mine = [[int(w, 16) for w in l.split("|")[1].split()] for l in lines]
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-show-encoding"], input="\n".join(texts) + "\n",
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0 and "error" not in r.stderr, r.stderr[:2000]
    encs = re.findall(r"encoding: \[([^\]]*)\]", r.stdout)
    assert len(encs) == len(texts) > 800
    for text, words, enc in zip(texts, mine, encs):
        b = bytes(int(x, 16) for x in enc.split(","))
        theirs = [int.from_bytes(b[i:i + 4], "little") for i in range(0, len(b), 4)]
        assert words == theirs, "%s: %s vs llvm-mc %s" % (text, [hex(w) for w in words], [hex(w) for w in theirs])


def test_code_object_without_the_assembler_is_the_assemblers(tmp_path):
    """jit_module.h jit_wrap: the template's .text grown by the generated code, headers / section table / symbols
    rewritten, against jit_assemble (libamd_comgr: assembler + linker) on the same page-padded code -- byte for byte."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()      # (the library's objects: the check links jit_module.o and what it refers to)
    objs = [os.path.join(CSRC, o) for o in ("jit_module.o", "escoin_capi.o", "sconv_generic.o", "sconv_tiled.o", "dense_mfma.o",
                                            "sconv_lowered.o", "code_memory.o", "stream_builder.o", "jit_codegen.o", "sconv_cpu.o",
                                            "sconv_cpu_kernel_avx2.o", "sconv_cpu_kernel_avx512.o")]
    assert all(os.path.exists(o) for o in objs)
    obj, exe = str(tmp_path / "jit_wrap_check.o"), str(tmp_path / "jit_wrap_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", "-c", os.path.join(ROOT, "tests", "cpp", "jit_wrap_check.cpp"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-o", exe, obj] + objs + ["-lamd_comgr", "-lhsa-runtime64", "-lpthread"])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode()
    assert out.returncode == 0 and "all cases OK" in text, text
    assert text.count("identical") == 6
