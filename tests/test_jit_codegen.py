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
