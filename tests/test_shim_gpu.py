"""The C++ host side: the Caffe-compatible Layer/Blob shim (caffe-escoin_amd/caffe_shim/) driven
like the reference's conv tests drive ConvolutionLayer, with WeightAlign() in SCONV mode."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "caffe-escoin_amd", "caffe_shim", "shim_selftest")


def test_shim_header_compiles_and_links():
    """CPU-side check: the header-only shim and its self-test build against the C ABI."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "caffe-escoin_amd", "csrc"), "-j4"],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "caffe-escoin_amd", "caffe_shim")],
                          stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_shim_selftest_on_gpu():
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    out = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    print(text)
    assert out.returncode == 0, text
    assert "all OK" in text
