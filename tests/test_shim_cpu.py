"""The C++ Layer/Blob shim in Caffe::CPU mode, float and double: the forward cases of the reference's own typed conv
tests (src/caffe/test/test_convolution_layer.cpp:231-265, 267-309, 443-468, 470-496, 498-589) plus pruned weights,
the BASELINE shapes, Reshape-after-align and the conv-mode flips -- on a machine without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "caffe-escoin_amd", "caffe_shim", "shim_selftest")


def test_shim_selftest_cpu_mode_float_and_double():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "caffe-escoin_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "caffe-escoin_amd", "caffe_shim")], stdout=subprocess.DEVNULL)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")     # CPU mode must not need a device
    out = subprocess.run([EXE, "--cpu-only"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    text = out.stdout.decode()
    print(text)
    assert out.returncode == 0, text
    assert "all OK" in text
    for case in ("TestSimpleConvolution", "TestDilatedConvolution", "Test1x1Convolution", "TestSimpleConvolutionGroup",
                 "TestSobelConvolution"):
        for dtype in ("float", "double"):
            assert any(l.startswith(dtype) and " CPU " in l and case in l and l.rstrip().endswith("OK")
                       for l in text.splitlines()), (dtype, case)
