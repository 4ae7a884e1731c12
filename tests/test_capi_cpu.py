"""C-ABI host logic without a GPU: the library loads, exports every symbol include/escoin.h
declares, validates geometry like LayerSetUp/Reshape, and its GPU entry points FAIL LOUDLY (no silent
fallback) on a machine without a HIP device; Caffe::CPU mode is explicit (tests/test_cpu_mode.py)."""
import ctypes as C
import os
import re

import numpy as np
import pytest


def _no_gpu(pkg):
    return pkg.device_count() == 0


def test_header_symbols_are_all_exported(pkg):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "escoin.h")).read()
    declared = sorted(set(re.findall(r"ESCOIN_API[^;(]*?(escoin_\w+)\s*\(", hdr)))
    assert declared, "no declarations parsed from include/escoin.h"
    assert sorted(pkg.API_SYMBOLS) == declared
    lib = C.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libescoin_hip.so does not export %s" % name


def test_out_shape_matches_reference_formula(pkg, oracle):
    # conv_layer.cpp:8-22 on the reference's test shapes and the config layers
    for (h, k, p, s, d) in [(6, 3, 0, 2, 1), (8, 3, 0, 1, 2), (56, 3, 1, 1, 1), (27, 5, 2, 1, 1),
                            (12, 5, 0, 1, 1), (227, 11, 0, 4, 1), (7, 3, 1, 1, 1)]:
        desc = pkg.ConvDesc(1, 1, h, h, 1, k, k, p, p, s, s, d, d, 1, 0, 0)
        want = (h + 2 * p - (d * (k - 1) + 1)) // s + 1
        assert pkg.out_shape(desc) == (want, want)
        assert oracle.out_hw(oracle.geom(1, h, h, 1, k, k, p, p, s, s, d, d)) == (want, want)


def test_padded_len(pkg):
    d = pkg.ConvDesc(1, 64, 56, 56, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1, 0, 0)
    assert pkg.lib().escoin_padded_len(C.byref(d)) == 207994
    d = pkg.ConvDesc(1, 96, 27, 27, 256, 5, 5, 2, 2, 1, 1, 1, 1, 2, 1, 0)
    assert pkg.lib().escoin_padded_len(C.byref(d)) == 80798       # AlexNet conv2


@pytest.mark.parametrize("bad", [
    dict(C=0), dict(M=0), dict(group=3), dict(stride_h=0), dict(dil_w=0), dict(pad_h=-1),
    dict(KH=9, H=4, pad_h=0),          # empty output
    dict(N=0),
])
def test_plan_create_rejects_bad_geometry(pkg, bad):
    base = dict(N=2, C=4, H=8, W=8, M=4, KH=3, KW=3, pad_h=1, pad_w=1, stride_h=1, stride_w=1,
                dil_h=1, dil_w=1, group=1, has_bias=1, fuse_relu=0)
    base.update(bad)
    desc = pkg.ConvDesc(**base)
    h = C.c_void_p()
    rc = pkg.lib().escoin_plan_create(C.byref(desc), C.byref(h))
    assert rc == -1 and not h.value
    assert pkg.lib().escoin_last_error()      # a message is recorded


def test_plan_lifecycle_and_options_on_host(pkg):
    desc = pkg.ConvDesc(4, 6, 8, 8, 6, 3, 3, 1, 1, 1, 1, 1, 1, 3, 1, 0)
    plan = pkg.Plan(desc)
    assert plan.out_hw == (8, 8)
    assert plan.nnz() == 0 and plan.workspace_bytes == 0
    plan.set_option("kernel", pkg.KERNEL_GENERIC)
    plan.set_option("conv_mode", pkg.CONV_MODE_SCONV)
    plan.set_option("conv_mode", pkg.CONV_MODE_LOWERED_SPARSE)   # the im2col + csrmm comparator
    plan.set_option("conv_mode", pkg.CONV_MODE_SCONV_PAR)
    plan.set_option("conv_mode", pkg.CONV_MODE_LOWERED_GEMM)     # -conv_mode 0: dense MFMA kernel
    plan.set_option("conv_mode", pkg.CONV_MODE_SCONV_PAR)
    for bad in (-1, 4):
        with pytest.raises(pkg.EscoinError):
            plan.set_option("conv_mode", bad)
    plan.set_option("tiling_batch", 256)
    plan.set_option("dense_threshold_pct", 30)
    plan.set_option("dense_gate", 1)
    for v in (-1, 0, 1):
        plan.set_option("stream_stores", v)
    with pytest.raises(pkg.EscoinError):
        plan.set_option("stream_stores", 2)
    with pytest.raises(pkg.EscoinError):
        plan.set_option("tiling_batch", -5)
    with pytest.raises(pkg.EscoinError):
        plan.set_option("dense_threshold_pct", 101)
    with pytest.raises(pkg.EscoinError):
        plan.set_option("no_such_option", 1)
    with pytest.raises(pkg.EscoinError):
        plan.nnz(group=7)
    plan.close()
    plan.close()                                     # idempotent


def test_forward_before_align_is_a_state_error(pkg):
    desc = pkg.ConvDesc(1, 2, 4, 4, 2, 3, 3, 1, 1, 1, 1, 1, 1, 1, 0, 0)
    plan = pkg.Plan(desc)
    rc = pkg.lib().escoin_forward(plan._h, C.c_void_p(16), None, C.c_void_p(16), 1, None)
    assert rc == -4
    rc = pkg.lib().escoin_forward(plan._h, None, None, C.c_void_p(16), 1, None)
    assert rc == -1


def test_gpu_entry_points_fail_loudly_without_device_and_cpu_mode_is_explicit(pkg, synth, oracle):
    """No SILENT fallback: without a HIP device the GPU entry points return ESCOIN_ENODEVICE and compute nothing.
    Caffe::CPU mode is a separate, explicit pair of entry points of the same library (tests/test_cpu_mode.py) -- a
    plan whose GPU align failed is still not a GPU plan."""
    if not _no_gpu(pkg):
        pytest.skip("a HIP device is visible")
    s = synth.lenet_conv2(N=1)[0]
    w = synth.pruned_weights(s, 1)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    with pytest.raises(pkg.EscoinError) as e:
        plan.weight_align(w)
    assert "no HIP device" in str(e.value)
    with pytest.raises(pkg.EscoinError) as e:
        plan.weight_align(w.astype(np.float64))
    assert "no HIP device" in str(e.value)
    rc = pkg.lib().escoin_forward(plan._h, C.c_void_p(16), None, C.c_void_p(16), 1, None)
    assert rc == -4                                   # still not aligned for the device: forward refuses
    # math_functions-level entry points validate their arguments on the host
    assert pkg.lib().escoin_gpu_stretch(None, None, 0, 1, 1, 0, 0, 1, 1, None) == -1
    # the explicit CPU mode works on the same machine, same library
    plan.weight_align_cpu(w)
    x, b = synth.activations(s, 2), synth.bias_vector(s, 3)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
    assert np.array_equal(plan.forward_cpu(x, b), oracle.conv_forward(g, x, w, b, gate=False))
    assert pkg.cpu_kernel_name().startswith("escoin_cpu_sconv_")


def test_set_csr_validates_on_host(pkg):
    desc = pkg.ConvDesc(1, 2, 4, 4, 2, 3, 3, 1, 1, 1, 1, 1, 1, 1, 0, 0)
    plan = pkg.Plan(desc)
    rp = np.array([0, 2, 1], np.int32)               # not monotone / does not match nnz
    ci = np.array([0, 5], np.int32)
    va = np.array([1, 2], np.float32)
    with pytest.raises(pkg.EscoinError):
        plan.set_csr(rp, ci, va, [2])
    with pytest.raises(pkg.EscoinError):
        plan.set_csr(np.array([0, 1, 2], np.int32), np.array([0, 99], np.int32), va, [2])
    # columns within a row: strictly ascending, as caffe_cpu_sparse_dense2csr's scan leaves them
    # (math_functions.cpp:92-105); unsorted or duplicated columns are refused before any device work
    for bad in ([5, 0], [5, 5]):
        with pytest.raises(pkg.EscoinError) as e:
            plan.set_csr(np.array([0, 2, 2], np.int32), np.array(bad, np.int32), va, [2])
        assert "ascending" in str(e.value)


def test_package_is_importable_under_alias():
    import __graft_entry__ as ge
    p = ge.load_package()
    assert p.__name__ == "caffe_escoin_amd" and hasattr(p, "synth") and hasattr(p, "Plan")


def test_product_library_reads_no_experiment_switches():
    """The product build reads ESCOIN_VERBOSE (diagnostics) and TMPDIR and nothing else: every tuning switch is a
    compile-time constant there and the wrong-result switches (ESCOIN_DBG, ESCOIN_JIT_ABL, ESCOIN_DENSE_ABL) exist in
    the -DESCOIN_ABLATIONS flavour only (csrc/knobs.h, tools/mkabl.sh).  The names must not even
    be in the binary: a leftover environment variable cannot change a result."""
    import re
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_path = os.path.join(ROOT, "caffe-escoin_amd", "libescoin_hip.so")
    assert os.path.exists(lib_path)
    out = subprocess.run(["strings", "-n", "6", lib_path], stdout=subprocess.PIPE, check=True).stdout.decode()
    names = set(re.findall(r"\bESCOIN_[A-Z0-9_]+\b", out))
    assert names <= {"ESCOIN_VERBOSE"}, sorted(names - {"ESCOIN_VERBOSE"})
    assert not re.search(r"ESCOIN_JIT_ABL|ESCOIN_DBG|ESCOIN_DENSE_ABL|ESCOIN_PROF", out)
    # the sources agree: getenv appears only in knobs.h (the non-product flavours), for ESCOIN_VERBOSE and for TMPDIR
    csrc = os.path.join(ROOT, "caffe-escoin_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".cpp", ".h")) or fn == "knobs.h":
            continue
        for line in open(os.path.join(csrc, fn)):
            for m in re.finditer(r'getenv\("([A-Z_0-9]+)"\)', line):
                assert m.group(1) in ("ESCOIN_VERBOSE", "TMPDIR"), (fn, line.strip())
    # and no A/B artefact sits in the package directory (it would ship with every push to a GPU box)
    extra = [f for f in os.listdir(os.path.join(ROOT, "caffe-escoin_amd")) if f.endswith(".so") and f != "libescoin_hip.so"]
    assert not extra, extra
