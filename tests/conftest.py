import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = ge.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def oracle():
    o = ge.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


class Golden(object):
    """One committed fixture (tests/golden/make_golden.py made it with oracle/_ref)."""

    def __init__(self, path):
        z = np.load(path)
        self.name = os.path.basename(path)[:-4]
        m = [int(v) for v in z["meta"]]
        (self.N, self.C, self.H, self.W, self.M, self.KH, self.KW, self.pad_h, self.pad_w,
         self.stride_h, self.stride_w, self.dil_h, self.dil_w, self.group, has_bias) = m
        self.x, self.w, self.top = z["x"], z["w"], z["top"]
        self.bias = z["bias"] if has_bias else None
        self.rowptr, self.colidx_stretched, self.values = z["rowptr"], z["colidx_stretched"], z["values"]

    def geom(self, oracle):
        return oracle.geom(self.C, self.H, self.W, self.M, self.KH, self.KW, self.pad_h, self.pad_w,
                           self.stride_h, self.stride_w, self.dil_h, self.dil_w, self.group)

    def desc(self, pkg, fuse_relu=False, N=None):
        return pkg.ConvDesc(self.N if N is None else N, self.C, self.H, self.W, self.M, self.KH,
                            self.KW, self.pad_h, self.pad_w, self.stride_h, self.stride_w,
                            self.dil_h, self.dil_w, self.group, int(self.bias is not None),
                            int(fuse_relu))


def golden_params():
    return [pytest.param(p, id=os.path.basename(p)[:-4]) for p in GOLDEN_FILES]


def rel_err(got, want):
    """max|a-b| / max(1e-6, max|ref|) over the tensor."""
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max() /
                 max(1e-6, float(np.abs(want).max())))


def naive_conv(x, w, bias, s):
    """Independent dense direct convolution in float64, in the style of the reference's own
    test helper caffe_conv() (src/caffe/test/test_convolution_layer.cpp:19-140): groups,
    stride, pad, dilation; no CSR, no padded layout."""
    N, C, H, W = x.shape
    M, cg, KH, KW = w.shape
    grp = C // cg
    mg = M // grp
    oh = (H + 2 * s.pad_h - (s.dil_h * (KH - 1) + 1)) // s.stride_h + 1
    ow = (W + 2 * s.pad_w - (s.dil_w * (KW - 1) + 1)) // s.stride_w + 1
    xp = np.zeros((N, C, H + 2 * s.pad_h, W + 2 * s.pad_w), np.float64)
    xp[:, :, s.pad_h:s.pad_h + H, s.pad_w:s.pad_w + W] = x
    out = np.zeros((N, M, oh, ow), np.float64)
    for g in range(grp):
        for kr in range(KH):
            for kc in range(KW):
                patch = xp[:, g * cg:(g + 1) * cg,
                           kr * s.dil_h:kr * s.dil_h + (oh - 1) * s.stride_h + 1:s.stride_h,
                           kc * s.dil_w:kc * s.dil_w + (ow - 1) * s.stride_w + 1:s.stride_w]
                wk = w[g * mg:(g + 1) * mg, :, kr, kc].astype(np.float64)
                out[:, g * mg:(g + 1) * mg] += np.einsum("mc,nchw->nmhw", wk, patch)
    if bias is not None:
        out += bias.astype(np.float64)[None, :, None, None]
    return out
