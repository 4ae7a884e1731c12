"""Deterministic synthetic tensors and the layer-shape tables of BASELINE.json's configs.

The reference ships no pruned .caffemodel (its run.sh:14 points outside the tree), so
every workload here is synthetic: activations uniform(-1,1), weights uniform(-1,1) with
EXACT-COUNT unstructured pruning per group (so the density never crosses the reference's
dense-fallback gates by chance: base_conv_layer.cpp:574 CPU >0.5, :750 GPU >0.2), bias
uniform(-0.1,0.1) where the prototxt has bias_term.

Everything is a pure function of (seed, index) through a splitmix64-style counter hash,
so numpy on the host, any rank of a multi-GPU run and the golden-fixture generator all
see the same numbers.
"""
from collections import namedtuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    """splitmix64 finaliser, vectorised over uint64."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(seed, start, count):
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix(idx + _mix(np.array([seed], dtype=np.uint64))[0])


def uniform(seed, count, lo=-1.0, hi=1.0, start=0):
    """count float32 values in [lo, hi): 24 random mantissa bits each."""
    h = hash_u64(seed, start, count)
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32)


ConvShape = namedtuple(
    "ConvShape", "name N C H W M KH KW pad_h pad_w stride_h stride_w dil_h dil_w group bias "
                 "sparsity count")
ConvShape.__new__.__defaults__ = (1,)


def shape(name, N, C, H, W, M, K, pad=0, stride=1, dil=1, group=1, bias=True, sparsity=0.9,
          count=1, KW=None, pad_w=None, stride_w=None, dil_w=None):
    return ConvShape(name, N, C, H, W, M, K, K if KW is None else KW, pad,
                     pad if pad_w is None else pad_w, stride,
                     stride if stride_w is None else stride_w, dil,
                     dil if dil_w is None else dil_w, group, bias, sparsity, count)


def out_hw(s):
    oh = (s.H + 2 * s.pad_h - (s.dil_h * (s.KH - 1) + 1)) // s.stride_h + 1
    ow = (s.W + 2 * s.pad_w - (s.dil_w * (s.KW - 1) + 1)) // s.stride_w + 1
    return oh, ow


def pruned_weights(s, seed):
    """(M, C/g, KH, KW) float32 with exactly round(sparsity*count) zeros per group."""
    cg = s.C // s.group
    per_group = (s.M // s.group) * cg * s.KH * s.KW
    w = uniform(seed, per_group * s.group).copy()
    # never let a kept weight be exactly zero (dense->CSR keeps != 0 only)
    w[w == 0.0] = np.float32(0.5)
    for g in range(s.group):
        nzero = int(round(s.sparsity * per_group))
        rank = np.argsort(hash_u64(seed ^ 0x5EED5EED, g * per_group, per_group), kind="stable")
        w[g * per_group + rank[:nzero]] = 0.0
    return w.reshape(s.M, cg, s.KH, s.KW)


def bias_vector(s, seed):
    return uniform(seed ^ 0xB1A5, s.M, -0.1, 0.1) if s.bias else None


def activations(s, seed, n0=0, n=None):
    """Images [n0, n0+n) of the batch; image k depends only on (seed, k)."""
    n = s.N if n is None else n
    per = s.C * s.H * s.W
    return uniform(seed ^ 0xAC71, n * per, start=n0 * per).reshape(n, s.C, s.H, s.W)


# ---------------------------------------------------------------------------------
# Layer tables (shapes from the reference's prototxts:
# models/lenet5/train_test.prototxt:70-92, models/bvlc_reference_caffenet/
# test_sconv.prototxt:112-258, models/resnet/test_sconv.prototxt,
# models/bvlc_googlenet/test_sconv.prototxt).
# ---------------------------------------------------------------------------------

def lenet_conv2(N=64, sparsity=0.5):
    return [shape("lenet_conv2", N, 20, 12, 12, 50, 5, sparsity=sparsity)]


def alexnet(N=128, sparsity=0.8):
    return [
        shape("alex_conv2", N, 96, 27, 27, 256, 5, pad=2, group=2, sparsity=sparsity),
        shape("alex_conv3", N, 256, 13, 13, 384, 3, pad=1, sparsity=sparsity),
        shape("alex_conv4", N, 384, 13, 13, 384, 3, pad=1, group=2, sparsity=sparsity),
        shape("alex_conv5", N, 384, 13, 13, 256, 3, pad=1, group=2, sparsity=sparsity),
    ]


def resnet50_3x3(N=256, sparsity=0.9):
    """The 16 branch2b 3x3 convolutions of ResNet-50 (no bias_term), as 4 distinct shapes."""
    return [
        shape("res2_branch2b", N, 64, 56, 56, 64, 3, pad=1, bias=False, sparsity=sparsity, count=3),
        shape("res3_branch2b", N, 128, 28, 28, 128, 3, pad=1, bias=False, sparsity=sparsity, count=4),
        shape("res4_branch2b", N, 256, 14, 14, 256, 3, pad=1, bias=False, sparsity=sparsity, count=6),
        shape("res5_branch2b", N, 512, 7, 7, 512, 3, pad=1, bias=False, sparsity=sparsity, count=3),
    ]


_GOOGLENET_1X1 = [
    ("conv2/3x3_reduce", 64, 56, [64]),
    ("inception_3a", 192, 28, [64, 96, 16, 32]),
    ("inception_3b", 256, 28, [128, 128, 32, 64]),
    ("inception_4a", 480, 14, [192, 96, 16, 64]),
    ("inception_4b", 512, 14, [160, 112, 24, 64]),
    ("inception_4c", 512, 14, [128, 128, 24, 64]),
    ("inception_4d", 512, 14, [112, 144, 32, 64]),
    ("inception_4e", 528, 14, [256, 160, 32, 128]),
    ("inception_5a", 832, 7, [256, 160, 32, 128]),
    ("inception_5b", 832, 7, [384, 192, 48, 128]),
    ("loss1/conv", 512, 4, [128]),
    ("loss2/conv", 528, 4, [128]),
]


def googlenet_1x1(N=256, sparsity=0.95):
    out = []
    for name, cin, hw, couts in _GOOGLENET_1X1:
        for i, m in enumerate(couts):
            out.append(shape("%s/1x1_%d" % (name, i), N, cin, hw, hw, m, 1, sparsity=sparsity))
    return out


def nnz_of(s):
    per_group = (s.M // s.group) * (s.C // s.group) * s.KH * s.KW
    return s.group * (per_group - int(round(s.sparsity * per_group)))


def algorithmic_bytes(s, n=None):
    """input once + output once + CSR once (+ bias), bytes per layer call."""
    n = s.N if n is None else n
    oh, ow = out_hw(s)
    return (4 * n * s.C * s.H * s.W + 4 * n * s.M * oh * ow + 8 * nnz_of(s) +
            4 * (s.M // s.group + 1) * s.group + (4 * s.M if s.bias else 0))


def flops(s, n=None):
    n = s.N if n is None else n
    oh, ow = out_hw(s)
    return 2 * n * oh * ow * nnz_of(s)
