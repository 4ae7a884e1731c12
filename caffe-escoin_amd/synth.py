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


SPARSITY_DISTS = ("uniform", "channel", "zero_inputs", "filters_tail")


def _row_counts(weights, total, cap):
    """Integer nonzero counts per row, proportional to `weights`, summing to `total` exactly, none above `cap`
    (largest remainders first; what a capped row cannot take goes to the others)."""
    w = np.asarray(weights, np.float64).copy()
    counts = np.zeros(len(w), np.int64)
    left = int(total)
    for _ in range(64):
        if left <= 0 or w.sum() <= 0:
            break
        share = w / w.sum() * left
        add = np.minimum(np.floor(share).astype(np.int64), cap - counts)
        counts += add
        left -= int(add.sum())
        w = np.where(counts >= cap, 0.0, w)
        if int(add.sum()) == 0:
            # less than one per open row is left: hand the rest out one by one, largest share first
            order = np.argsort(-(share - np.floor(share)), kind="stable")
            for i in order:
                if left <= 0:
                    break
                if counts[i] < cap and w[i] > 0:
                    counts[i] += 1
                    left -= 1
            if left > 0:
                for i in np.argsort(-counts, kind="stable")[::-1]:
                    take = min(left, cap - counts[i])
                    counts[i] += take
                    left -= take
                    if left <= 0:
                        break
            break
    return counts


def pruned_weights(s, seed, dist="uniform"):
    """(M, C/g, KH, KW) float32 with exactly round(sparsity*count) zeros per group.

    dist -- how the nonzeros lie (the total per group is the same for all, so the algorithmic bytes and flops of a
    layer do not depend on it):
      "uniform"       unstructured random pruning (the default; every BASELINE config)
      "channel"       per-OUTPUT-channel density drawn from U(0, 2 d) -- what magnitude pruning leaves (the nets the
                      reference runs are SkimCaffe-pruned, run.sh:14: channels differ widely in density)
      "zero_inputs"   20 % of the INPUT channels are entirely zero; the others carry the same total
      "filters_tail"  10 % of the filters (output channels) are entirely zero, 5 % hold four times the mean
                      density (a heavy tail), the rest share what is left
    """
    cg = s.C // s.group
    mg = s.M // s.group
    kk = s.KH * s.KW
    per_group = mg * cg * kk
    w = uniform(seed, per_group * s.group).copy()
    # never let a kept weight be exactly zero (dense->CSR keeps != 0 only)
    w[w == 0.0] = np.float32(0.5)
    if dist == "uniform":
        for g in range(s.group):
            nzero = int(round(s.sparsity * per_group))
            rank = np.argsort(hash_u64(seed ^ 0x5EED5EED, g * per_group, per_group), kind="stable")
            w[g * per_group + rank[:nzero]] = 0.0
        return w.reshape(s.M, cg, s.KH, s.KW)
    if dist not in SPARSITY_DISTS:
        raise ValueError("unknown sparsity distribution %r" % (dist,))
    d = 1.0 - s.sparsity
    for g in range(s.group):
        keep_total = per_group - int(round(s.sparsity * per_group))
        u = (hash_u64(seed ^ 0xD157, g * (mg + cg), mg + cg) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
        u_m, u_c = u[:mg], u[mg:]
        in_ok = np.ones(cg, bool)
        if dist == "channel":
            row_w = 2.0 * d * u_m + 1e-9
        elif dist == "zero_inputs":
            row_w = np.ones(mg)
            n_dead = int(round(0.2 * cg)) if cg >= 5 else 0
            in_ok[np.argsort(u_c, kind="stable")[:n_dead]] = False
        else:       # filters_tail
            order = np.argsort(u_m, kind="stable")
            n_dead = int(round(0.1 * mg)) if mg >= 10 else 0
            n_tail = max(1, int(round(0.05 * mg))) if mg >= 4 else 0
            row_w = np.ones(mg)
            row_w[order[:n_dead]] = 0.0
            tail = order[n_dead:n_dead + n_tail]
            # the tail rows at 4 d each; the others share the rest evenly
            rest = max(1, mg - n_dead - n_tail)
            tail_share = min(4.0 * d, 1.0) * cg * kk
            other_share = max(0.0, (keep_total - tail_share * n_tail) / rest)
            row_w[tail] = tail_share / max(other_share, 1e-9)
        cap = int(in_ok.sum()) * kk
        counts = _row_counts(row_w, min(keep_total, cap * int((row_w > 0).sum())), cap)
        allowed = np.repeat(in_ok, kk)            # positions of a row a nonzero may take
        pos_all = np.nonzero(allowed)[0]
        for m in range(mg):
            base = g * per_group + m * cg * kk
            h = hash_u64(seed ^ 0x5EED5EED, base, cg * kk)
            keep = pos_all[np.argsort(h[pos_all], kind="stable")[:int(counts[m])]]
            row = w[base:base + cg * kk]
            mask = np.zeros(cg * kk, bool)
            mask[keep] = True
            row[~mask] = 0.0
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
