#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run in the build container where /root/reference is mounted.

Every expected output here is produced by oracle/_ref: the REFERENCE's own CPU kernel
(include/caffe/util/sconv.hpp:594-678 caffe_cpu_sconv_default<false>, compiled in place by
oracle/Makefile) driven over whole batches by oracle/ref_driver.cpp.  The fixtures are data only:
inputs, geometry, the CSR the reference's WeightAlign would hold (rowptr, stretched colidx,
values) and the expected top blob.  The cases follow the reference's own conv tests
(src/caffe/test/test_convolution_layer.cpp: k3 s2 on 2x3x6x4 :231-265, dilation 2 on 2x3x8x7
:267-309, 1x1 :443-468, group 3 :470-496) plus scaled-down layers of BASELINE.json's configs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
oracle = ge.load_oracle()
synth = pkg.synth
S = synth.shape

CASES = [
    # name, shape, seed
    S("ref_simple_k3s2", 2, 3, 6, 4, 4, 3, stride=2, sparsity=0.4),
    S("ref_dilated_k3d2", 2, 3, 8, 7, 4, 3, dil=2, sparsity=0.4),
    S("ref_1x1", 2, 3, 6, 4, 4, 1, sparsity=0.25),
    S("ref_group3", 2, 6, 6, 4, 3, 3, group=3, sparsity=0.3),
    S("lenet_conv2_n2", 2, 20, 12, 12, 50, 5, sparsity=0.5),
    S("alex_like_g2_k5p2", 1, 16, 27, 27, 32, 5, pad=2, group=2, sparsity=0.8),
    S("alex_like_k3p1_w13", 2, 48, 13, 13, 64, 3, pad=1, sparsity=0.8),
    S("res2_like_nobias", 1, 16, 56, 56, 16, 3, pad=1, bias=False, sparsity=0.9),
    S("res4_like_nobias", 2, 64, 14, 14, 64, 3, pad=1, bias=False, sparsity=0.9),
    S("res5_like_nobias", 3, 256, 7, 7, 128, 3, pad=1, bias=False, sparsity=0.9),
    S("googlenet_like_1x1", 1, 96, 28, 28, 32, 1, sparsity=0.95),
    S("nonsquare_k3x5_s1x2_p1x2", 2, 5, 9, 11, 6, 3, KW=5, pad=1, pad_w=2, stride=1, stride_w=2,
      sparsity=0.5),
    S("empty_rows_k3p1", 1, 4, 5, 5, 8, 3, pad=1, sparsity=0.97),
    # (round 6) padding in one direction only, pad_h == 0 < pad_w: the geometry class whose last row's right padding lies
    # behind the reference's own allocation (base_conv_layer.cpp:71).  The reference kernel runs here on a buffer with the
    # pad_w zero floats of slack the oracle defines (oracle_padded_len), i.e. its arithmetic with the padding the layer
    # specifies; and the mirror case pad_w == 0 < pad_h, which needs no slack.
    S("padw_only_k1x3", 2, 4, 6, 9, 5, 1, KW=3, pad=0, pad_w=1, sparsity=0.5),
    S("padw_only_k3x5_p0x2", 2, 3, 7, 10, 4, 3, KW=5, pad=0, pad_w=2, sparsity=0.6),
    S("padh_only_k3x1", 2, 4, 6, 9, 5, 3, KW=1, pad=1, pad_w=0, sparsity=0.5),
]


def main():
    if not oracle.have_ref():
        oracle.build()
    if not oracle.have_ref():
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    only_new = "--only-new" in sys.argv      # (keeps the committed files byte-identical: np.savez_compressed is not)
    for k, s in enumerate(CASES):
        seed = 1000 + 17 * k
        if only_new and os.path.exists(os.path.join(HERE, s.name + ".npz")):
            continue
        w = synth.pruned_weights(s, seed)
        b = synth.bias_vector(s, seed + 1)
        x = synth.activations(s, seed + 2)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                        s.dil_h, s.dil_w, s.group)
        top = oracle.ref_conv_forward(g, x, w, b)
        mg, cg = s.M // s.group, s.C // s.group
        rps, cis, vas = [], [], []
        for grp in range(s.group):
            rp, ci, va = oracle.dense2csr(w[grp * mg:(grp + 1) * mg].reshape(mg, cg * s.KH * s.KW))
            cis.append(oracle.stretch(rp, ci, s.KH, s.KW, s.H, s.W, s.pad_h, s.pad_w))
            rps.append(rp)
            vas.append(va)
        meta = np.array([s.N, s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h,
                         s.stride_w, s.dil_h, s.dil_w, s.group, int(s.bias)], np.int32)
        out = dict(meta=meta, x=x, w=w, top=top, rowptr=np.concatenate(rps),
                   colidx_stretched=np.concatenate(cis), values=np.concatenate(vas))
        if b is not None:
            out["bias"] = b
        path = os.path.join(HERE, s.name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s top%s nnz=%d  %.1f KB" % (s.name, top.shape, len(out["values"]),
                                               os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
