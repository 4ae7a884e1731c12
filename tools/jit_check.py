#!/usr/bin/env python3
"""GPU box: one BASELINE layer shape through a chosen kernel against the oracle, with the error
localised (which images / channels / rows are off).   python tools/jit_check.py googlenet 0 3 [jit|tiled|auto]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package(); oracle = ge.load_oracle(); synth = pkg.synth
which, idx, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kern = {"jit": pkg.KERNEL_JIT, "tiled": pkg.KERNEL_TILED, "auto": pkg.KERNEL_AUTO}[sys.argv[4] if len(sys.argv) > 4 else "jit"]
sets = {"lenet": synth.lenet_conv2, "alexnet": synth.alexnet, "resnet50": synth.resnet50_3x3, "googlenet": synth.googlenet_1x1}
s = sets[which](N=N)[idx]
w, b, x = synth.pruned_weights(s, 100), synth.bias_vector(s, 200), synth.activations(s, 300)
g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
want = oracle.conv_forward(g, x, w, b, gate=False, threads=4)
plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kern)
plan.weight_align(w)
dev = torch.device("cuda:0")
got = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev) if b is not None else None).cpu().numpy()
err = np.abs(got - want)
print(s.name, plan.kernel_name, "rel_err %.3g" % (err.max() / max(1e-6, np.abs(want).max())))
bad = err > 1e-3
if bad.any():
    print(" bad fraction %.4f" % bad.mean())
    print(" per image  :", [round(float(bad[n].mean()), 3) for n in range(min(N, 8))])
    print(" per channel:", [round(float(bad[:, m].mean()), 2) for m in range(min(s.M, 32))])
    rows = bad.any(axis=(0, 1, 3))
    print(" rows bad   :", np.nonzero(rows)[0].tolist()[:60])
    cols = bad.any(axis=(0, 1, 2))
    print(" cols bad   :", np.nonzero(cols)[0].tolist()[:60])
