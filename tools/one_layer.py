#!/usr/bin/env python3
"""Times ONE ResNet-50 3x3 layer shape back to back (no other kernel in between): separates a
kernel's own speed from what its neighbours in the bench step do to the chip's clocks.
    ESCOIN_LIB=... python tools/one_layer.py res5 [launches]
"""
import importlib
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")
synth = pkg.synth


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "res5"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    sp = float(os.environ.get("ONE_LAYER_SPARSITY", "0")) or None
    shapes = {s.name.split("_")[0]: s for s in synth.resnet50_3x3(N=256, sparsity=sp or 0.9)}
    shapes.update({"alex" + str(i + 2): s for i, s in enumerate(synth.alexnet(N=128, sparsity=sp or 0.8))})
    shapes.update({"goog%d" % i: s for i, s in enumerate(synth.googlenet_1x1(N=256, sparsity=sp or 0.95))})
    s = shapes[which]
    kern = {"auto": pkg.KERNEL_AUTO, "jit": pkg.KERNEL_JIT, "tiled": pkg.KERNEL_TILED}[os.environ.get("ONE_LAYER_KERNEL", "auto")]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kern)
    plan.weight_align(synth.pruned_weights(s, 1))
    dev = torch.device("cuda:0")
    # ONE_LAYER_BUFS=k: rotate k input / output pairs (more than the 256 MB Infinity Cache holds: HBM-cold)
    nb = int(os.environ.get("ONE_LAYER_BUFS", "1"))
    xs = [torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1 for _ in range(nb)]
    ys = [torch.empty((s.N, s.M) + tuple(plan.out_hw), device=dev) for _ in range(nb)]
    x = xs[0]
    bias = torch.zeros(s.M, device=dev) if s.bias else None
    for _ in range(20):
        y = plan.forward(x, bias)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        a.record()
        for i in range(n):
            y = plan.forward(xs[i % nb], bias, ys[i % nb])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    if os.environ.get("ONE_LAYER_GAPS"):
        # the same launches with the chip idle in between (host sleep): is the back-to-back rate
        # limited by sustained power / clocks?
        import time
        gaps = []
        for _ in range(40):
            time.sleep(float(os.environ["ONE_LAYER_GAPS"]) * 1e-3)
            a.record()
            y = plan.forward(x, None)
            b.record()
            torch.cuda.synchronize()
            gaps.append(a.elapsed_time(b) * 1e3)
        gaps.sort()
        print("   single launches %s ms apart: median %.1f us, min %.1f, max %.1f" %
              (os.environ["ONE_LAYER_GAPS"], gaps[len(gaps) // 2], gaps[0], gaps[-1]))
    print("%s %-40s %s us per launch (5 x %d launches): %s" % (os.path.basename(os.environ.get("ESCOIN_LIB", "default")), plan.kernel_name, which, n, " ".join("%.1f" % t for t in ts)))


if __name__ == "__main__":
    main()
