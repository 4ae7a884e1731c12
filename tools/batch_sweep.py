#!/usr/bin/env python3
"""GPU box: us per launch of KERNEL_AUTO (and of the generic kernel) over the batch size, per layer shape -- the reference's
SCONV mode runs image by image (conv_layer.cu:16-26), SCONV_PAR the whole batch; a drop-in sees every batch size.
    python tools/batch_sweep.py [res4 res5 goog25 ...] > profiles/<tag>_batch_sweep.md"""
import importlib
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")
synth = pkg.synth


def shapes_at(n):
    d = {s.name.split("_")[0]: s for s in synth.resnet50_3x3(N=n)}
    d.update({"alex%d" % (i + 2): s for i, s in enumerate(synth.alexnet(N=n))})
    d.update({"goog%d" % i: s for i, s in enumerate(synth.googlenet_1x1(N=n))})
    return d


def time_plan(plan, s, dev, launches):
    x = torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1
    y = torch.empty((s.N, s.M) + tuple(plan.out_hw), device=dev)
    for _ in range(10):
        plan.forward(x, None, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        a.record()
        for _ in range(launches):
            plan.forward(x, None, y)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / launches * 1e3)
    return best


def main():
    which = sys.argv[1:] or ["res2", "res3", "res4", "res5", "goog5", "goog25", "goog33", "alex3"]
    dev = torch.device("cuda:0")
    batches = [1, 2, 4, 8, 16, 32, 64, 128, 256]
    print("us per launch (images/s in thousands) of `KERNEL_AUTO`; in brackets the kernel it ran and the generic kernel's us\n")
    print("| layer | " + " | ".join("N=%d" % n for n in batches) + " |")
    print("|---|" + "---|" * len(batches))
    for name in which:
        cells = []
        for n in batches:
            s = shapes_at(n)[name]
            w = synth.pruned_weights(s, 1)
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_AUTO)
            plan.weight_align(w)
            us = time_plan(plan, s, dev, 50)
            kn = plan.kernel_name
            kn = "code" if "jit" in kn else "stream" if "tiled" in kn else "dense" if "dense" in kn else "generic" if "generic" in kn else kn
            gen = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_GENERIC)
            gen.weight_align(w)
            gus = time_plan(gen, s, dev, 10 if n >= 64 else 30)
            cells.append("%.1f (%.1f k) [%s; %.0f]" % (us, n / us * 1e3, kn, gus))
            plan.close(); gen.close()
        print("| %s | " % name + " | ".join(cells) + " |", flush=True)


if __name__ == "__main__":
    main()
