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


def straight_line(which, batches, dev):
    """Arbitrary batches against the straight line through N = 128 and N = 256 (images/s as a fraction of what the line
    through those two points gives at that batch: time(N) = t128 + (N - 128) * (t256 - t128) / 128).  HBM-cold: four
    rotating bottom / top pairs where they fit."""
    print("us per launch of `KERNEL_AUTO` and images/s relative to the straight line through N = 128 and N = 256 (1.00 = on the line; "
          "< 0.90 = a cliff); in brackets the tiling's images per tile x tiles per workgroup round\n")
    print("| layer | " + " | ".join("N=%d" % n for n in batches) + " |")
    print("|---|" + "---|" * len(batches))

    def cold_time(name, n):
        s = shapes_at(n)[name]
        w = synth.pruned_weights(s, 1)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_AUTO)
        plan.weight_align(w)
        oh, ow = plan.out_hw
        per_pair = 4 * n * (s.C * s.H * s.W + s.M * oh * ow)
        nb = max(1, min(4, int(3e9 // per_pair)))
        xs = [torch.rand((n, s.C, s.H, s.W), device=dev) * 2 - 1 for _ in range(nb)]
        ys = [torch.empty((n, s.M, oh, ow), device=dev) for _ in range(nb)]
        for i in range(12):
            plan.forward(xs[i % nb], None, ys[i % nb])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e30
        for _ in range(3):
            a.record()
            for i in range(40):
                plan.forward(xs[i % nb], None, ys[i % nb])
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 40 * 1e3)
        info = plan.tiling_info
        plan.close()
        return best, info

    import re
    for name in which:
        t128, _ = cold_time(name, 128)
        t256, _ = cold_time(name, 256)
        cells = []
        for n in batches:
            us, info = cold_time(name, n)
            line_us = t128 + (n - 128) * (t256 - t128) / 128.0
            m = {k: int(v) for k, v in re.findall(r"\b(nseg|bands|columns|band)=(\d+)", info)}
            tiles = n * m.get("bands", 1) if m.get("band") else -(-n // max(1, m.get("nseg", 1)))
            cells.append("%.1f (%.2f) [%dx%d t, %d col]" % (us, line_us / us, max(1, m.get("nseg", 1)), tiles, m.get("columns", 1)))
        print("| %s (128: %.1f, 256: %.1f) | " % (name, t128, t256) + " | ".join(cells) + " |", flush=True)


def main():
    args = sys.argv[1:]
    batches = [1, 2, 4, 8, 16, 32, 64, 128, 256]
    line = False
    if "--batches" in args:          # e.g. --batches 96,100,192,200,250,255,257,293,300,341,384,512
        i = args.index("--batches")
        batches = [int(v) for v in args[i + 1].split(",")]
        del args[i:i + 2]
        line = True
    which = args or ["res2", "res3", "res4", "res5", "goog5", "goog25", "goog33", "alex3"]
    dev = torch.device("cuda:0")
    if line:
        return straight_line(which, batches, dev)
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
