#!/usr/bin/env python3
"""GPU box: the data KERNEL_AUTO's small-launch rule (escoin_capi.hip) is fitted to.  For 1 .. 32 images of every distinct
GoogLeNet 1x1 shape (and the four ResNet 3x3 shapes for reference): generated code and the generic kernel, both FORCED
(plan option "kernel"), 60 launches, best of three, on one reused bottom / top pair (what an image-by-image caller --
the reference's SCONV mode, conv_layer.cu:16-26 -- does).  One line of JSON per cell with the features the rule may use.
    python tools/small_launch_fit.py > gpurun_out/<tag>/small_launch_fit.jsonl"""
import importlib
import json
import os
import re
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")
synth = pkg.synth


def time_plan(plan, x, y, launches):
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
    dev = torch.device("cuda:0")
    batches = [1, 2, 4, 8, 16, 32]
    seen = set()
    for n in batches:
        shapes = list(synth.googlenet_1x1(N=n)) + list(synth.resnet50_3x3(N=n))
        for s in shapes:
            key = (s.C, s.H, s.M, s.KH, n)
            if key in seen:
                continue
            seen.add(key)
            w = synth.pruned_weights(s, 1)
            x = torch.rand((n, s.C, s.H, s.W), device=dev) * 2 - 1
            row = {"layer": s.name, "N": n, "C": s.C, "HW": s.H, "M": s.M, "K": s.KH, "mflop": synth.flops(s, n) * 1e-6}
            for tag, kern in (("code", pkg.KERNEL_JIT), ("generic", pkg.KERNEL_GENERIC), ("auto", pkg.KERNEL_AUTO)):
                plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kern)
                plan.weight_align(w)
                y = torch.empty((n, s.M) + tuple(plan.out_hw), device=dev)
                row[tag + "_us"] = round(time_plan(plan, x, y, 60), 2)
                if tag == "code":
                    info = plan.tiling_info
                    row["tiling"] = info
                    for f in ("G", "n_icb", "columns", "tpl", "nseg", "bands", "tr", "nbuf"):
                        m = re.search(r"\b%s=(\d+)" % f, info)
                        row[f] = int(m.group(1)) if m else None
                if tag == "auto":
                    row["auto_kernel"] = "generic" if "generic" in plan.kernel_name else "code" if "jit" in plan.kernel_name else plan.kernel_name
                    row["rule"] = plan.stat("small_launch_rule")
                plan.close()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
