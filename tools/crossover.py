#!/usr/bin/env python3
"""BASELINE.json configs[4] -- where the dense-MFMA fallback and the two sparse walks cross.

For each layer shape and weight sparsity: forward time of (a) generated code (KERNEL_JIT), (b) the
LDS-staged stream kernel (KERNEL_TILED), (c) the dense fp32-MFMA implicit GEMM (KERNEL_DENSE), and what
KERNEL_AUTO picks -- the gate the reference hard-codes as `density > 0.2 -> GEMM`
(base_conv_layer.cpp:750-755, 805-811), re-measured for this hardware and these kernels.  Every launch reads
its blobs from HBM: the layer's bottom / top pair rotates over enough copies to exceed the 256 MB
Infinity Cache, as in a net where 2-3 GB of other layers' traffic passes between two launches of a layer.

Runs on the GPU box:  python tools/crossover.py [--batch 256] [--json out.json] > profiles/<tag>_crossover.md
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def time_plan(torch, plan, xs, bias, tops, reps=12):
    nb = len(xs)
    for i in range(4):
        plan.forward(xs[i % nb], bias, tops[i % nb])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for i in range(reps):
            plan.forward(xs[i % nb], bias, tops[i % nb])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)     # us
    return best


def shapes_of(synth, batch, which):
    gl = synth.googlenet_1x1(N=batch)
    rn = synth.resnet50_3x3(N=batch)
    al = synth.alexnet(N=128)
    full = [gl[0], gl[1], gl[5], gl[9], gl[25], gl[33], rn[0], rn[1], rn[2], rn[3], al[0], al[1], al[2], al[3]]
    if which == "all":
        return full
    pick = set(which.split(","))
    return [s for s in full if s.name.split("/")[0].split("_branch")[0] in pick or s.name in pick]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--layers", default="all")
    ap.add_argument("--sparsities", default="0,50,60,70,80,85,90,95")
    ap.add_argument("--json", default=None, help="also write the raw numbers here")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    synth = pkg.synth
    dev = torch.device("cuda:0")
    sparsities = [float(v) / 100.0 for v in args.sparsities.split(",")]
    kinds = [("jit", pkg.KERNEL_JIT), ("stream", pkg.KERNEL_TILED), ("dense", pkg.KERNEL_DENSE), ("auto", pkg.KERNEL_AUTO)]
    names = {pkg.KERNEL_JIT: "jit", pkg.KERNEL_TILED: "stream", pkg.KERNEL_DENSE: "dense", pkg.KERNEL_GENERIC: "generic"}
    print("us per launch: generated code / stream kernel / dense MFMA -> what KERNEL_AUTO runs (its time; `!` where "
          "AUTO is more than 2 % behind the best of the three)\n")
    print("| layer (C@HxW -> M, K) | " + " | ".join("@%d %%" % round(100 * s) for s in sparsities) + " |")
    print("|---|" + "---|" * len(sparsities))
    raw = []
    for s in shapes_of(synth, args.batch, args.layers):
        oh, ow = synth.out_hw(s)
        per_pair = 4.0 * s.N * (s.C * s.H * s.W + s.M * oh * ow)
        nb = int(max(2, min(8, np.ceil(600e6 / per_pair))))
        xs = [torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1 for _ in range(nb)]
        tops = [torch.empty((s.N, s.M, oh, ow), device=dev) for _ in range(nb)]
        bias = torch.zeros(s.M, device=dev) if s.bias else None
        cells = []
        t_dense = None
        for sp in sparsities:
            ss = s._replace(sparsity=sp)
            w = synth.pruned_weights(ss, 7)
            t = {}
            picked, code_mb = None, 0.0
            for kname, kid in kinds:
                if kname == "dense" and t_dense is not None:
                    t[kname] = t_dense           # (the dense kernel's time does not depend on the zeros)
                    continue
                if kname in ("jit", "stream") and sp == 0.0 and s.KH > 1:
                    t[kname] = None              # (a dense 3x3 / 5x5 layer as generated code: hundreds of MB)
                    continue
                try:
                    plan = pkg.Plan(pkg.ConvDesc.from_shape(ss), kernel=kid)
                    plan.weight_align(w)
                except pkg.EscoinError:
                    t[kname] = None
                    continue
                t[kname] = time_plan(torch, plan, xs, bias, tops)
                if kname == "dense":
                    t_dense = t[kname]
                if kname == "auto":
                    picked = names.get(plan.stat("kernel_choice"), "?")
                    code_mb = plan.stat("code_bytes") / 1e6
                plan.close()
            three = [v for k, v in t.items() if k != "auto" and v is not None]
            best = min(three)
            flag = "!" if t["auto"] is not None and t["auto"] > 1.02 * best else ""
            f = lambda v: "--" if v is None else "%.0f" % v
            cells.append("%s / %s / %s -> %s %s%s" % (f(t["jit"]), f(t["stream"]), f(t["dense"]), picked, f(t["auto"]), flag))
            raw.append({"layer": s.name, "C": s.C, "H": s.H, "M": s.M, "K": s.KH, "group": s.group, "N": s.N,
                        "sparsity": sp, "us": t, "auto": picked, "auto_code_mb": code_mb,
                        "nnz": int(round((1 - sp) * s.M * (s.C // s.group) * s.KH * s.KW))})
            sys.stderr.write("%s @%d: %s\n" % (s.name, round(100 * sp), cells[-1]))
        print("| %s (%d@%dx%d -> %d, %dx%d, N=%d) | %s |" % (s.name, s.C, s.H, s.W, s.M, s.KH, s.KW, s.N, " | ".join(cells)))
        sys.stdout.flush()
        del xs, tops
        torch.cuda.empty_cache()
    if args.json:
        json.dump(raw, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
