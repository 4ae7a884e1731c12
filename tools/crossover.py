#!/usr/bin/env python3
"""Sparse (tiled, VALU) vs dense (fp32 MFMA implicit GEMM) forward time as a function of weight
sparsity -- BASELINE.json configs[4]: GoogLeNet-v1 1x1 convs and the dense-fallback crossover --
and, at the layer's nominal sparsity, the lowering baseline the reference compares against
(conv_mode LOWERED_SPARSE: im2col + CSR x dense, run.sh:8-12).
Runs on the GPU box:  python tools/crossover.py [--batch 256] > profiles/<tag>_crossover.md"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def time_plan(torch, plan, x, bias, top, reps=10):
    for _ in range(3):
        plan.forward(x, bias, top)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.forward(x, bias, top)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dense-only", action="store_true", help="only time the dense MFMA kernel")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    synth = pkg.synth
    dev = torch.device("cuda:0")
    gl = synth.googlenet_1x1(N=args.batch)
    layers = [gl[0], gl[1], gl[5], gl[9], gl[25], gl[33], synth.resnet50_3x3(N=args.batch)[0],
              synth.resnet50_3x3(N=args.batch)[2], synth.alexnet(N=128)[1]]
    sparsities = [0.0, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95]
    print("| layer (C@HxW -> M, K) | dense MFMA us (TFLOP/s dense) | " +
          " | ".join("sparse @%d%% us" % round(100 * s) for s in sparsities) +
          " | crossover | lowered csrmm @nominal us (direct speedup) |")
    print("|---|---|" + "---|" * (len(sparsities) + 2))
    for s in layers:
        x = torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1
        oh, ow = synth.out_hw(s)
        top = torch.empty((s.N, s.M, oh, ow), device=dev)
        bias = torch.zeros(s.M, device=dev) if s.bias else None
        wd = synth.pruned_weights(s._replace(sparsity=0.0), 7)
        pd = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_DENSE)
        pd.weight_align(wd)
        td = time_plan(torch, pd, x, bias, top)
        pd.close()
        dense_flops = 2.0 * s.N * oh * ow * s.M * (s.C // s.group) * s.KH * s.KW
        if args.dense_only:
            print("%-28s %4d@%dx%d -> %4d %dx%d  dense %8.1f us  %6.1f TFLOP/s" %
                  (s.name, s.C, s.H, s.W, s.M, s.KH, s.KW, td, dense_flops / td / 1e6))
            sys.stdout.flush()
            continue
        row, cross = [], None
        for sp in sparsities:
            ss = s._replace(sparsity=sp)
            ps = pkg.Plan(pkg.ConvDesc.from_shape(ss), kernel=pkg.KERNEL_TILED)
            ps.weight_align(synth.pruned_weights(ss, 7))
            t = time_plan(torch, ps, x, bias, top)
            ps.close()
            row.append("%.0f" % t)
            if cross is None and t < td:
                cross = sp
        # lowering baseline and direct path at the layer's own sparsity
        pl = pkg.Plan(pkg.ConvDesc.from_shape(s), conv_mode=pkg.CONV_MODE_LOWERED_SPARSE)
        pl.weight_align(synth.pruned_weights(s, 7))
        tl = time_plan(torch, pl, x, bias, top, reps=5)
        pl.set_option("conv_mode", pkg.CONV_MODE_SCONV_PAR)
        tn = time_plan(torch, pl, x, bias, top)
        pl.close()
        print("| %s (%d@%dx%d -> %d, %dx%d) | %.0f (%.1f) | %s | %s | %.0f @%d%% (%.1fx) |" %
              (s.name, s.C, s.H, s.W, s.M, s.KH, s.KW, td, dense_flops / td / 1e6, " | ".join(row),
               ("sparse wins from %d%%" % round(100 * cross)) if cross is not None else "dense wins everywhere",
               tl, round(100 * s.sparsity), tl / tn))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
