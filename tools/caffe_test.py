#!/usr/bin/env python3
"""`caffe test -conv_mode {1,2,3}` for the sparse convolution layers only.

The reference's `caffe test` (tools/caffe.cpp:262-362) loads a model + a pruned .caffemodel,
runs `-iterations` forward passes and prints "[cxh] Total CONV time" per pass
(caffe.cpp:338-339, accumulated in Net::ForwardFromTo, net.cpp:592-604).  This harness does
that for the layers on the SCONV path: every sparse conv layer of the named model gets a plan
(WeightAlign), then each iteration launches all of them on one HIP stream and reports the
convolution time from HIP events.  Non-convolution layers are out of scope.

    python tools/caffe_test.py --model resnet50 --iterations 5
    python tools/caffe_test.py --model alexnet --export /tmp/alexnet_pruned.caffemodel
    python tools/caffe_test.py --model alexnet --weights /tmp/alexnet_pruned.caffemodel --check

--weights takes the pruned weights (and biases) from a .caffemodel by layer name instead of the
synthetic generator; --export writes the synthetic pruned model in that format, so the same
file can be fed to the reference's own `caffe test`.  --check compares every layer's first
images with the CPU oracle (test infrastructure, not part of the product path).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def model_layers(synth, model, batch, sparsity):
    table = {"resnet50": (synth.resnet50_3x3, 256, 0.9), "alexnet": (synth.alexnet, 128, 0.8),
             "googlenet": (synth.googlenet_1x1, 256, 0.95), "lenet": (synth.lenet_conv2, 64, 0.5)}
    if model not in table:
        raise SystemExit("unknown --model %s (have: %s)" % (model, ", ".join(sorted(table))))
    fn, n, sp = table[model]
    shapes = fn(N=batch or n, sparsity=sp if sparsity is None else sparsity)
    out = []
    for s in shapes:
        for rep in range(s.count):
            name = s.name if s.count == 1 else "%s_%d" % (s.name, rep)
            out.append((name, s))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--weights", default=None, help=".caffemodel with pruned weights (by layer name)")
    ap.add_argument("--export", default=None, help="write the synthetic pruned model here and exit")
    ap.add_argument("--conv_mode", type=int, default=3, choices=[1, 2, 3],
                    help="1 = LOWERED_SPARSE comparator (im2col + csrmm), 2/3 = direct sparse convolution")
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--sparsity", type=float, default=None)
    ap.add_argument("--dense-gate", action="store_true",
                    help="honour the reference's density > 0.2 -> dense GEMM gate")
    ap.add_argument("--check", action="store_true", help="compare with the CPU oracle (first 2 images)")
    args = ap.parse_args()

    pkg = ge.load_package()
    from caffe_escoin_amd import caffemodel as cm
    synth = pkg.synth
    layers = model_layers(synth, args.model, args.batch, args.sparsity)

    weights = {}
    for i, (name, s) in enumerate(layers):
        weights[name] = (synth.pruned_weights(s, 1000 + 31 * i), synth.bias_vector(s, 2000 + 31 * i))
    if args.export:
        cl = [cm.CaffeLayer(name, "Convolution", [w] + ([b] if b is not None else []), cm.conv_param_of(s))
              for (name, s) in layers for (w, b) in [weights[name]]]
        cm.write_caffemodel(args.export, args.model + "_pruned", cl)
        print("wrote %s: %d convolution layers" % (args.export, len(cl)))
        return 0
    if args.weights:
        _, file_layers = cm.read_caffemodel(args.weights)
        got = cm.conv_weights(file_layers)
        used = 0
        for name, s in layers:
            if name not in got:
                continue
            w, b = got[name]
            want = (s.M, s.C // s.group, s.KH, s.KW)
            if tuple(w.shape) != want:
                raise SystemExit("%s: weight blob %s, layer expects %s" % (name, w.shape, want))
            weights[name] = (w, b if s.bias else None)
            used += 1
        print("[cxh] %s: weights of %d / %d layers taken from the file" % (args.weights, used, len(layers)))

    import torch
    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("caffe_test.py needs a HIP device: the product path has no CPU fallback")
    dev = torch.device("cuda", 0)
    print("[cxh] GPU device name: %s" % torch.cuda.get_device_name(0))

    plans, bottoms, tops, biases = [], {}, {}, []
    t0 = time.perf_counter()
    for name, s in layers:
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), conv_mode=args.conv_mode)
        if args.dense_gate:
            plan.set_option("dense_gate", 1)
        w, b = weights[name]
        plan.weight_align(w)
        plans.append(plan)
        biases.append(torch.from_numpy(b).to(dev) if b is not None else None)
        key = (s.C, s.H, s.W, s.M, s.KH, s.stride_h, s.pad_h)
        if key not in bottoms:
            g = torch.Generator(device=dev)
            g.manual_seed(len(bottoms) + 1)
            bottoms[key] = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=g) * 2 - 1
            oh, ow = synth.out_hw(s)
            tops[key] = torch.empty((s.N, s.M, oh, ow), device=dev)
    torch.cuda.synchronize()
    print("[cxh] WeightAlign of %d layers: %.1f ms" % (len(layers), 1e3 * (time.perf_counter() - t0)))

    def key_of(s):
        return (s.C, s.H, s.W, s.M, s.KH, s.stride_h, s.pad_h)

    per_layer = np.zeros(len(layers))
    conv_total = 0.0
    print("Running for %d iterations." % args.iterations)
    for it in range(args.iterations + 1):          # iteration 0 is an untimed warm-up
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in layers]
        for li, (name, s) in enumerate(layers):
            ev[li][0].record()
            plans[li].forward(bottoms[key_of(s)], biases[li], tops[key_of(s)])
            ev[li][1].record()
        torch.cuda.synchronize()
        if it == 0:
            continue
        ms = np.array([a.elapsed_time(b) for a, b in ev])
        per_layer += ms
        conv_total += ms.sum()
        print("[cxh] Total CONV time: %.2f ms" % ms.sum())
    print("[cxh] Average CONV time: %.3f ms over %d iterations (batch %d)" %
          (conv_total / args.iterations, args.iterations, layers[0][1].N))
    print("%-26s %-34s %9s %9s %8s" % ("layer", "kernel", "us", "TFLOP/s", "GB/s"))
    for li, (name, s) in enumerate(layers):
        us = 1e3 * per_layer[li] / args.iterations
        print("%-26s %-34s %9.1f %9.2f %8.0f" % (name, plans[li].kernel_name[:34], us,
                                                 synth.flops(s) / us * 1e-6,
                                                 synth.algorithmic_bytes(s) / us * 1e-3))

    if args.check:
        oracle = ge.load_oracle()
        worst = 0.0
        for li, (name, s) in enumerate(layers):
            n = min(2, s.N)
            x = bottoms[key_of(s)][:n].contiguous()
            top = plans[li].forward(x, biases[li]).cpu().numpy()
            w, b = weights[name]
            geom = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                               s.dil_h, s.dil_w, s.group)
            want = oracle.conv_forward(geom, x.cpu().numpy(), w, b, gate=args.dense_gate)
            err = float(np.abs(top.astype(np.float64) - want).max() / max(1e-6, np.abs(want).max()))
            worst = max(worst, err)
            if err > 1e-4:
                raise SystemExit("%s: relative error %.3g vs the oracle" % (name, err))
        print("[cxh] oracle check: worst relative error %.3g over %d layers" % (worst, len(layers)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
