#!/usr/bin/env python3
"""What lies between the kernels of a step: the same launches (bench.py's layers, own bottom / top per
layer) timed (a) with an event between every two launches, as bench.py records them, (b) with two
events around the whole run, (c) captured once into a HIP graph and replayed.
    python tools/gap_probe.py [resnet50|googlenet|alexnet] [steps]
"""
import importlib
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")
synth = pkg.synth


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    shapes = {"resnet50": lambda: synth.resnet50_3x3(N=256, sparsity=0.9), "googlenet": lambda: synth.googlenet_1x1(N=256, sparsity=0.95),
              "alexnet": lambda: synth.alexnet(N=128, sparsity=0.8)}[wl]()
    dev = torch.device("cuda:0")
    layers = []
    for k, s in enumerate(shapes):
        for rep in range(getattr(s, "count", 1)):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
            plan.weight_align(synth.pruned_weights(s, 100 + k))
            x = torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1
            oh, ow = synth.out_hw(s)
            y = torch.empty((s.N, s.M, oh, ow), device=dev)
            b = torch.zeros(s.M, device=dev) if s.bias else None
            layers.append((plan, x, b, y))
    side = torch.cuda.Stream()

    def step(stream_ptr=None):
        for (plan, x, b, y) in layers:
            plan.forward(x, b, y)

    def timed(fn, n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / n

    for _ in range(20):
        step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(layers) + 1)]

    def step_events():
        evs[0].record()
        for i, (plan, x, b, y) in enumerate(layers):
            plan.forward(x, b, y)
            evs[i + 1].record()
    res = {}
    for rep in range(2):
        res["events between launches"] = timed(step_events, steps)
        res["two events"] = timed(step, steps)
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
        torch.cuda.synchronize()
        for rep in range(2):
            res["graph replay"] = timed(g.replay, steps)
    print("%s: %d launches per step" % (wl, len(layers)))
    for k, v in res.items():
        print("  %-26s %.4f ms per step" % (k, v))


if __name__ == "__main__":
    main()
