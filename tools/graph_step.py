#!/usr/bin/env python3
"""GPU box: is escoin_forward capturable into a HIP graph, and what does replaying a whole step as one graph save over
launching its layers one by one?  (The reference's Net::Forward launches layer by layer; a host framework that
captures its forward pass needs the operator to launch on the capturing stream without synchronising or allocating.)
    python tools/graph_step.py [googlenet|resnet50|alexnet]"""
import importlib
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")
synth = pkg.synth


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "googlenet"
    shapes = {"googlenet": synth.googlenet_1x1, "resnet50": synth.resnet50_3x3, "alexnet": synth.alexnet}[wl]()
    dev = torch.device("cuda:0")
    layers = []
    for s in shapes:
        for rep in range(s.count):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
            plan.weight_align(synth.pruned_weights(s, 1 + rep))
            x = torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1
            y = torch.empty((s.N, s.M) + tuple(plan.out_hw), device=dev)
            b = torch.zeros(s.M, device=dev) if s.bias else None
            layers.append((plan, x, b, y))

    def step():
        for plan, x, b, y in layers:
            plan.forward(x, b, y)

    def timed(fn, n=60):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            e.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) / n)
        return best

    eager = timed(step)
    ref = [y.clone() for _, _, _, y in layers]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    for _, _, _, y in layers:
        y.zero_()
    with torch.cuda.graph(g):
        step()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(r, y) for r, (_, _, _, y) in zip(ref, layers))
    graph = timed(g.replay)
    print("%s: %d launches per step; eager %.4f ms per step, one graph replay %.4f ms (%+.1f %%); outputs identical: %s" %
          (wl, len(layers), eager, graph, (graph / eager - 1) * 100, same))


if __name__ == "__main__":
    main()
