#!/usr/bin/env python3
"""The layers of a bench step are independent (each has its own bottom / top pair), like the parallel 1x1
branches of an inception module: issued round robin on S HIP streams, a layer's launch, start-up and drain overlap
its neighbours' streaming phase -- a layer's workgroup needs a whole CU (160 KiB of LDS), so two kernels never
share one, but the next kernel's workgroups move onto CUs as the previous kernel's retire instead of after its
last one has.  What a net-level scheduler could get out of the layer-level drop-in; the reference (and bench.py's
default line) runs everything on one stream.
    python tools/two_streams.py [--workload googlenet] [--streams 1,2,3]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="googlenet")
    ap.add_argument("--streams", default="1,2,3")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    synth = pkg.synth
    dev = torch.device("cuda:0")
    shapes, name = bench.workload_layers(synth, args.workload, None, None)
    layers = []
    lid = 0
    for si, s in enumerate(shapes):
        for rep in range(s.count):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
            plan.weight_align(synth.pruned_weights(s, 1000 + 31 * lid))
            b = synth.bias_vector(s, 2000 + 31 * lid)
            x = torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1
            oh, ow = synth.out_hw(s)
            layers.append((plan, x, torch.from_numpy(b).to(dev) if b is not None else None, torch.empty((s.N, s.M, oh, ow), device=dev)))
            lid += 1
    n_img = shapes[0].N
    alg = sum(synth.algorithmic_bytes(s, s.N) * s.count for s in shapes)
    for S in [int(v) for v in args.streams.split(",")]:
        streams = [torch.cuda.current_stream()] if S == 1 else [torch.cuda.Stream() for _ in range(S)]

        def step():
            for li, (plan, x, b, top) in enumerate(layers):
                with torch.cuda.stream(streams[li % S]):
                    plan.forward(x, b, top)
        for _ in range(60):
            step()
        torch.cuda.synchronize()
        best = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / args.steps * 1e3)
        ms = float(np.median(best))
        print("%-10s %d stream(s): %.4f ms per step (min %.4f max %.4f)  %.1f k images/s  %.0f GB/s algorithmic = %.3f of 8 TB/s" %
              (args.workload, S, ms, min(best), max(best), n_img / ms, alg / ms / 1e6, alg / ms / 1e6 / 8000))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
