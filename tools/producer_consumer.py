#!/usr/bin/env python3
"""Does a pointwise layer's top blob belong in the cache?  A GoogLeNet-style pair -- a 1x1 "reduce"
convolution and the 3x3 convolution that reads its output -- timed as a pair with the producer's
"stream_stores" option off and on (non-temporal stores keep the blob out of L2 / Infinity Cache).
bench.py's layers have no consumer; a net does.
    python tools/producer_consumer.py
"""
import importlib
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")
synth = pkg.synth

PAIRS = [("3a", 192, 28, 96, 128), ("3b", 256, 28, 128, 192), ("4a", 480, 14, 96, 208), ("4e", 528, 14, 160, 320), ("5a", 832, 7, 160, 320)]


def main():
    dev = torch.device("cuda:0")
    N = 256
    for (name, C, H, Mr, M3) in PAIRS:
        sp = synth.shape(name + "_reduce", N, C, H, H, Mr, 1, sparsity=0.95)
        sc = synth.shape(name + "_3x3", N, Mr, H, H, M3, 3, pad=1, sparsity=0.9)
        res = {}
        for ss in (0, 1, 0, 1):
            prod = pkg.Plan(pkg.ConvDesc.from_shape(sp), stream_stores=ss)
            prod.weight_align(synth.pruned_weights(sp, 1))
            cons = pkg.Plan(pkg.ConvDesc.from_shape(sc))
            cons.weight_align(synth.pruned_weights(sc, 2))
            nb = 4      # rotate buffers: the producer's input comes from HBM as in a net
            xs = [torch.rand((N, C, H, H), device=dev) * 2 - 1 for _ in range(nb)]
            mid = [torch.empty((N, Mr, H, H), device=dev) for _ in range(nb)]
            out = [torch.empty((N, M3, H, H), device=dev) for _ in range(nb)]
            bp, bc = torch.zeros(Mr, device=dev), torch.zeros(M3, device=dev)

            def pair(i):
                prod.forward(xs[i % nb], bp, mid[i % nb])
                cons.forward(mid[i % nb], bc, out[i % nb])
            for i in range(20):
                pair(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(4):
                a.record()
                for i in range(100):
                    pair(i)
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 100 * 1e3)
            res.setdefault(ss, []).append(best)
            prod.close()
            cons.close()
        print("%s: %d -> %d @%dx%d then 3x3 -> %d:  pair with plain stores %s us, with streaming stores %s us" %
              (name, C, Mr, H, H, M3, " / ".join("%.1f" % v for v in res[0]), " / ".join("%.1f" % v for v in res[1])))


if __name__ == "__main__":
    main()
