#!/usr/bin/env python3
"""Times (and, under rocprofv3, profiles) the dense fp32-MFMA kernel on one layer shape.
    python tools/dense_probe.py res4 [reps]      shapes: res2 res3 res4 res5 alex3 g3b g4e g_c2r"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    import torch
    pkg = ge.load_package()
    synth = pkg.synth
    which = sys.argv[1] if len(sys.argv) > 1 else "res4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    gl = synth.googlenet_1x1(N=256)
    table = {"res2": synth.resnet50_3x3(N=256)[0], "res3": synth.resnet50_3x3(N=256)[1],
             "res4": synth.resnet50_3x3(N=256)[2], "res5": synth.resnet50_3x3(N=256)[3],
             "alex3": synth.alexnet(N=128)[1], "g_c2r": gl[0], "g3b": gl[5], "g4e": gl[25]}
    s = table[which]
    dev = torch.device("cuda:0")
    x = torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1
    oh, ow = synth.out_hw(s)
    top = torch.empty((s.N, s.M, oh, ow), device=dev)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_DENSE)
    plan.weight_align(synth.pruned_weights(s._replace(sparsity=0.0), 7))
    for _ in range(3):
        plan.forward(x, None, top)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.forward(x, None, top)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * s.N * oh * ow * s.M * (s.C // s.group) * s.KH * s.KW
    print("%s: %.1f us  %.1f TFLOP/s dense" % (s.name, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
