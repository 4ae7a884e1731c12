#!/usr/bin/env python3
"""In-kernel stamp profile of escoin_dense_mfma_kernel on named dense 1x1 shapes of the ResNet-50 chain.
    ESCOIN_LIB=$PWD/tools/ab/libescoin_abl.so ESCOIN_PROF=1 python tools/dense_stamps.py          (stamps, stderr)
    python tools/dense_stamps.py                                                                  (timing only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

SHAPES = [("k64_56_to256", 64, 56, 256, 1), ("c1024_14_to512_s2", 1024, 14, 512, 2), ("c1024_14_to256", 1024, 14, 256, 1),
          ("c256_56_to512_s2", 256, 56, 512, 2), ("c256_56_to64", 256, 56, 64, 1)]


def main():
    import torch
    pkg = ge.load_package()
    synth = pkg.synth
    dev = torch.device("cuda:0")
    only = sys.argv[1:] or None
    for name, c, hw, m, stride in SHAPES:
        if only and name not in only:
            continue
        s = synth.shape(name, 256, c, hw, hw, m, 1, stride=stride, bias=False, sparsity=0.0)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_DENSE)
        plan.weight_align(synth.pruned_weights(s, 7))
        oh, ow = plan.out_hw
        xs = [torch.rand((s.N, s.C, s.H, s.W), device=dev) * 2 - 1 for _ in range(3)]
        ys = [torch.empty((s.N, s.M, oh, ow), device=dev) for _ in range(3)]
        prof = os.environ.get("ESCOIN_PROF") == "1"
        n = 3 if prof else 30
        for i in range(3):
            plan.forward(xs[i % 3], None, ys[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            plan.forward(xs[i % 3], None, ys[i % 3])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        fl = 2.0 * s.N * oh * ow * s.M * s.C
        by = 4.0 * s.N * (s.C * s.H * s.W + s.M * oh * ow)
        print("%s %d@%dx%d->%d s%d: %.1f us  %.1f TFLOP/s  blobs %.0f MB = %.2f TB/s%s" %
              (name, c, hw, hw, m, stride, us, fl / us / 1e6, by / 1e6, by / us / 1e6, "  (stamped: synchronising launches)" if prof else ""), flush=True)
        sys.stderr.flush()
        plan.close()


if __name__ == "__main__":
    main()
