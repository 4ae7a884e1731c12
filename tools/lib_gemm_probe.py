#!/usr/bin/env python3
"""What the ROCm libraries do on the dense 1x1 layers of the ResNet-50 chain (fp32, batch 256):
torch.matmul (hipBLASLt / rocBLAS strided-batched SGEMM) and F.conv2d (MIOpen) beside this repo's
fp32-MFMA kernel.  A yardstick for DESIGN 4.4, not a product path.
    python tools/lib_gemm_probe.py
LIBPROBE_ONLY=1: this repo's kernel only (ESCOIN_LIB=... for a variant build); LIBPROBE_BUFS=k rotates k
bottom / top pairs (more than the Infinity Cache holds: HBM-cold).
"""
import importlib
import os
import sys
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("caffe-escoin_amd")

SHAPES = [(64, 256, 56), (256, 64, 56), (64, 64, 56), (256, 128, 28), (128, 512, 28), (512, 128, 28), (512, 256, 14), (256, 1024, 14),
          (1024, 256, 14), (1024, 512, 7), (512, 2048, 7), (2048, 512, 7)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    N = 256
    only = bool(os.environ.get("LIBPROBE_ONLY"))
    nb = int(os.environ.get("LIBPROBE_BUFS", "1"))
    print("%-22s %10s %10s %10s   (TFLOP/s)" % ("C -> M @ HxW", "matmul", "conv2d", "escoin"))
    for (C, M, H) in SHAPES:
        x = torch.rand((N, C, H, H), device=dev) * 2 - 1
        w = torch.rand((M, C, 1, 1), device=dev) * 2 - 1
        flops = 2.0 * N * C * M * H * H
        w2, x3 = w.view(M, C), x.view(N, C, H * H)
        t_mm = float("inf") if only else timeit(lambda: torch.matmul(w2, x3))
        t_cv = float("inf") if only else timeit(lambda: F.conv2d(x, w))
        d = pkg.ConvDesc(N, C, H, H, M, 1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 0)
        plan = pkg.Plan(d, kernel=pkg.KERNEL_DENSE)
        plan.weight_align(w.cpu().numpy())
        y = torch.empty((N, M, H, H), device=dev)
        xs = [x] + [torch.rand_like(x) for _ in range(nb - 1)]
        ys = [y] + [torch.empty_like(y) for _ in range(nb - 1)]
        cnt = [0]

        def run():
            i = cnt[0] % nb
            cnt[0] += 1
            plan.forward(xs[i], None, ys[i])
        t_es = timeit(run)
        plan.forward(x, None, y)
        ref = torch.matmul(w2, x3).view(N, M, H, H)
        err = float((y - ref).abs().max() / ref.abs().max())
        print("%4d -> %4d @ %2dx%-2d     %10.1f %10.1f %10.1f   err %.1e" % (C, M, H, H, flops / t_mm / 1e6, flops / t_cv / 1e6, flops / t_es / 1e6, err))


if __name__ == "__main__":
    main()
