#!/usr/bin/env python3
"""GPU box: every distinct dense 1x1 layer shape of the ResNet-50 bottleneck chain (tools/caffe_test.py
--model resnet50_chain) on the fp32-MFMA kernel, back to back: us, dense TFLOP/s, algorithmic GB/s.
    python tools/chain_dense.py [batch]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge
import caffe_test

pkg = ge.load_package(); synth = pkg.synth
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None      # e.g. "64x56x256,256x56x64": C x H x M filters
chain = caffe_test.resnet50_chain(synth, batch, 0.9)
dev = torch.device("cuda:0")
seen, tot_us, tot_fl = {}, 0.0, 0.0
for (name, kind, s, relu, role) in chain:
    if kind != "dense":
        continue
    key = (s.C, s.H, s.M, s.stride_h)
    if only and "%dx%dx%d" % (s.C, s.H, s.M) not in only:
        continue
    if key not in seen:
        w = synth.pruned_weights(s, 5)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=pkg.KERNEL_DENSE)
        plan.weight_align(w)
        xs = [torch.rand((s.N, s.C, s.H, s.W), device=dev) for _ in range(3)]
        oh, ow = synth.out_hw(s)
        ys = [torch.empty((s.N, s.M, oh, ow), device=dev) for _ in range(3)]
        for i in range(6):
            plan.forward(xs[i % 3], None, ys[i % 3])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 30
        for i in range(n):
            plan.forward(xs[i % 3], None, ys[i % 3])
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        fl = 2.0 * s.N * oh * ow * s.M * s.C
        by = 4.0 * s.N * (s.C * s.H * s.W + s.M * oh * ow)
        seen[key] = (us, fl, by, 0)
        plan.close()
        del xs, ys
    us, fl, by, cnt = seen[key]
    seen[key] = (us, fl, by, cnt + 1)
for key, (us, fl, by, cnt) in seen.items():
    print("C=%4d %3dx%-3d -> M=%4d s%d  x%d  %8.1f us  %6.1f TFLOP/s  %7.0f GB/s  (MFMA floor %.0f us, HBM floor %.0f us)" %
          (key[0], key[1], key[1], key[2], key[3], cnt, us, fl / us * 1e-6, by / us * 1e-3, fl / 157.3e6, by / 8e6))
    tot_us += us * cnt; tot_fl += fl * cnt
print("chain dense total %.2f ms, %.1f TFLOP/s average" % (tot_us * 1e-3, tot_fl / tot_us * 1e-6))
