"""GPU box: what Dtype = double costs on the device (order-preserving generic kernel in fp64) beside the fp32 paths,
ResNet-50 3x3 shapes @90 %, batch 64.   python tools/dbg/f64_rate.py"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("caffe-escoin_amd"); synth = pkg.synth
n = 64
dev = torch.device("cuda:0")
tot = {"f64": 0.0, "f32 generic": 0.0, "f32 auto": 0.0}
for s in synth.resnet50_3x3(N=n):
    w = synth.pruned_weights(s, 1)
    x = torch.from_numpy(synth.activations(s, 2, 0, n)).to(dev)
    res = {}
    for name, kernel, wt, xt in (("f64", pkg.KERNEL_AUTO, w.astype(np.float64), x.double()), ("f32 generic", pkg.KERNEL_GENERIC, w, x), ("f32 auto", pkg.KERNEL_AUTO, w, x)):
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel)
        plan.weight_align(wt)
        y = plan.forward(xt, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if name != "f32 auto" else 20
        a.record()
        for _ in range(reps): plan.forward(xt, None, y)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        res[name] = (ms, plan.kernel_name)
        tot[name] += ms * s.count / n
        plan.close()
    print("%-14s " % s.name + "  ".join("%s %.3f ms (%s)" % (k, v[0], v[1]) for k, v in res.items()), flush=True)
print("images/s over the 16 layers: " + ", ".join("%s %.0f" % (k, 1e3 / v) for k, v in tot.items()))
