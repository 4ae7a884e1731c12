"""Host only: Caffe::CPU mode with one image per job against two or three (option cpu_images_per_job), small-image layers.
    python tools/dbg/cpu_images_per_job.py <threads> <batch>"""
import importlib, sys, time, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("caffe-escoin_amd"); synth = pkg.synth
thr = int(sys.argv[1]); n = int(sys.argv[2])
shapes = [synth.resnet50_3x3(N=n)[3]] + [synth.googlenet_1x1(N=n)[i] for i in (29, 32, 33, 37)] + synth.lenet_conv2(N=n)
for s in shapes:
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    w = synth.pruned_weights(s, 1); plan.weight_align_cpu(w)
    x = synth.activations(s, 3, 0, n)
    y = np.zeros((n, s.M) + tuple(plan.out_hw), np.float32)
    res = {}
    for rnd in range(3):
        for mode in (1, 0):
            plan.set_option("cpu_images_per_job", mode)
            plan.forward_cpu(x, None, n_threads=thr, out=y)
            best = 1e9
            for _ in range(5):
                t = time.perf_counter(); plan.forward_cpu(x, None, n_threads=thr, out=y); best = min(best, time.perf_counter() - t)
            res[mode] = min(res.get(mode, 1e9), best)
            if mode == 0: ni = plan.stat("cpu_images_per_job")
    print("%-26s C=%4d %2dx%-2d M=%4d ni=%d cb=%d  one %9.0f img/s  multi %9.0f img/s  %+5.0f %%" % (s.name, s.C, s.H, s.W, s.M, ni, plan.stat("cpu_channel_block"), n / res[1], n / res[0], 100 * (res[1] / res[0] - 1)), flush=True)
