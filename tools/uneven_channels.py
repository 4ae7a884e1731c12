import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("caffe-escoin_amd"); synth = pkg.synth
rng = np.random.RandomState(3)
for (C, H, M) in [(256, 14, 256), (128, 28, 128)]:
    s = synth.shape("u", 256, C, H, H, M, 3, pad=1, sparsity=0.0)
    w = synth.pruned_weights(s, 5)
    keep = rng.uniform(0.0, 0.2, size=M)
    w = (w * (rng.uniform(size=w.shape) < keep[:, None, None, None])).astype(np.float32)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s)); plan.weight_align(w)
    x = torch.rand((256, C, H, H), device="cuda:0") * 2 - 1
    for _ in range(30): y = plan.forward(x, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): y = plan.forward(x, None)
    b.record(); torch.cuda.synchronize()
    print("BALANCE=%s C%d %dx%d M%d channel densities U(0,0.2): %.1f us  (%s, density %.3f)" % (os.environ.get("ESCOIN_BALANCE", "1"), C, H, H, M, a.elapsed_time(b) / 200 * 1e3, plan.kernel_name, float((w != 0).mean())))
