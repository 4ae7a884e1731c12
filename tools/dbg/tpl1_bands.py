"""GPU box, experiments flavour: pointwise band-mode layers with one quad per lane (half-height bands, twice the tiles): parity + time."""
import importlib, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, sys, numpy as np, torch
sys.path.insert(0, %r)
pkg = importlib.import_module("caffe-escoin_amd"); synth = pkg.synth
import __graft_entry__ as ge
oracle = ge.load_oracle()
idx, n = int(sys.argv[1]), 256
s = synth.googlenet_1x1(N=n)[idx]
w, b = synth.pruned_weights(s, 1), synth.bias_vector(s, 2)
plan = pkg.Plan(pkg.ConvDesc.from_shape(s)); plan.weight_align(w)
dev = torch.device("cuda:0")
xs = [torch.rand((n, s.C, s.H, s.W), device=dev) * 2 - 1 for _ in range(4)]
bd = torch.from_numpy(b).to(dev)
ys = [torch.empty((n, s.M) + tuple(plan.out_hw), device=dev) for _ in range(4)]
for i in range(12): plan.forward(xs[i %% 4], bd, ys[i %% 4])
torch.cuda.synchronize()
g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0, 1, 1, 1, 1, 1)
want = oracle.conv_forward(g, xs[3][:6].cpu().numpy(), w, b, gate=False)
err = float(np.abs(ys[3][:6].cpu().numpy() - want).max() / np.abs(want).max())
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    a.record()
    for i in range(40): plan.forward(xs[i %% 4], bd, ys[i %% 4])
    e.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(e) / 40 * 1e3)
print("RESULT %%s %%.1f us err %%.2e | %%s" %% (s.name, best, err, plan.tiling_info))
''' % ROOT
for idx in (0, 1, 2, 5, 6, 8):
    for env in ({}, {"ESCOIN_FORCE_TPL": "1", "ESCOIN_TPL1_BANDS": "1"}):
        out = subprocess.run([sys.executable, "-c", CHILD, str(idx)], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [l for l in out.stdout.decode().splitlines() if l.startswith("RESULT")]
        print(("tpl1+bands " if env else "auto       ") + (lines[0][7:] if lines else "FAILED: " + out.stderr.decode()[-300:]), flush=True)
