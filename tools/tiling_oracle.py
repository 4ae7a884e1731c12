#!/usr/bin/env python3
"""GPU box, experiments flavour (tools/mkabl.sh exp; ESCOIN_LIB must point at it): for one layer shape and batch, every
tiling the cost model could have chosen -- workgroup columns (passes) x images per tile -- forced through
ESCOIN_FORCE_PASSES / ESCOIN_FORCE_NSEG, timed HBM-cold, beside KERNEL_AUTO's own pick.  One child process per tiling
(the switches are read once per process).
    ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so python tools/tiling_oracle.py res4 192 [257 ...]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import importlib, os, sys, torch
sys.path.insert(0, %r)
pkg = importlib.import_module("caffe-escoin_amd"); synth = pkg.synth
name, n = sys.argv[1], int(sys.argv[2])
d = {s.name.split("_")[0]: s for s in synth.resnet50_3x3(N=n)}
d.update({"alex%%d" %% (i + 2): s for i, s in enumerate(synth.alexnet(N=n))})
d.update({"goog%%d" %% i: s for i, s in enumerate(synth.googlenet_1x1(N=n))})
s = d[name]
plan = pkg.Plan(pkg.ConvDesc.from_shape(s)); plan.weight_align(synth.pruned_weights(s, 1))
dev = torch.device("cuda:0")
oh, ow = plan.out_hw
nb = max(1, min(4, int(3e9 // (4 * n * (s.C * s.H * s.W + s.M * oh * ow)))))
xs = [torch.rand((n, s.C, s.H, s.W), device=dev) * 2 - 1 for _ in range(nb)]
ys = [torch.empty((n, s.M, oh, ow), device=dev) for _ in range(nb)]
for i in range(12): plan.forward(xs[i %% nb], None, ys[i %% nb])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e30
for _ in range(3):
    a.record()
    for i in range(40): plan.forward(xs[i %% nb], None, ys[i %% nb])
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 40 * 1e3)
print("RESULT %%.1f %%s | %%s" %% (best, plan.kernel_name, plan.tiling_info))
""" % ROOT


def run(name, n, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD, name, str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
    for line in out.stdout.decode().splitlines():
        if line.startswith("RESULT"):
            us, rest = line[7:].split(" ", 1)
            return float(us), rest
    return None, ""


def main():
    name = sys.argv[1]
    for n in [int(v) for v in sys.argv[2:]]:
        us0, info0 = run(name, n, {})
        m = {k: int(v) for k, v in re.findall(r"\b(columns|nseg|G|n_icb)=(\d+)", info0)}
        print("%s N=%d AUTO: %.1f us  columns=%s nseg=%s G=%s" % (name, n, us0 or -1, m.get("columns"), m.get("nseg"), m.get("G")), flush=True)
        seen = set()
        base = m.get("columns", 1)
        nsegs = [int(v) for v in os.environ.get("ORACLE_NSEGS", "0,1").split(",")]
        tpls = [int(v) for v in os.environ.get("ORACLE_TPLS", "2").split(",")]       # 1 = one quad per lane (pointwise layers)
        pass_list = [int(v) for v in os.environ.get("ORACLE_PASSES", "1,2,3,4,5,6,7,8,10,12,16").split(",")]
        for passes in sorted(set(pass_list + [base])):
          for tpl in tpls:
            for nseg in nsegs:
                us, info = run(name, n, {"ESCOIN_FORCE_PASSES": str(passes), "ESCOIN_FORCE_NSEG": str(nseg), "ESCOIN_FORCE_TPL": str(tpl)})
                mm = {k: int(v) for k, v in re.findall(r"\b(columns|nseg|G|n_icb|tpl)=(\d+)", info)}
                key = (mm.get("columns"), mm.get("nseg"), mm.get("G"), mm.get("tpl"))
                if us is None or key in seen:
                    continue
                seen.add(key)
                print("    columns=%s nseg=%s G=%s tpl=%s blocks=%s: %.1f us%s" % (key[0], key[1], key[2], key[3], mm.get("n_icb"), us, "  <-- better than AUTO by %.0f %%" % (100 * (1 - us / us0)) if us0 and us < 0.97 * us0 else ""), flush=True)


if __name__ == "__main__":
    main()
