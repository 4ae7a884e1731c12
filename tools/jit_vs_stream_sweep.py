#!/usr/bin/env python3
"""GPU box: the sparsity sweep of profiles/r03_jit_vs_stream.md -- bench.py per (set, sparsity, kernel),
ms per step and us per distinct layer shape, generated code against the stream kernel.
    python tools/jit_vs_stream_sweep.py > gpurun_out/jit_vs_stream_sweep.md"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(workload, sparsity, kernel):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--sparsity", str(sparsity),
           "--kernel", kernel, "--no-cpu"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900).stdout.decode()
    d = json.loads(out.strip().splitlines()[-1])
    return "%.3f (%s)" % (d["ms_per_step"], " ".join("%.1f" % l["us"] for l in d["roofline"]["per_layer"])), d.get("parity_max_rel_err")


def main():
    print("| set | sparsity | generated code | stream kernel |")
    print("|---|---|---|---|")
    worst = 0.0
    for wl in ("resnet50", "alexnet"):
        for sp in (0.6, 0.7, 0.8, 0.85, 0.9, 0.95):
            a, pa = run(wl, sp, "jit")
            b, pb = run(wl, sp, "tiled")
            worst = max(worst, pa or 0, pb or 0)
            print("| %s | %d %% | %s | %s |" % (wl, round(sp * 100), a, b))
            sys.stdout.flush()
    print("\nworst parity_max_rel_err of the %d runs: %.2g" % (24, worst))


if __name__ == "__main__":
    main()
