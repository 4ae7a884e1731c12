#!/usr/bin/env python3
"""GPU box: the north-star sparsity sweep (60-95 %) of the ResNet-50 and AlexNet sets -- bench.py per (set, sparsity,
kernel): ms per step and us per distinct layer shape for generated code, the stream kernel and KERNEL_AUTO; the last
column says how far AUTO is from the better fixed kernel (within 2 % at every point).
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
    return ("%.3f (%s)" % (d["ms_per_step"], " ".join("%.1f" % l["us"] for l in d["roofline"]["per_layer"])), d.get("parity_max_rel_err"),
            d["ms_per_step"], d["value"])


def main():
    print("| set | sparsity | generated code | stream kernel | KERNEL_AUTO | AUTO images/s | AUTO vs the better fixed kernel |")
    print("|---|---|---|---|---|---|---|")
    worst, gap = 0.0, 0.0
    for wl in ("resnet50", "alexnet"):
        for sp in (0.6, 0.7, 0.8, 0.85, 0.9, 0.95):
            a, pa, ta, _ = run(wl, sp, "jit")
            b, pb, tb, _ = run(wl, sp, "tiled")
            c, pc, tc, vc = run(wl, sp, "auto")
            worst = max(worst, pa or 0, pb or 0, pc or 0)
            g = tc / min(ta, tb) - 1.0
            gap = max(gap, g)
            print("| %s | %d %% | %s | %s | %s | %.0f | %+.1f %% |" % (wl, round(sp * 100), a, b, c, vc, 100 * g))
            sys.stdout.flush()
    print("\nworst parity_max_rel_err of the %d runs: %.2g; AUTO at most %.1f %% behind the better fixed kernel" % (36, worst, 100 * gap))


if __name__ == "__main__":
    main()
