#!/usr/bin/env python3
"""Conv-layer images/s of the ResNet-50 and AlexNet sets at 60-95 % weight sparsity (north star:
"images/sec on synthetic AlexNet/ResNet-50 shapes at 60-95 % weight sparsity ... alongside the
reference CPU path timed on the same box's host cores").  Runs bench.py once per point and prints a
markdown table (GPU box):

    python tools/sparsity_sweep.py > gpurun_out/r02_sparsity_sweep.md
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(workload, sparsity, cpu_budget):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--sparsity", str(sparsity),
           "--cpu-budget", str(cpu_budget)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900).stdout.decode()
    return json.loads(out.strip().splitlines()[-1])


def main():
    budget = float(os.environ.get("SWEEP_CPU_BUDGET", "8"))
    print("# Conv-layer forward images/s vs weight sparsity, one MI355X, fp32 (tools/sparsity_sweep.py)\n")
    print("Each row is one `bench.py --workload W --sparsity S` run (same contract as the bench line: inputs "
          "resident in HBM, default step counts, parity of the timed outputs against the oracle in the last column). "
          "`kernel mix` = what KERNEL_AUTO chose per conv group (dense fp32 MFMA above 50 % density). CPU = the "
          "reference's sconv on the box's host cores (default loop nest / its register-blocked kernel), "
          "bounded sample.\n")
    print("| set | sparsity | images/s | ms per step | sparse TFLOP/s | HBM frac (algorithmic) | binding frac | "
          "CPU images/s (cores) | CPU blocked images/s | GPU / CPU blocked | parity |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for wl in ("resnet50", "alexnet"):
        for sp in (0.6, 0.7, 0.8, 0.9, 0.95):
            d = run(wl, sp, budget)
            r, c = d["roofline"], d["cpu_baseline"]
            be = (c.get("best_effort") or {}).get("value")
            print("| %s | %d %% | %.0f | %.3f | %.1f | %.3f | %.3f | %.0f (%d) | %s | %s | %.1e |" % (
                wl, round(sp * 100), d["value"], d["ms_per_step"], r.get("sparse_tflops", 0.0), r["frac"],
                r.get("binding_frac", 0.0), c["value"], c["cores"], "%.0f" % be if be else "-",
                "%.0fx" % (d["value"] / be) if be else "-", d.get("parity_max_rel_err", float("nan"))))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
