#!/usr/bin/env python3
"""profiles/r03_summary.md from the committed bench lines (profiles/r03_bench_<set>.json) and the plans'
tilings as `tools/caffe_test.py --model <set> --tilings` printed them on the GPU box (gpurun_out/til/<set>.txt).
    python tools/make_summary.py [tag]"""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = ["# %s: per-layer summary of the final binary (one MI355X)\n" % tag,
       "us / TFLOP/s / fractions from `profiles/%s_bench_<set>.json` (`python bench.py --workload <set>`: every layer with its own bottom / top pair, per-launch events in 10 of the timed steps with their own cost taken off); `traffic / alg` = HBM bytes by PMC (`profiles/traffic_<set>.json`, separate rocprofv3 passes) over the algorithmic bytes of SURVEY 8(d); tiling = `escoin_plan_tiling_info()` as printed by `tools/caffe_test.py --model <set> --tilings` (cut = the image as the kernel walks it, tpl = quads per lane, G = output channels per wave, columns = workgroup columns per conv group, nseg = images per tile, bands = row bands per image, qpc = quads per staged channel plane, icb x n_icb = input channels per block x blocks, nbuf = plane buffers).\n" % tag]
for wl in ("resnet50", "alexnet", "googlenet", "lenet"):
    d = json.load(open("%s/profiles/%s_bench_%s.json" % (root, tag, wl)))
    r = d["roofline"]
    til = {}
    path = "%s/gpurun_out/til/%s.txt" % (root, wl)
    if os.path.exists(path):
        for line in open(path):
            m = re.match(r"(\S+)\s+((generated-code|stream) cut=.*)$", line.strip())
            if m:
                til[m.group(1)] = m.group(2)
    out.append("\n## %s\n\n%.0f images/s, %s ms per step, %.3f of the HBM roofline on the set (binding fraction %.3f), parity_max_rel_err %.2g\n" %
               (d["config"]["workload"], d["value"], d["ms_per_step"], r["frac"], r["binding_frac"], d["parity_max_rel_err"]))
    out.append("| layer | x | us | sparse TFLOP/s | alg GB/s | HBM frac | binding frac | traffic / alg | tiling |\n|---|---|---|---|---|---|---|---|---|")
    for l in r["per_layer"]:
        name = l["layer"]
        t = til.get(name) or next((v for k, v in til.items() if k.startswith(name.split("_branch")[0][:4])), "")
        t = re.sub(r" (n_ocg|oc_waves|pix_waves|lds|period|tr)=\S+", "", t)
        out.append("| %s | %d | %s | %s | %s | %.3f | %.3f | %s | %s |" % (name, l["count"], l["us"], l["sparse_TFLOPs"], l["alg_GBps"], l["hbm_frac"],
                                                                     l["binding_frac"], l["traffic_over_alg"], t))
open("%s/profiles/%s_summary.md" % (root, tag), "w").write("\n".join(out) + "\n")
print("profiles/%s_summary.md" % tag)
