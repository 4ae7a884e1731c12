#!/usr/bin/env python3
"""Condenses what tools/evidence.sh left under gpurun_out/<tag>/ into tracked files under profiles/:
   <round>_probe_valu_rate.txt, <round>_probe_mixload.txt, <round>_stamp_profile.md, <round>_jit_ablations.md
     python tools/make_evidence.py gpurun_out/r04a r04
"""
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")


def last(path, pat):
    hits = [l.rstrip("\n") for l in open(path) if pat in l]
    return hits[-1] if hits else ""


for name in ("probe_valu_rate", "probe_mixload"):
    if not os.path.exists(os.path.join(src, name + ".txt")):
        continue
    with open(os.path.join(out, "%s_%s.txt" % (tag, name)), "w") as f:
        f.write("# tools/probes/%s.hip on one MI355X (tools/evidence.sh), raw output\n" % name)
        f.write(open(os.path.join(src, name + ".txt")).read())

layers = ["res2", "res3", "res4", "res5", "goog0", "goog5", "goog13", "goog25", "goog33", "goog37"]
names = {"goog0": "conv2/3x3_reduce 64@56x56->64", "goog5": "inception_3b/1x1 256@28x28->128",
         "goog13": "inception_4b/1x1 512@14x14->160", "goog25": "inception_4e/1x1 528@14x14->256",
         "goog33": "inception_5b/1x1 832@7x7->384", "goog37": "loss1/conv 512@4x4->128",
         "res2": "res2 64@56x56->64 3x3", "res3": "res3 128@28x28->128 3x3", "res4": "res4 256@14x14->256 3x3",
         "res5": "res5 512@7x7->512 3x3"}
cats = ["tab+zero", "hdr load", "vmcnt wait", "barrier", "issue_fill", "loop", "epilogue", "tile misc"]
with open(os.path.join(out, "%s_stamp_profile.md" % tag), "w") as f:
    f.write("# In-kernel stamp profile of the shipped kernels (generated-code path), one MI355X, %s\n\n" % tag)
    f.write("`tools/evidence.sh`: the `-DESCOIN_ABLATIONS` build of the tree (`tools/mkabl.sh`) with `ESCOIN_PROF=1`, one layer at\n"
            "batch 256 per process (`tools/one_layer.py <layer> 2`, four rotating blob pairs: every launch reads from HBM).\n"
            "Cycle stamps (`s_memtime`, shader clock) of WAVE 0 of every workgroup, summed over its tiles and blocks and averaged\n"
            "over the workgroups of the LAST launch; `GHz` = shader cycles / `s_memrealtime` (100 MHz) over a workgroup's life.\n"
            "Categories: `tile misc` = everything before the first tile's table (kernel start: arguments, bias / unit-offset\n"
            "loads, first quad table, first fills) plus the gap between two tiles; `tab+zero` = next tile's quad table +\n"
            "accumulator reset; `hdr load`..`issue_fill` = block top (wait for this wave's DMA pieces, workgroup barrier, unit\n"
            "offset, fill bookkeeping, entry into the unit's code); `loop` = the walk (generated code, including the plane DMA and\n"
            "code touches it issues); `epilogue` = shift-and-sum, bias, ReLU, stores issued.  `us` = cycles / GHz.  `kernel us` =\n"
            "the same layer timed by HIP events with the product build in the same process order (200 launches).\n\n")
    f.write("| layer | kernel us (product build / stamped build) | WG life us @ GHz | " + " | ".join(cats) + " | walk by wave id (cycles; waves 0-3 dispatched first) |\n")
    f.write("|---|---|---|" + "---|" * (len(cats) + 1) + "\n")
    budget = []
    for L in layers:
        p = os.path.join(src, "stamp_%s.log" % L)
        if not os.path.exists(p):
            continue
        life = last(p, "workgroup lifetime:")
        spread = last(p, "per-WG wave spread")
        w0 = last(p, "wave0 cycles/WG")
        m = re.search(r"lifetime: (\d+) shader cycles in ([\d.]+) us -> ([\d.]+) GHz", life)
        cyc, us, ghz = int(m.group(1)), float(m.group(2)), float(m.group(3))
        vals = {k: int(v) for k, v in re.findall(r"(tab\+zero|hdr load|vmcnt wait|barrier|issue_fill|loop|epilogue|tile misc)=(\d+)", w0)}
        byw = spread.split("loop by wave id:")[1].split()
        t_prod = last(os.path.join(src, "time_%s.log" % L), "us per launch").split(":")[-1].split()
        t_abl = last(os.path.join(src, "time_abl_%s.log" % L), "us per launch").split(":")[-1].split()
        kp, ka = sorted(map(float, t_prod))[len(t_prod) // 2], sorted(map(float, t_abl))[len(t_abl) // 2]
        f.write("| %s | %.1f / %.1f | %.1f @ %.2f | %s | %s |\n" %
                (names[L], kp, ka, us, ghz,
                 " | ".join("%d (%.1f us, %d %%)" % (vals[c], vals[c] / ghz * 1e-3, round(100.0 * vals[c] / sum(vals.values()))) for c in cats),
                 " ".join(byw)))
        budget.append((L, kp, ka, us, ghz, vals))
    f.write("\n## Launch budget (us; adds up to the stamped build's launch)\n\n")
    f.write("`start-up + tile gaps` = tile misc; `tables` = tab+zero; `block tops` = hdr load + vmcnt wait + barrier + issue_fill;\n"
            "`dispatch + drain` = stamped kernel time - workgroup life (launch overhead, the last workgroup's tail, stores draining).\n\n")
    f.write("| layer | start-up + tile gaps | tables | block tops | walk | epilogue | dispatch + drain | sum = stamped launch | product build |\n|---|---|---|---|---|---|---|---|---|\n")
    for L, kp, ka, us, ghz, v in budget:
        c = lambda k: v[k] / ghz * 1e-3
        tops = c("hdr load") + c("vmcnt wait") + c("barrier") + c("issue_fill")
        scale = us / (sum(v.values()) / ghz * 1e-3)      # (stamps cover the workgroup's life up to rounding)
        parts = [c("tile misc") * scale, c("tab+zero") * scale, tops * scale, c("loop") * scale, c("epilogue") * scale, ka - us]
        f.write("| %s | %s | %.1f | %.1f |\n" % (names[L], " | ".join("%.1f" % x for x in parts), sum(parts), kp))

if not os.path.exists(os.path.join(src, "jit_abl_res2.txt")):
    print("written (stamp profile only)")
    sys.exit(0)
with open(os.path.join(out, "%s_jit_ablations.md" % tag), "w") as f:
    f.write("# Timing-only ablations of the generated-code kernel, ResNet-50 3x3 shapes @90 %%, batch 256, one MI355X, %s\n\n" % tag)
    f.write("`tools/evidence.sh`: `tools/one_layer.py <layer> 100` (four rotating blob pairs: every launch reads from HBM), us per launch,\n"
            "median of five runs of 100 launches.  Results are WRONG for every row but the first: these builds only tell where the time goes.\n"
            "`ESCOIN_JIT_ABL` (code generator): 1 = no FMAs emitted, 2 = no LDS reads, 4 = no weight moves, 8 = empty units (the units only\n"
            "stage the next block's planes and return); `ESCOIN_DBG` (kernel body): 1 = plane DMA reads zeros (no HBM reads), 2 = units never\n"
            "entered (plane DMA issued from the C++ block top instead), 3 = both, 4 = units entered but the plane DMA issued from C++,\n"
            "128 = no stores, 64 = no code touches before the first unit.\n\n")
    rows = [("ABL=0", "the product"), ("ABL=1", "no FMAs"), ("ABL=2", "no LDS reads"), ("ABL=4", "no weight moves"),
            ("ABL=3", "no FMAs, no LDS reads"), ("ABL=7", "rows and waits only"), ("ABL=8", "empty units"),
            ("DBG=1", "zero-fill plane DMA"), ("DBG=2", "no walk, DMA from the block top"), ("DBG=3", "no walk, zero fill"),
            ("DBG=4", "walk + DMA from the block top"), ("DBG=128", "no stores"), ("DBG=64", "no first-unit code touches")]
    data = {}
    for L in ("res2", "res3", "res4", "res5"):
        for line in open(os.path.join(src, "jit_abl_%s.txt" % L)):
            key = line.split()[0]
            ts = sorted(map(float, line.split(":")[-1].split()))
            data[(L, key)] = ts[len(ts) // 2]
    f.write("| build | what is left out | res2 | res3 | res4 | res5 |\n|---|---|---|---|---|---|\n")
    for key, what in rows:
        f.write("| `%s` | %s | %s |\n" % (key.replace("ABL", "ESCOIN_JIT_ABL").replace("DBG", "ESCOIN_DBG"), what,
                                        " | ".join("%.1f" % data[(L, key)] for L in ("res2", "res3", "res4", "res5"))))
print("written")
