#!/usr/bin/env python3
"""Condenses round 5's GPU-box outputs (gpurun_out/r05*/, written by tools/_run_r05*.sh) into the tracked files under
profiles/.  Pure text processing; run from the repo root:  python tools/make_r05_reports.py"""
import collections
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda *p: os.path.join(ROOT, "gpurun_out", *p)
P = lambda *p: os.path.join(ROOT, "profiles", *p)

LAYER = {"goog0": "conv2/3x3_reduce 64@56x56->64", "goog1": "inception_3a/1x1 192@28x28->64", "goog2": "3a/3x3_reduce 192@28x28->96",
         "goog3": "3a/5x5_reduce 192@28x28->16", "goog4": "3a/pool_proj 192@28x28->32", "goog5": "inception_3b/1x1 256@28x28->128",
         "goog6": "3b/3x3_reduce 256@28x28->128", "goog7": "3b/5x5_reduce 256@28x28->32", "goog8": "3b/pool_proj 256@28x28->64",
         "goog9": "inception_4a/1x1 480@14x14->192", "goog12": "4a/pool_proj 480@14x14->64", "goog25": "inception_4e/1x1 528@14x14->256",
         "goog33": "inception_5b/1x1 832@7x7->384"}


def times_of(line):
    m = re.search(r"launches\): (.*)", line)
    return [float(x) for x in m.group(1).split()] if m else None


def notes(name):
    """Hand-written reading of a table, kept beside it (profiles/<name>_notes.md) and appended to the generated file."""
    fn = P(name + "_notes.md")
    return ("\n" + open(fn).read()) if os.path.exists(fn) else ""


def pointwise_ends():
    src = G("r05b", "ends", "times.txt")
    if not os.path.exists(src):
        return
    t = collections.defaultdict(dict)
    for line in open(src):
        m = re.match(r"(goog\d+) (product|abl ESCOIN_DBG=(\d+)) : ", line)
        if m and times_of(line):
            t[m.group(1)][m.group(3) or "product"] = min(times_of(line))
    stamps = {}
    for fn in glob.glob(G("r05b", "ends", "stamp_*.log")):
        key = os.path.basename(fn)[6:-4]
        d = {}
        for line in open(fn):
            m = re.search(r"lifetimes: min ([\d.]+) us, max ([\d.]+) us; starts spread over ([\d.]+) us; first start to last end ([\d.]+)", line)
            if m:
                d.update(min=float(m.group(1)), max=float(m.group(2)), span=float(m.group(4)))
            m = re.search(r"workgroup lifetime: \d+ shader cycles in ([\d.]+) us -> ([\d.]+) GHz", line)
            if m:
                d.update(mean=float(m.group(1)), ghz=float(m.group(2)))
            m = re.search(r"mean lifetime by XCD \(us\): ([\d. ]+)\|", line)
            if m:
                d["xcd"] = [float(x) for x in m.group(1).split()]
            m = re.search(r"wave0 cycles/WG.*?:(.*)", line)
            if m:
                d["phases"] = dict((k.strip(), int(v)) for k, v in re.findall(r"([a-z+ _]+)=(\d+)\(", m.group(1)))
        stamps[key] = d
    alg = {"goog0": 411.0, "goog5": 308.3, "goog25": 157.4, "goog33": 61.0}      # MB per launch at batch 256
    with open(P("r05_pointwise_ends.md"), "w") as f:
        f.write("# What the ENDS of an HBM-bound pointwise launch cost, and what could give them back (r05, one MI355X)\n\n")
        f.write("`tools/pointwise_ends.sh` (`tools/_run_r05b.sh`): one layer per image size at batch 256, 95 % sparsity, HBM-cold (four rotating\n"
                "bottom / top pairs, `tools/one_layer.py`, best of 5 x 200 launches).  `abl` = the `-DESCOIN_ABLATIONS` flavour\n"
                "(`tools/mkabl.sh`; the wrong-result switches exist only there): `ESCOIN_DBG=128` issues NO STORES (what a perfectly hidden\n"
                "epilogue could give at most -- it also removes the write traffic itself, which nothing can), `ESCOIN_DBG=2` runs NO WALK,\n"
                "130 = both.  Stamps (`ESCOIN_PROF=1`): lifetimes of the 256 workgroups of the last launch.\n\n")
        f.write("| layer | product us | abl us | no stores | no walk | neither | alg. MB | copy at 6.3 TB/s + 8 us | workgroup life min / mean / max us | first start -> last end | launch - mean life |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
        for L in ("goog0", "goog5", "goog25", "goog33"):
            if L not in t:
                continue
            d, s = t[L], stamps.get(L, {})
            bound = alg[L] / 6.3e3 * 1e3 + 8.0
            f.write("| %s | %.1f | %.1f | %.1f (%+.0f %%) | %.1f (%+.0f %%) | %.1f | %.0f | %.1f | %.1f / %.1f / %.1f | %.1f | %.1f |\n" % (
                LAYER[L], d["product"], d["0"], d["128"], 100 * (d["128"] / d["0"] - 1), d["2"], 100 * (d["2"] / d["0"] - 1), d["130"], alg[L], bound,
                s.get("min", 0), s.get("mean", 0), s.get("max", 0), s.get("span", 0), d["0"] - s.get("mean", 0)))
        f.write("\nMean workgroup lifetime by XCD (linear workgroup id % 8), us -- with the stores / without:\n\n")
        for L in ("goog0", "goog5"):
            if L in stamps and "nostore_" + L in stamps:
                f.write("* %s: %s / %s\n" % (LAYER[L], " ".join("%.1f" % v for v in stamps[L].get("xcd", [])),
                                             " ".join("%.1f" % v for v in stamps["nostore_" + L].get("xcd", []))))
        f.write("\nWave 0's cycles per workgroup by phase (stamped build):\n\n| layer | " + " | ".join(
            ["tab+zero", "block tops", "walk", "epilogue", "start-up + tile gaps"]) + " |\n|---|---|---|---|---|---|\n")
        for L in ("goog0", "goog5", "goog25", "goog33"):
            ph = stamps.get(L, {}).get("phases")
            if not ph:
                continue
            tot = float(sum(ph.values()))
            tops = ph.get("hdr load", 0) + ph.get("vmcnt wait", 0) + ph.get("barrier", 0) + ph.get("issue_fill", 0)
            f.write("| %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.0f %% |\n" % (LAYER[L], 100 * ph.get("tab+zero", 0) / tot, 100 * tops / tot,
                    100 * ph.get("loop", 0) / tot, 100 * ph.get("epilogue", 0) / tot, 100 * ph.get("tile misc", 0) / tot))
        f.write(notes("r05_pointwise_ends"))


def half_workgroups():
    rows = []
    for src in (G("r05c", "waves4.txt"), G("r05d", "half.txt")):
        if not os.path.exists(src):
            continue
        cur = None
        for line in open(src):
            m = re.match(r"(goog\d+) \[(.*?)\] : (.*)", line)
            if m:
                cur = (m.group(1), m.group(2), m.group(3))
                continue
            ts = times_of(line) or ([float(x) for x in line.split()] if re.match(r"^[\d. ]+$", line.strip()) and line.strip() else None)
            if ts and cur:
                g = re.search(r"G=(\d+).*?n_ocblk=(\d+).*?oc_waves=(\d+).*?bands=(\d+).*?n_icb=(\d+)", cur[2])
                rows.append((os.path.basename(os.path.dirname(src)), cur[0], cur[1], min(ts), g.groups() if g else None))
                cur = None
    if not rows:
        return
    with open(P("r05_half_workgroups.md"), "w") as f:
        f.write("# Two 4-wave workgroups per CU on the HBM-bound pointwise layers (r05, one MI355X)\n\n"
                "`tools/_run_r05c.sh` / `tools/_run_r05d.sh`: the experiments flavour (`tools/mkabl.sh exp`), one layer per process, batch 256, 95 %,\n"
                "HBM-cold (four rotating pairs), best of 5 x 200 launches.  r05c forces four waves through the tuning switches (every layer, three\n"
                "buffer shapes); r05d is the shipped rule (`sconv_tiled.hip`, half-workgroup rule) switched off (`ESCOIN_HALF_WG=0`) and as it is\n"
                "(in r05d the rule still admitted up to 192 channels with one quad per lane: the 128-channel rows are why it no longer does).\n\n")
        f.write("| run | layer | switches | us | G | columns | waves per workgroup | bands | blocks per tile |\n|---|---|---|---|---|---|---|---|---|\n")
        for run, L, knobs, us, g in rows:
            f.write("| %s | %s | `%s` | %.1f | %s |\n" % (run, LAYER.get(L, L), knobs or "(8 waves, as shipped in round 4)", us, " | ".join(g) if g else " | | | | "))
        f.write(notes("r05_half_workgroups"))


def small_launch():
    src = G("r05b", "small_launch_fit.jsonl")
    if not os.path.exists(src):
        return
    rows = [json.loads(l) for l in open(src) if l.strip().startswith("{")]
    pw = [r for r in rows if r["K"] == 1]
    by = collections.defaultdict(dict)
    for r in pw:
        by[(r["C"], r["HW"], r["M"])][r["N"]] = r
    worst, over = 0.0, 0
    with open(P("r05_small_launch_fit.md"), "w") as f:
        f.write("# Small launches: generated code vs the generic kernel, and KERNEL_AUTO's rule (r05, one MI355X)\n\n"
                "`tools/small_launch_fit.py`: 1 .. 32 images of every distinct GoogLeNet 1x1 shape @95 %, both kernels FORCED (plan option `kernel`),\n"
                "60 launches, best of three, one reused bottom / top pair (an image-by-image caller: the reference's SCONV mode, `conv_layer.cu:16-26`).\n"
                "Cell = `code us / generic us -> what KERNEL_AUTO's rule ran (its us)`; `!` marks a pick more than 10 % behind the faster kernel.\n"
                "The rule (`escoin_capi.hip`): generic ~ max(7.0, 5.8 + 0.125 r, 6.2 + waves (0.143 + 0.0473 r) / 1000) us, r = nonzeros per output row,\n"
                "waves = N M ceil(OH OW / 64); code ~ 7.6 + (0.6 chained | 1.1 per-block calls) x blocks per tile + 0.1 x MB of blobs; the lower estimate wins;\n"
                "only pointwise launches under 64 MFLOP that fit one round of workgroups are considered.  A function of options, weights and batch:\n"
                "`tests/test_gpu_parity.py::test_kernel_auto_is_the_same_in_fresh_processes` resolves seven boundary cases in 20 fresh processes.\n\n")
        f.write("| layer | " + " | ".join("N=%d" % n for n in (1, 2, 4, 8, 16, 32)) + " |\n|---|" + "---|" * 6 + "\n")
        for k in sorted(by, key=lambda k: (-k[1], k[0], k[2])):
            cells = []
            for n in (1, 2, 4, 8, 16, 32):
                r = by[k].get(n)
                if not r:
                    cells.append("")
                    continue
                best = min(r["code_us"], r["generic_us"])
                picked = r["generic_us"] if r["auto_kernel"] == "generic" else r["code_us"]
                reg = picked / best
                worst = max(worst, reg)
                over += reg > 1.10
                cells.append("%.1f / %.1f -> %s (%.1f)%s" % (r["code_us"], r["generic_us"], r["auto_kernel"], r["auto_us"], " !" if reg > 1.10 else ""))
            f.write("| %d@%dx%d->%d | " % (k[0], k[1], k[1], k[2]) + " | ".join(cells) + " |\n")
        f.write("\n%d cells; the rule's pick (by the forced kernels' times of this run) is at worst %.1f %% behind the faster kernel, %d cell(s) more than 10 %%.\n" % (
            len(pw), 100 * (worst - 1), over))
        r3 = [r for r in rows if r["K"] == 3]
        if r3:
            f.write("\n3x3 layers for reference (never considered by the rule: generated code at every batch): " + "; ".join(
                "%s N=%d code %.1f generic %.1f" % (r["layer"].split("_")[0], r["N"], r["code_us"], r["generic_us"]) for r in r3 if r["N"] in (1, 8, 32)) + "\n")


def skew():
    rows = []
    for wl in ("resnet50", "alexnet"):
        for d in ("uniform", "i", "ii", "iii"):
            for run in ("r05f", "r05b"):
                fn = G(run, "bench_%s_dist_%s.json" % (wl, d))
                try:
                    rows.append((wl, d, json.load(open(fn)), run))
                    break
                except Exception:
                    pass
    if not rows:
        return
    name = {"uniform": "uniform (the BASELINE configs)", "i": "(i) per-output-channel density ~U(0, 2d)", "ii": "(ii) 20 % of the input channels all zero",
            "iii": "(iii) 10 % of the filters all zero, 5 % of the rows at 4d"}
    base = {wl: [r for r in rows if r[0] == wl and r[1] == "uniform"][0][2]["ms_per_step"] for wl in set(r[0] for r in rows)}
    with open(P("r05_skew.md"), "w") as f:
        f.write("# Skewed sparsity, as a pruned model has it (r05, one MI355X)\n\n"
                "`python bench.py --no-cpu [--workload alexnet] --sparsity-dist {uniform,i,ii,iii}`: the same total nonzero count per layer\n"
                "(`synth.pruned_weights(s, seed, dist)`: algorithmic bytes and flops unchanged), laid out as a magnitude-pruned model's would be -- the nets the\n"
                "reference runs are SkimCaffe-pruned (`run.sh:14`).  `deal` = WeightAlign's channel deal as `escoin_plan_stat` reports it: slowest wave / mean\n"
                "wave per block, barrier-weighted over all blocks (worst single block in brackets); parity = `parity_max_rel_err` of the timed outputs vs the oracle.\n"
                "GPU parity per distribution: `tests/test_gpu_parity.py::test_skewed_sparsity_distributions`.\n\n")
        f.write("| workload | distribution | ms per step | vs uniform | parity | generated code MB | WeightAlign ms | per layer shape: us, deal |\n|---|---|---|---|---|---|---|---|\n")
        for wl, d, j, run in rows:
            deal = {c["layer"]: c for c in j.get("channel_deal", [])}
            per = "; ".join("%s %.1f us, %.3f (%.2f)" % (l["layer"].replace("_branch2b", ""), l["us"], deal.get(l["layer"], {}).get("slowest_over_mean", 0),
                                                         deal.get(l["layer"], {}).get("worst_block", 0)) for l in j["roofline"]["per_layer"])
            f.write("| %s | %s | %.4f | %+.1f %% | %.1e | %.1f | %.0f | %s |\n" % (wl, name[d], j["ms_per_step"], 100 * (j["ms_per_step"] / base[wl] - 1),
                    j["parity_max_rel_err"], j["generated_code_bytes"] / 1e6, j["weight_align_ms"]["total"], per))
        f.write(notes("r05_skew"))


def batch_sweep():
    after, before, oracle = G("r05h", "batch_sweep_line.md"), G("r05b", "batch_sweep_line.md"), G("r05g", "tiling_oracle.txt")
    if not os.path.exists(after):
        return

    def quant(table):
        """Adds, per cell, what whole tiles on whole workgroups cost: rounds needed / rounds an ideal split would need."""
        out = []
        for line in table.splitlines():
            if not line.startswith("| ") or line.startswith("| layer") or line.startswith("|---"):
                out.append(line)
                continue
            cells = line.split(" | ")
            new = [cells[0]]
            for c in cells[1:]:
                m = re.search(r"\[(\d+)x(\d+) t, (\d+) col\]", c)
                if m:
                    tiles, cols = int(m.group(2)), int(m.group(3))
                    gx = max(1, 256 // cols)
                    rounds = -(-tiles // gx)
                    c = c.rstrip(" |") + " q=%.2f" % (tiles / float(gx) / rounds)
                new.append(c)
            out.append(" | ".join(new) + (" |" if not new[-1].rstrip().endswith("|") else ""))
        return "\n".join(out)

    with open(P("r05_batch_sweep.md"), "w") as f:
        f.write("# Batches the tilings were not tuned for (r05, one MI355X)\n\n"
                "`python tools/batch_sweep.py --batches 96,100,192,200,250,255,257,293,300,341,384,512 <layers>`: every plan created for ITS batch, HBM-cold\n"
                "(rotating bottom / top pairs), 40 launches, best of three.  Cell = `us (images/s relative to the straight line through this run's N = 128 and N = 256\n"
                "points) [images per tile x tiles, workgroup columns] q = tiles / (workgroups per column x rounds)` -- q is what whole tiles on whole\n"
                "workgroups can deliver at best: q = 0.67 means the last of three rounds is one tile wide.\n\n"
                "## After round 5's change to the tiling cost model (tilings that cannot chain are priced a quarter up)\n\n")
        f.write(quant(open(after).read().split("\n\n", 1)[1] if "\n\n" in open(after).read() else open(after).read()) + "\n\n")
        if os.path.exists(before):
            f.write("## Before (round 4's model, `gpurun_out/r05b`)\n\n")
            f.write(quant(open(before).read().split("\n\n", 1)[1]) + "\n\n")
        if os.path.exists(oracle):
            f.write("## What the model could have chosen (`tools/tiling_oracle.py`, experiments flavour: every workgroup-column count x images per tile forced; round 4's model as AUTO)\n\n```\n")
            keep = []
            for line in open(oracle):
                if " AUTO: " in line or "better than AUTO" in line:
                    keep.append(line.rstrip())
            f.write("\n".join(keep) + "\n```\n")
        f.write(notes("r05_batch_sweep"))


def copy_files():
    for src, dst in ((G("r05b", "bench_resnet50_1rank_rccl.json"), P("r05_bench_resnet50_1rank_rccl.json")),
                     (G("r05b", "rccl_selfcheck.txt"), P("r05_rccl_selfcheck.txt"))):
        if os.path.exists(src):
            open(dst, "w").write(open(src).read())


if __name__ == "__main__":
    pointwise_ends()
    half_workgroups()
    small_launch()
    skew()
    batch_sweep()
    copy_files()
    print("written:", sorted(os.path.basename(p) for p in glob.glob(P("r05_*"))))
