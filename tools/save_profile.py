#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>_<workload>/ directory (tools/profile.sh) into small committed
files under profiles/: the rocprofv3 --stats table with kernel names shortened, and the per-kernel
PMC sums (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, per launch).

    python tools/save_profile.py gpurun_out/prof_r01_resnet50 r01_resnet50 [bench.json]

With the bench line of the same workload (its roofline.per_layer: layer shapes in launch order and how
often each repeats) the traffic is also broken down per LAYER SHAPE from the dispatch order -- every
3x3 layer of the ResNet set runs under one kernel name, and res2's band-halo re-read should be visible
on its own.
"""
import csv
import glob
import json
import os
import re
import sys


HELPER = re.compile(r"locator|code_copy|sk_clear")


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(escoin::)?(escoin_\w+(<[^>]*>)?)", name)
    if m:
        return m.group(2)
    return (name[:60] + "...") if len(name) > 63 else name


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    # gpurun merges every run's files into the same local directory: take the newest of each kind
    def newest(pattern):
        files = glob.glob(pattern, recursive=True)
        return [max(files, key=os.path.getmtime)] if files else []

    stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open("profiles/%s_kernel_stats.csv" % tag, "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py ... (tools/profile.sh)\n")
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:12]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                            r["Percentage"], r["MinNs"], r["MaxNs"]])
    out = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_ea"):
        for fn in newest(os.path.join(src, sub, "**", "*counter_collection.csv")):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                if not k.startswith("escoin"):
                    continue
                d = out.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
                d[0] += float(r["Counter_Value"])
                d[1] += 1
    summary = {}
    for k, cs in out.items():
        summary[k] = {c: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(1, v[1])} for c, v in cs.items()}
    with open("profiles/%s_pmc.json" % tag, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    # ---- per layer shape, from the dispatch order: the escoin launches of a step come in the order of
    # the bench line's per_layer list (each entry `count` times)
    per_layer = None
    if len(sys.argv) > 3:
        try:
            bl = json.load(open(sys.argv[3]))["roofline"]["per_layer"]
            order = []
            for l in bl:
                order += [l["layer"]] * int(l["count"])
            per_layer = {}
            for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
                for fn in newest(os.path.join(src, sub, "**", "*counter_collection.csv")):
                    rows = [r for r in csv.DictReader(open(fn)) if r["Counter_Name"] == cname and short(r["Kernel_Name"]).startswith("escoin")
                            and not HELPER.search(r["Kernel_Name"])]
                    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
                    if len(rows) % len(order):
                        print("per-layer breakdown skipped: %d escoin dispatches are not whole steps of %d" % (len(rows), len(order)))
                        per_layer = None
                        break
                    for i, r in enumerate(rows):
                        d = per_layer.setdefault(order[i % len(order)], {}).setdefault(cname, [0.0, 0, short(r["Kernel_Name"])])
                        d[0] += float(r["Counter_Value"])
                        d[1] += 1
                if per_layer is None:
                    break
        except Exception as e:      # noqa: BLE001 (a profile without the breakdown is still a profile)
            print("per-layer breakdown skipped:", e)
            per_layer = None
    # bench.py's roofline.traffic: HBM bytes per launch of each escoin kernel.  FETCH_SIZE and
    # WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md gfx950 counts half the bytes of
    # 16 B/lane streaming reads (buffer_load ... lds included), so fetches are doubled.
    workload = tag.split("_", 1)[1] if "_" in tag else tag
    traffic = {}
    for k, cs in summary.items():
        if HELPER.search(k):      # (not a layer's launch: the locator / code copy of WeightAlign, stream-K's flag kernel)
            continue
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            traffic[k] = int((2 * cs["FETCH_SIZE"]["per_launch"] + cs["WRITE_SIZE"]["per_launch"]) * 1024)
            traffic["_fetch_kib_per_launch"] = cs["FETCH_SIZE"]["per_launch"]
            traffic["_write_kib_per_launch"] = cs["WRITE_SIZE"]["per_launch"]
    if traffic:
        import subprocess
        import time
        # which configuration the counters were collected on: bench.py only uses the file for exactly this one
        try:
            cfg = json.load(open(sys.argv[3]))["config"] if len(sys.argv) > 3 else {}
            if "per_gpu_batch" in cfg:
                traffic["_batch"] = int(cfg["per_gpu_batch"])
                traffic["_sparsity_pct"] = int(cfg["sparsity_pct"])
        except Exception:
            pass
        try:
            traffic["_commit"] = subprocess.run(["git", "log", "-1", "--format=%h"], stdout=subprocess.PIPE,
                                                stderr=subprocess.DEVNULL, timeout=10).stdout.decode().strip() or None
        except Exception:
            traffic["_commit"] = None
        traffic["_saved"] = time.strftime("%Y-%m-%d %H:%M:%S")
        if per_layer:
            traffic["_per_layer"] = {
                name: {"kernel": cs["FETCH_SIZE"][2],
                       "hbm_bytes_per_launch": int((2 * cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] + cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1]) * 1024),
                       "fetch_kib": cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1], "write_kib": cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1]}
                for name, cs in per_layer.items() if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs}
        traffic["_note"] = ("HBM bytes per launch of the dominant kernel, averaged over the launches of "
                            "the bench steps: (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 "
                            "--pmc passes (tools/profile.sh, profiles/%s_pmc.json); FETCH_SIZE doubled "
                            "per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16 B/lane "
                            "streaming reads); WRITE_SIZE as is." % tag)
        with open("profiles/traffic_%s.json" % workload, "w") as f:
            json.dump(traffic, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True)[:600])


if __name__ == "__main__":
    main()
