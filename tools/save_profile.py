#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>_<workload>/ directory (tools/profile.sh) into small committed
files under profiles/: the rocprofv3 --stats table with kernel names shortened, and the per-kernel
PMC sums (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, per launch).

    python tools/save_profile.py gpurun_out/prof_r01_resnet50 r01_resnet50
"""
import csv
import glob
import json
import os
import re
import sys


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
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
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
    # bench.py's roofline.traffic: HBM bytes per launch of each escoin kernel.  FETCH_SIZE and
    # WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md gfx950 counts half the bytes of
    # 16 B/lane streaming reads (buffer_load ... lds included), so fetches are doubled.
    workload = tag.split("_", 1)[1] if "_" in tag else tag
    traffic = {}
    for k, cs in summary.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            traffic[k] = int((2 * cs["FETCH_SIZE"]["per_launch"] + cs["WRITE_SIZE"]["per_launch"]) * 1024)
            traffic["_fetch_kib_per_launch"] = cs["FETCH_SIZE"]["per_launch"]
            traffic["_write_kib_per_launch"] = cs["WRITE_SIZE"]["per_launch"]
    if traffic:
        import subprocess
        import time
        try:
            traffic["_commit"] = subprocess.run(["git", "log", "-1", "--format=%h"], stdout=subprocess.PIPE,
                                                stderr=subprocess.DEVNULL, timeout=10).stdout.decode().strip() or None
        except Exception:
            traffic["_commit"] = None
        traffic["_saved"] = time.strftime("%Y-%m-%d %H:%M:%S")
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
