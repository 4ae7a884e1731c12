#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group over `python3 <script args>` and
# prints per-kernel per-launch averages.   bash tools/pmc_pass.sh <outdir> "<counters;counters;...>" <python args...>
set -u
OUT=$1; shift
GROUPS_=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
IFS=';' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $g --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
  echo "pass $i done" >> "$OUT/progress.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "escoin" not in k:
            continue
        k = k.split("(")[0][-60:]
        d[k][r["Counter_Name"]][0] += float(r["Counter_Value"]); d[k][r["Counter_Name"]][1] += 1
for k, cs in d.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.1f  (%d launches)" % (c, v[0] / max(1, v[1]), v[1]))
PY
