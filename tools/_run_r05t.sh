#!/bin/bash
# r05t: code touches on / off per GoogLeNet layer (experiments flavour)
set -o pipefail
O=gpurun_out/r05t; mkdir -p $O
export ESCOIN_LIB=$PWD/tools/ab/libescoin_exp.so
for rep in 1 2 3; do for V in 1 0; do
  ESCOIN_JIT_PREFETCH=$V timeout -k 10 300 python bench.py --no-cpu --workload googlenet > $O/goog_pref${V}_$rep.json 2> $O/err.txt || echo failed
done; done
python - <<'PY'
import json
rows = {}
for v in (1, 0):
    for rep in (1, 2, 3):
        d = json.load(open("gpurun_out/r05t/goog_pref%d_%d.json" % (v, rep)))
        for l in d["roofline"]["per_layer"]:
            rows.setdefault(l["layer"], {}).setdefault(v, []).append(l["us"])
        rows.setdefault("_step", {}).setdefault(v, []).append(d["ms_per_step"] * 1e3)
for k, r in rows.items():
    a, b = min(r[1]), min(r[0])
    print("%-24s touches %7.1f  none %7.1f  %+5.1f %%" % (k, a, b, 100 * (b / a - 1)))
PY
