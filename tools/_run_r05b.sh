#!/bin/bash
# r05 second GPU call: pointwise-ends bounds, batch sweep, skewed sparsity, RCCL path on one rank, rule re-check
set -o pipefail
O=gpurun_out/r05b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== bench (driver command)"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { echo bench failed; tail -5 $O/bench_driver.err; exit 1; }
python - <<'PY'
import json; d=json.load(open("gpurun_out/r05b/bench_driver.json")); print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "align", d["weight_align_ms"]["total"], d["weight_align_ms"]["max_per_layer"], d["weight_align_ms"]["first_load_ms"], "parity", d["parity_max_rel_err"])
PY
echo "== rccl"; HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python bench.py --gpus 1 --force-dist --no-cpu > $O/bench_resnet50_1rank_rccl.json 2> $O/bench_resnet50_1rank_rccl.err || { echo "force-dist failed"; tail -20 $O/bench_resnet50_1rank_rccl.err; }
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_selfcheck.py > $O/rccl_selfcheck.txt 2>&1 || echo "selfcheck failed"
tail -2 $O/rccl_selfcheck.txt
echo "== pointwise ends"; timeout -k 10 600 bash tools/pointwise_ends.sh $O/ends > $O/ends.log 2>&1 || { echo ends failed; tail -5 $O/ends.log; }
echo "== batch sweep"; timeout -k 10 900 python tools/batch_sweep.py --batches 96,100,192,200,250,255,257,293,300,341,384,512 res2 res3 res4 res5 goog0 goog5 goog25 goog33 alex3 > $O/batch_sweep_line.md 2> $O/batch_sweep_line.err || { echo sweep failed; tail -5 $O/batch_sweep_line.err; }
echo "== skew"
for D in uniform i ii iii; do
  timeout -k 10 300 python bench.py --no-cpu --sparsity-dist $D > $O/bench_resnet50_dist_$D.json 2> $O/bench_resnet50_dist_$D.err || { echo "skew $D failed"; tail -5 $O/bench_resnet50_dist_$D.err; }
  timeout -k 10 300 python bench.py --no-cpu --workload alexnet --sparsity-dist $D > $O/bench_alexnet_dist_$D.json 2> $O/bench_alexnet_dist_$D.err || { echo "skew alex $D failed"; }
done
echo "== mall probe"; timeout -k 10 120 tools/probes/probe_mall_share > $O/probe_mall_share.txt 2>&1 || echo "mall probe failed"
echo "== rule"; timeout -k 10 600 python tools/small_launch_fit.py > $O/small_launch_fit.jsonl 2> $O/small_launch_fit.err || echo "fit failed"
ls $O
