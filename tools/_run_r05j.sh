#!/bin/bash
# r05j: the round's profile pass on the final binary: rocprofv3 kernel stats + PMC passes + bench lines, all four workloads
set -u
O=gpurun_out/r05j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in resnet50 googlenet alexnet lenet; do
  PROF_OUT=gpurun_out timeout -k 10 600 bash tools/profile.sh r05 $wl > $O/profile_$wl.log 2>&1; echo "profile $wl rc=$?"
  timeout -k 10 300 python bench.py --workload $wl > gpurun_out/r05_bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_resnet50_driver_command.json 2> $O/bench_driver.err; echo "driver cmd rc=$?"
timeout -k 10 300 python bench.py --workload googlenet --stream-stores --no-cpu > gpurun_out/r05_bench_googlenet_stream_stores.json 2>/dev/null
timeout -k 10 300 python bench.py --workload googlenet --streams 2 --no-cpu > gpurun_out/r05_bench_googlenet_2streams.json 2>/dev/null
timeout -k 10 300 python bench.py --workload googlenet --streams 2 --stream-stores --no-cpu > gpurun_out/r05_bench_googlenet_2streams_stream_stores.json 2>/dev/null
timeout -k 10 300 python bench.py --global-batch 2048 --no-cpu > gpurun_out/r05_bench_resnet50_global_batch_2048_one_gpu.json 2>/dev/null; echo "gb2048 rc=$?"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python bench.py --gpus 1 --force-dist --no-cpu > gpurun_out/r05_bench_resnet50_1rank_rccl.json 2> $O/bench_rccl.err; echo "rccl rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --no-cpu > gpurun_out/r05_bench_resnet50_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks.err; echo "2 ranks rc=$?"
ls gpurun_out/prof_r05_*/ | head -40
