#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes for the bench.
#   gpurun -- 'bash tools/profile.sh r01 resnet50'
# Writes gpurun_out/prof_<tag>/...; tools/save_profile.py condenses it into profiles/.
set -u
TAG=${1:-r01}
WL=${2:-resnet50}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=${PROF_OUT:-gpurun_out}/prof_${TAG}_${WL}
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --workload $WL --steps 5 --warmup 2 --no-cpu --no-extras"
# (PMC passes: inputs generated with one call per layer, see bench.py device_images)
export ESCOIN_BENCH_BULK_INPUTS=0
# the kernel-stats pass runs bench.py with its default step counts (the command the driver times)
# (--no-extras: the other configurations and the sparsity sweep that the default ResNet run appends launch the SAME kernel
#  names at other sparsities; without them the per-kernel averages are the headline's.  FULL=1 adds a pass of the whole command.)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload $WL --no-cpu --no-extras > $OUT/bench_stats.json 2> $OUT/bench_stats.log
if [ "${FULL:-0}" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $OUT/bench_full.json 2> $OUT/bench_full.log
fi
export ESCOIN_BENCH_BULK_INPUTS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq2.log
# (round 5) where the L2s' read requests go: TCC_EA0_RDREQ = all, _DRAM = those addressed to device memory (as opposed to GMI / IO).
# NOTE: the Infinity Cache sits BEHIND the fabric, so neither counter tells a hit in it from an HBM read -- recorded to show exactly that
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $OUT/pmc_ea -- python3 $ARGS > /dev/null 2> $OUT/pmc_ea.log
# keep what tools/save_profile.py reads (gpurun copies back at most 64 MiB)
find $OUT -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
find $OUT -type f -name "*.log" -size +200k -delete
echo done > $OUT/done.txt
find $OUT -name "*.csv" | head -20
