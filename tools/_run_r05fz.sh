#!/bin/bash
# r05fz: wide random parity sweep (tools/fuzz_parity.py) on the product library: cases per seed, then the seeds
O=gpurun_out/r05fz; mkdir -p $O
n=$1; shift
for seed in "$@"; do
  timeout -k 10 1000 python tools/fuzz_parity.py $n $seed > $O/fuzz_$seed.txt 2> $O/fuzz_err_$seed.txt; echo "seed $seed rc=$?"
  grep -c FAIL $O/fuzz_$seed.txt; grep FAIL $O/fuzz_$seed.txt | head -20; grep "^cases" $O/fuzz_$seed.txt
done
