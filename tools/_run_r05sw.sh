#!/bin/bash
# r05sw: the sparsity sweep (north star: 60-95 %) and the dense / sparse crossover table on the round's final binary
O=gpurun_out/r05sw; mkdir -p $O
SWEEP_CPU_BUDGET=6 timeout -k 10 1100 python tools/sparsity_sweep.py > $O/sparsity_sweep.md 2> $O/sweep.err; echo "sweep rc=$?"
timeout -k 10 1000 python tools/crossover.py --sparsities 0,10,20,30,40,50,60,70,80,85,90,95 --json $O/crossover.json > $O/crossover.md 2> $O/crossover.err; echo "crossover rc=$?"
tail -5 $O/sparsity_sweep.md; tail -12 $O/crossover.md
