#!/bin/bash
# r05sw2: the re-fitted dense / sparse model: GPU suite, then the crossover table again (AUTO's picks and times)
O=gpurun_out/r05sw2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout -k 10 1000 python tools/crossover.py --sparsities 0,10,20,30,40,50,60,70,80,85,90,95 --json $O/crossover.json > $O/crossover.md 2> $O/crossover.err; echo "crossover rc=$?"
grep -c "!" $O/crossover.md
