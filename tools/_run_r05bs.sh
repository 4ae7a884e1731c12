#!/bin/bash
# r05bs: the two batch sweeps again on the round's final binary (powers of two: what AUTO runs; off-grid batches against the 128 / 256 line)
O=gpurun_out/r05bs; mkdir -p $O
timeout -k 10 700 python tools/batch_sweep.py res2 res3 res4 res5 goog5 goog25 goog33 alex3 > $O/batch_sweep_pow2.md 2> $O/pow2.err; echo "pow2 rc=$?"
timeout -k 10 900 python tools/batch_sweep.py --batches 96,100,192,200,250,255,257,293,300,341,384,512 res2 res3 res4 res5 goog0 goog5 goog25 goog33 alex3 > $O/batch_sweep_line.md 2> $O/line.err; echo "line rc=$?"
tail -3 $O/batch_sweep_line.md | cut -c1-300
