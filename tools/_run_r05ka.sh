#!/bin/bash
# r05ka: where the HIP runtime puts kernel arguments (HIP_FORCE_DEV_KERNARG): launch latency of the short pointwise launches
O=gpurun_out/r05ka; mkdir -p $O; : > $O/ka.txt
for rep in 1 2; do for v in unset 0 1; do for wl in googlenet resnet50; do
  if [ $v = unset ]; then E=""; else E="HIP_FORCE_DEV_KERNARG=$v"; fi
  echo "$wl $v $(env $E timeout -k 10 300 python bench.py --no-cpu --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")" | tee -a $O/ka.txt
done; done; done
