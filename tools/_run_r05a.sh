#!/bin/bash
# r05 first GPU call: GPU suite on the refactored product build, counter list, small-launch fit data
set -o pipefail
O=gpurun_out/r05a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt
tail -5 $O/pytest_gpu.log
(rocprofv3 -L 2>/dev/null | grep -i "TCC_EA\|DRAM\|MALL\|TCC_REQ\|TCC_HIT\|TCC_MISS" | sort -u | head -150) > $O/counters.txt || true
timeout -k 10 600 python tools/small_launch_fit.py > $O/small_launch_fit.jsonl 2> $O/small_launch_fit.err || echo "fit failed"
wc -l $O/small_launch_fit.jsonl
