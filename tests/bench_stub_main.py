"""CPU test-suite entry of bench.py: the SAME launcher, rendezvous, sharding, broadcast, checks and
reporting as `python bench.py`, with the device side replaced by tests/bench_stub_backend.py (forward
by the CPU oracle).  The product benchmark has no such switch: only this script, which lives in
tests/, can put the stub in.  Its bench line is marked `"test_backend": true` and is not a measurement.

    python tests/bench_stub_main.py --gpus 2 --workload lenet --batch 3 --steps 1 --warmup 0 --no-cpu
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bench  # noqa: E402


def _factory(local_rank):
    import bench_stub_backend
    return bench_stub_backend.make_backend(local_rank)


if __name__ == "__main__":
    bench.main(backend_factory=_factory, script=os.path.abspath(__file__))
