"""bench.py's own multi-rank code -- shard spans, CSR broadcast + set_csr, global-index image seeds,
the max-over-ranks timing, the oracle self-check and the cross-rank checksum check -- driven by two
gloo ranks on the CPU with the forward stubbed by the oracle (the GPU box runs the same functions
with the HIP plans and RCCL)."""
import json
import os
import socket
import sys
import time

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


from bench_stub_backend import StubBackend as _StubBackend  # noqa: E402


def _worker(rank, world, port, out_dir, argv):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    oracle = ge.load_oracle()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    out = bench.run(bench.parse_args(argv), _StubBackend(oracle), pkg, pkg.synth, ge.load_oracle, dist)
    assert (out is None) == (rank != 0)
    if out is not None:
        with open(os.path.join(out_dir, "out.json"), "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, argv):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), argv), nprocs=2, join=True)
    return json.load(open(os.path.join(str(tmp_path), "out.json")))


def test_bench_weak_scaling_two_ranks(tmp_path):
    out = _run(tmp_path, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "lenet",
                          "--batch", "3", "--no-cpu"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 6 and out["config"]["weight_broadcast_ms"] is not None
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert out["parity_max_rel_err"] <= 1e-4                  # every rank's shard vs the oracle
    assert out["cross_rank_checksum_rel_diff"] <= 1e-5        # rank 1's images recomputed on rank 0
    assert "parity_failed" not in out
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "binding_frac"}


def test_bench_strong_scaling_uneven_shards(tmp_path):
    """--global-batch 5 over 2 ranks: 3 + 2 images; value counts all 5."""
    out = _run(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "alexnet",
                          "--global-batch", "5", "--no-cpu"])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 5
    assert abs(out["value"] - 5 / (out["ms_per_step"] * 1e-3)) <= 0.06 + 1e-3 * out["value"]   # (value is rounded to 0.1)
    assert out["parity_max_rel_err"] <= 1e-4 and out["cross_rank_checksum_rel_diff"] <= 1e-5
    assert out["config"]["layers_per_step"] == 4


def test_image_seeds_depend_on_the_global_index_only():
    sys.path.insert(0, ROOT)
    import torch
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    be = _StubBackend(None)
    s = pkg.synth.lenet_conv2(N=8)[0]
    whole = bench.device_images(be, s, 0, 0, 8)
    parts = torch.cat([bench.device_images(be, s, 0, 0, 3), bench.device_images(be, s, 0, 3, 5)])
    assert torch.equal(whole, parts)
    assert not torch.equal(whole[0], whole[1])
    assert not torch.equal(whole, bench.device_images(be, s, 1, 0, 8))


def test_host_cpu_info_respects_affinity():
    sys.path.insert(0, ROOT)
    import bench
    info = bench.host_cpu_info()
    assert 1 <= info["physical_cores"] <= info["hw_threads"] <= len(os.sched_getaffinity(0))
    assert isinstance(info["model"], str)


def test_cpu_baseline_leg_fields(monkeypatch):
    """bench.py's cpu_baseline on two tiny shapes: the reference kernels from oracle/_ref, thread
    sweep, aligned once outside the timed call, default nest and blocked best-effort leg."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    oracle = ge.load_oracle()
    synth = pkg.synth
    shapes = [synth.shape("t3", 4, 16, 14, 14, 16, 3, pad=1, sparsity=0.9, count=2),
              synth.shape("t1", 4, 16, 7, 7, 24, 1, sparsity=0.9)]
    monkeypatch.setattr(bench, "_HOST_INFO", {"hw_threads": 2, "physical_cores": 2, "model": "test", "cgroup_quota": None})
    out = bench.cpu_baseline(oracle, synth, shapes, budget_s=0.5)
    assert out["unit"] == "images/s" and out["value"] > 0 and out["cores"] in (1, 2)
    assert set(out["thread_sweep"]) == {"1", "2"} and out["single_thread_value"] > 0
    assert out["kind"] in ("reference", "port") and "sample" in out
    if out["kind"] == "reference":
        assert out["best_effort"]["value"] > 0 and "sconv_unit_stride" in out["best_effort"]["kernel"]
