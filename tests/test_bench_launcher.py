"""`python bench.py --gpus N` as ONE command: the parent starts its own ranks (child processes of
torch.distributed.run), never touches the GPU, relays rank 0's JSON line and the worst exit code --
the counterpart of the reference's single `caffe` command starting a worker per GPU
(tools/caffe.cpp:254-256, parallel.cpp:328-358).  Runs here on two gloo ranks with the test backend
(tests/bench_stub_main.py -> tests/bench_stub_backend.py: forward by the oracle; bench.py itself has no
switch that could point it at the oracle)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_MAIN = os.path.join(ROOT, "tests", "bench_stub_main.py")


def _run_one_command(extra, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, STUB_MAIN] + extra, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    return p.returncode, lines, p.stderr.decode()


def test_one_command_starts_two_ranks_and_prints_one_json_line():
    rc, lines, err = _run_one_command(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "lenet",
                                       "--batch", "3", "--no-cpu"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                 # ONE line on stdout, everything else on stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 6
    assert out["weight_broadcast_ms"] is not None and out["weight_broadcast_ms"] >= 0
    assert out["dist_backend"] == "gloo" and out["test_backend"] is True
    assert out["parity_max_rel_err"] <= 1e-4 and out["cross_rank_checksum_rel_diff"] <= 1e-5
    assert "torch.distributed.run" in err         # the launcher said what it started


def test_one_command_strong_scaling_global_batch():
    rc, lines, err = _run_one_command(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "lenet",
                                       "--global-batch", "5", "--no-cpu"])
    assert rc == 0, err[-2000:]
    out = json.loads(lines[-1])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 5 and out["n_ranks_seen"] == 2


def test_product_bench_has_no_test_backend_switch():
    """a bench that an environment variable can point at the oracle is one variable away from
    a fake number.  bench.py's own main() knows one backend."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "ESCOIN_BENCH_TEST_BACKEND" not in src and "import bench_stub" not in src
    assert "getenv" not in src.split("def main(")[1] and "environ.get(\"ESCOIN" not in src.split("def main(")[1]


def test_scale_command_line_eight_ranks_global_batch_2048():
    """The driver's SCALE run at its largest point -- `bench.py --gpus 8 --global-batch 2048`, BASELINE
    configs[3] -- end to end as one command: eight gloo ranks, 8 x 256-image shards of the ResNet-50 set,
    one JSON line, n_ranks_seen 8 (the stub computes only the images the two checks read)."""
    rc, lines, err = _run_one_command(["--gpus", "8", "--global-batch", "2048", "--steps", "1", "--warmup", "0",
                                       "--repeats", "1", "--settle-ms", "0", "--no-cpu"],
                                      env_extra={"ESCOIN_STUB_CHECKED_IMAGES_ONLY": "1", "OMP_NUM_THREADS": "1"},
                                      timeout=900)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 2048 and "batch 256/GPU" in out["config"]["workload"]
    assert out["config"]["layers_per_step"] == 16 and "ResNet-50" in out["config"]["workload"]
    assert out["parity_max_rel_err"] <= 1e-4 and out["cross_rank_checksum_rel_diff"] <= 1e-5
    assert out["weight_broadcast_ms"] is not None and out["weight_receive_ms"]["total"] >= 0
    assert out["repeats"] == 1 and out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]


def test_launcher_relays_a_failing_rank():
    """A rank that dies (here: a workload name no rank accepts) must not look like a success."""
    rc, lines, err = _run_one_command(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "nope",
                                       "--no-cpu"])
    assert rc != 0
    assert not lines


def test_launcher_does_not_import_torch_or_touch_the_gpu():
    """The parent of `--gpus N` must start its children before anything initialises HIP: the
    launcher path imports neither torch nor the package's library."""
    code = (
        "import sys, os\n"
        "sys.argv = ['bench.py', '--gpus', '2']\n"
        "sys.path.insert(0, %r)\n"
        "import bench, subprocess\n"
        "class P(object):\n"
        "    stdout = []\n"
        "    def wait(self): return 7\n"
        "seen = {}\n"
        "def popen(cmd, **kw):\n"
        "    seen['cmd'] = cmd\n"
        "    seen['torch'] = 'torch' in sys.modules\n"
        "    seen['pkg'] = 'caffe_escoin_amd' in sys.modules\n"
        "    return P()\n"
        "subprocess.Popen = popen\n"
        "os.environ.pop('WORLD_SIZE', None)\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    rc = e.code\n"
        "assert rc == 7, rc\n"
        "assert seen['torch'] is False and seen['pkg'] is False, seen\n"
        "assert 'torch' not in sys.modules\n"
        "c = seen['cmd']\n"
        "assert c[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in c and '2' in c\n"
        "assert c[-2:] == ['--gpus', '2']\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=120)
    assert p.returncode == 0 and b"ok" in p.stdout, p.stderr.decode()[-2000:]


def test_every_tool_script_compiles():
    """tools/*.py run on the GPU box only (sweeps, probes' drivers, the parity fuzzer): at least their syntax is checked here."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) >= 20
    for f in files:
        with open(f) as src:
            compile(src.read(), f, "exec")


def test_traffic_bookkeeping_is_keyed_by_workload_batch_and_sparsity():
    """roofline.traffic comes from a committed PMC file; it only counts for the configuration it was collected on
    (a --global-batch 2048 line reported batch-256 traffic over batch-2048 bytes, 0.211)."""
    import importlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    path = os.path.join(ROOT, "profiles", "traffic_resnet50.json")
    blob = json.load(open(path))
    kernel = [k for k in blob if k.startswith("escoin_sconv")][0]
    batch = blob.get("_batch", 256)
    sp = blob.get("_sparsity_pct", 90)
    v, prov = bench.traffic_with_provenance("resnet50", kernel, batch, sp)
    assert v == blob[kernel] and prov["batch"] == batch and prov["sparsity_pct"] == sp
    # a batch the file does not have -> null, not a ratio of mismatched quantities
    assert bench.traffic_with_provenance("resnet50", kernel, 2048, sp) == (None, None)
    assert bench.traffic_with_provenance("resnet50", kernel, batch, 60) == (None, None)
    assert bench.traffic_per_layer("resnet50", 2048, sp) == {}
    assert bench.traffic_per_layer("resnet50", batch, sp)        # the matching configuration has its per-layer table
    assert bench.traffic_with_provenance("no_such_workload", kernel, 1, 1) == (None, None)
