#!/usr/bin/env python3
"""bench.py -- conv-layer forward images/s of the direct-sparse-convolution path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet50|alexnet|googlenet|lenet]

A "step" is one pass of the hot path (escoin_forward, the Forward_gpu drop-in) over one batch
of synthetic input for every conv layer of the workload.  Default workload = BASELINE.json's
configs[2]: the 16 ResNet-50 3x3 branch2b layers at 90 % weight sparsity, batch 256 per GPU,
fp32.  For N > 1 the driver launches one process per GPU (torch.distributed.run); the batch
dimension is sharded (weak scaling: 256 images per GPU), the sparse weights are broadcast
once from rank 0 over RCCL, and no collective sits in the timed region.

Rank 0 prints ONE JSON line on stdout (everything else goes to stderr).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def workload_layers(synth, name, batch, sparsity):
    if name == "resnet50":
        return synth.resnet50_3x3(N=batch or 256, sparsity=0.9 if sparsity is None else sparsity), \
            "ResNet-50 16x 3x3 branch2b conv layers"
    if name == "alexnet":
        return synth.alexnet(N=batch or 128, sparsity=0.8 if sparsity is None else sparsity), \
            "AlexNet conv2-conv5"
    if name == "googlenet":
        return synth.googlenet_1x1(N=batch or 256, sparsity=0.95 if sparsity is None else sparsity), \
            "GoogLeNet-v1 1x1 convs"
    if name == "lenet":
        return synth.lenet_conv2(N=batch or 64, sparsity=0.5 if sparsity is None else sparsity), \
            "LeNet-5 conv2"
    raise SystemExit("unknown workload %s" % name)


def cpu_baseline(oracle, synth, shapes, budget_s):
    """Reference CPU sconv path timed on this box's host cores on a bounded sample.

    kind "reference": oracle/_ref = the reference's own kernel (sconv.hpp:594-678) compiled in
    place, batch loop parallelised over images with OpenMP the way its ICC build does
    (conv_layer.cpp:41-43); kind "port": the C restatement when _ref is absent."""
    cores = os.cpu_count() or 1
    use_ref = oracle.have_ref()
    per_image_s = 0.0
    per_image_1t_s = 0.0      # as the reference's g++ build runs it: one thread, serial batch loop
    sample = []
    share = budget_s / max(1, len(shapes))
    for k, s in enumerate(shapes):
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                        s.dil_h, s.dil_w, s.group)
        w = synth.pruned_weights(s, 1000 + k)
        b = synth.bias_vector(s, 2000 + k)
        run = (lambda x: oracle.ref_conv_forward(g, x, w, b, threads=cores)) if use_ref else \
              (lambda x: oracle.conv_forward(g, x, w, b, threads=cores, gate=False))
        n = cores
        x = synth.activations(s, 3000 + k, 0, n)
        run(x)                                        # warm (threads, page faults)
        t0 = time.perf_counter()
        run(x)
        t1 = time.perf_counter() - t0
        reps = int(max(1, min(64, share / max(t1, 1e-4))))
        if reps > 1:
            n = cores * min(reps, 8)
            x = synth.activations(s, 3000 + k, 0, n)
            t0 = time.perf_counter()
            run(x)
            t1 = time.perf_counter() - t0
        per_image_s += s.count * t1 / n
        run1 = (lambda x: oracle.ref_conv_forward(g, x, w, b, threads=1)) if use_ref else \
               (lambda x: oracle.conv_forward(g, x, w, b, threads=1, gate=False))
        x1 = synth.activations(s, 3000 + k, 0, 2)
        run1(x1[:1])
        t0 = time.perf_counter()
        run1(x1)
        per_image_1t_s += s.count * (time.perf_counter() - t0) / 2
        sample.append("%s:%dimg" % (s.name, n))
        log("  cpu %-16s %6d img in %.3f s -> %.1f img/s/layer (%d threads)" %
            (s.name, n, t1, n / t1, cores))
    return {"value": round(1.0 / per_image_s, 3), "unit": "images/s", "cores": cores,
            "kind": "reference" if use_ref else "port",
            "single_thread_value": round(1.0 / per_image_1t_s, 3),
            "sample": "whole-batch forward of " + ", ".join(sample) +
                      " per distinct layer shape, OpenMP over images; per-image time summed "
                      "over all %d layers; single_thread_value = the same on one thread (2 images "
                      "per shape), the way the reference's g++ build runs its batch loop"
                      % sum(s.count for s in shapes)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="resnet50")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU")
    ap.add_argument("--sparsity", type=float, default=None)
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "tiled"])
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds for the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    pkg = ge.load_package()
    synth = pkg.synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    shapes, wl_name = workload_layers(synth, args.workload, args.batch, args.sparsity)
    per_gpu_batch = shapes[0].N
    kernel = {"auto": pkg.KERNEL_AUTO, "generic": pkg.KERNEL_GENERIC, "tiled": pkg.KERNEL_TILED}[args.kernel]

    # ---- WeightAlign on rank 0, RCCL broadcast of the CSR, set_csr everywhere else ----------
    layers = []          # (shape, plan, bias, shape_index)
    t_bcast = 0.0
    lid = 0
    for si, s in enumerate(shapes):
        for rep in range(s.count):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel)
            mg = s.M // s.group
            if rank == 0:
                plan.weight_align(synth.pruned_weights(s, 1000 + 31 * lid))
            if world > 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                csr = plan.get_csr() if rank == 0 else None
                got = pkg.shard.broadcast_csr(csr, s.group, s.group * (mg + 1), synth.nnz_of(s),
                                              src=0, device=dev)
                torch.cuda.synchronize()
                t_bcast += time.perf_counter() - t0
                if rank != 0:
                    plan.set_csr(*got)
            bias = synth.bias_vector(s, 2000 + 31 * lid)
            bias = torch.from_numpy(bias).to(dev) if bias is not None else None
            layers.append((s, plan, bias, si))
            lid += 1

    # ---- synthetic activations resident in HBM (image k seeded by its GLOBAL index) -----------
    gen = torch.Generator(device=dev)
    bottoms, tops = [], []
    for si, s in enumerate(shapes):
        gen.manual_seed(977 * (si + 1) + rank * 7919)
        bottoms.append(torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1)
        oh, ow = synth.out_hw(s)
        tops.append(torch.empty((s.N, s.M, oh, ow), device=dev))
    torch.cuda.synchronize()

    def step(events=None):
        # one HIP event between consecutive launches (the end of launch i is the start of launch
        # i + 1): half the marker packets of a start/stop pair per launch, and a launch's duration
        # then includes its dispatch gap, which is what the step pays for it
        if events is not None:
            events[0].record()
        for li, (s, plan, bias, si) in enumerate(layers):
            plan.forward(bottoms[si], bias, tops[si])
            if events is not None:
                events[li + 1].record()

    for _ in range(args.warmup):
        step()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(layers) + 1)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- per-kernel accounting from the HIP events recorded inside the timed region ----------
    per_kernel = {}
    layer_ms = []
    for li, (s, plan, bias, si) in enumerate(layers):
        ms = [ev[k][li].elapsed_time(ev[k][li + 1]) for k in range(args.steps)]
        m = float(np.mean(ms))
        layer_ms.append(m)
        d = per_kernel.setdefault(plan.kernel_name, {"ms": 0.0, "bytes": 0, "flops": 0, "launches": 0})
        d["ms"] += m
        d["bytes"] += synth.algorithmic_bytes(s)
        d["flops"] += synth.flops(s)
        d["launches"] += 1
    seen = set()
    for li, (s, plan, bias, si) in enumerate(layers):
        if si in seen:
            continue
        seen.add(si)
        m = layer_ms[li]
        log("  gpu %-16s %-44s %8.1f us  %7.1f GB/s alg  %6.2f TFLOP/s  x%d" %
            (s.name, plan.kernel_name, m * 1e3, synth.algorithmic_bytes(s) / m / 1e6,
             synth.flops(s) / m / 1e9, s.count))
    dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["ms"])
    achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom_name)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                "launches_per_step": dom["launches"],
                "algorithmic_bytes_per_launch": int(dom["bytes"] / dom["launches"]),
                "sparse_tflops": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 2)}

    ms_per_step = elapsed / args.steps * 1e3
    value = per_gpu_batch * world / (ms_per_step * 1e-3)
    out = {
        "metric": "conv-layer fwd images/sec", "value": round(value, 1), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s @%d%% sparsity, batch %d/GPU, fp32" %
                               (wl_name, round(100 * shapes[0].sparsity), per_gpu_batch),
                   "global_batch": per_gpu_batch * world, "layers_per_step": len(layers),
                   "kernel": args.kernel, "parallelism": "batch-sharded x%d" % world,
                   "weight_broadcast_ms": round(t_bcast * 1e3, 3) if world > 1 else None},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu:
        oracle = ge.load_oracle()
        log("cpu_baseline (bounded sample, %.0f s budget):" % args.cpu_budget)
        out["cpu_baseline"] = cpu_baseline(oracle, synth, shapes, args.cpu_budget)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
