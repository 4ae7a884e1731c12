#!/usr/bin/env python3
"""`caffe test -conv_mode {0,1,2,3}` for the convolution layers.

The reference's `caffe test` (tools/caffe.cpp:262-362) loads a model + a pruned .caffemodel,
runs `-iterations` forward passes and prints "[cxh] Total CONV time" per pass
(caffe.cpp:338-339, accumulated in Net::ForwardFromTo, net.cpp:592-604).  This harness does
that for the layers on the SCONV path: every sparse conv layer of the named model gets a plan
(WeightAlign), then each iteration launches all of them on one HIP stream and reports the
convolution time from HIP events.  Non-convolution layers are out of scope.

    python tools/caffe_test.py --model resnet50 --iterations 5
    python tools/caffe_test.py --model resnet50_chain --iterations 5      # + the dense 1x1 convs
    python tools/caffe_test.py --model alexnet --export /tmp/alexnet_pruned.caffemodel
    python tools/caffe_test.py --model alexnet --weights /tmp/alexnet_pruned.caffemodel --check

--model resnet50_chain runs ResNet-50's 16 bottleneck blocks as a chain: the dense 1x1 convolutions
(branch2a / branch2c / branch1; `EscConvolution` in the reference's prototxt, here the fp32 MFMA
kernel) around the sparse 3x3 ones, activations flowing from block to block, with the per-type time
buckets of Net::GetConvTime / GetOtherTime / GetTotalTime (net.cpp:516-565) printed the way
tools/caffe.cpp:338-343 does.  BatchNorm / Scale / ReLU / Eltwise are not part of the hot path: a
torch stand-in keeps the activations in range and is timed in the "other" bucket.

--weights takes the pruned weights (and biases) from a .caffemodel by layer name instead of the
synthetic generator; --export writes the synthetic pruned model in that format, so the same
file can be fed to the reference's own `caffe test`.  --check compares every layer's first
images with the CPU oracle (test infrastructure, not part of the product path).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def model_layers(synth, model, batch, sparsity):
    table = {"resnet50": (synth.resnet50_3x3, 256, 0.9), "alexnet": (synth.alexnet, 128, 0.8),
             "googlenet": (synth.googlenet_1x1, 256, 0.95), "lenet": (synth.lenet_conv2, 64, 0.5)}
    if model not in table:
        raise SystemExit("unknown --model %s (have: %s)" % (model, ", ".join(sorted(table))))
    fn, n, sp = table[model]
    shapes = fn(N=batch or n, sparsity=sp if sparsity is None else sparsity)
    out = []
    for s in shapes:
        for rep in range(s.count):
            name = s.name if s.count == 1 else "%s_%d" % (s.name, rep)
            out.append((name, s))
    return out


# ResNet-50 bottleneck stages after conv1 + pool1 (models/resnet/test_sconv.prototxt):
# (channels of the 3x3, output channels, blocks, stride of the first block)
_RESNET50_STAGES = [(64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2)]


def resnet50_chain(synth, batch, sparsity, sparsity_1x1=0.0):
    """[(name, kind, shape, relu, role)]: role in {"2a", "2b", "2c", "1"}; kind "sparse" for the
    pruned 3x3 (branch2b), "dense" for the 1x1 ones (sparsity 0; with --prune-1x1 they are pruned too and
    KERNEL_AUTO decides per layer what runs them)."""
    out = []
    s1 = float(sparsity_1x1)
    cin, hw = 64, 56
    for si, (mid, cout, blocks, stride) in enumerate(_RESNET50_STAGES):
        for b in range(blocks):
            st = stride if b == 0 else 1
            tag = "res%d%s" % (si + 2, "abcdef"[b])
            ohw = hw // st
            if b == 0:
                out.append((tag + "_branch1", "dense",
                            synth.shape(tag + "_branch1", batch, cin, hw, hw, cout, 1, stride=st, bias=False, sparsity=s1),
                            False, "1"))
            out.append((tag + "_branch2a", "dense",
                        synth.shape(tag + "_branch2a", batch, cin, hw, hw, mid, 1, stride=st, bias=False, sparsity=s1),
                        True, "2a"))
            out.append((tag + "_branch2b", "sparse",
                        synth.shape(tag + "_branch2b", batch, mid, ohw, ohw, mid, 3, pad=1, bias=False, sparsity=sparsity),
                        True, "2b"))
            out.append((tag + "_branch2c", "dense",
                        synth.shape(tag + "_branch2c", batch, mid, ohw, ohw, cout, 1, bias=False, sparsity=s1),
                        False, "2c"))
            cin, hw = cout, ohw
    return out


def _npz_blobs(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def run_chain(args, pkg, synth):
    import torch
    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("caffe_test.py needs a HIP device: the product path has no CPU fallback")
    print("[cxh] GPU device name: %s" % torch.cuda.get_device_name(0))
    base = _time_chain(args, pkg, synth, 0.0, "1x1 dense (the reference's EscConvolution layers)")
    if args.prune_1x1 is None:
        return 0
    # the what-if BASELINE configs[4] implies: the chain's 1x1 weights pruned too, KERNEL_AUTO per layer
    sp = args.prune_1x1 / 100.0
    pruned = _time_chain(args, pkg, synth, sp, "1x1 pruned @%g %%, KERNEL_AUTO per layer" % args.prune_1x1)
    print("[cxh] ResNet-50 chain, batch %d, conv time per forward pass (ms):" % (args.batch or 256))
    print("%-46s %12s %12s %12s" % ("", "sparse 3x3", "1x1 layers", "CONV total"))
    for label, r in (("1x1 dense (fp32 MFMA)", base), ("1x1 pruned @%g %% (KERNEL_AUTO)" % args.prune_1x1, pruned)):
        print("%-46s %12.3f %12.3f %12.3f" % (label, r["sparse"], r["dense"], r["sparse"] + r["dense"]))
    print("[cxh] Total CONV time: %.2f ms -> %.2f ms with the 1x1 layers on the sparse path (x%.2f)" %
          (base["sparse"] + base["dense"], pruned["sparse"] + pruned["dense"],
           (base["sparse"] + base["dense"]) / max(1e-9, pruned["sparse"] + pruned["dense"])))
    kinds = {}
    for name, kn in pruned["kernels"]:
        kinds[kn] = kinds.get(kn, 0) + 1
    print("kernels of the pruned chain: " + ", ".join("%d x %s" % (v, k) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])))
    return 0


def _time_chain(args, pkg, synth, sparsity_1x1, label):
    import torch
    dev = torch.device("cuda", 0)
    batch = args.batch or 256
    chain = resnet50_chain(synth, batch, 0.9 if args.sparsity is None else args.sparsity, sparsity_1x1)
    print("[cxh] --- %s ---" % label)
    plans, weights = [], []
    persisted = _npz_blobs(args.load_aligned) if (args.load_aligned and sparsity_1x1 == 0.0) else None
    need_w = persisted is None or args.check
    t0 = time.perf_counter()
    align_ms, t_gen, n_fast = [], 0.0, 0
    for i, (name, kind, s, relu, role) in enumerate(chain):
        tg = time.perf_counter()
        w = None
        if need_w:
            w = synth.pruned_weights(s, 1000 + 31 * i)
            # keep the activations O(1) through 16 blocks: He-style scale for the surviving weights
            w = (w * np.float32(np.sqrt(6.0 / max(1.0, (1.0 - s.sparsity) * w[0].size)))).astype(np.float32)
        t_gen += time.perf_counter() - tg
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), conv_mode=args.conv_mode)
        if args.dense_gate:
            plan.set_option("dense_gate", 1)
        ta = time.perf_counter()
        if persisted is not None:
            n_fast += int(plan.import_aligned(persisted[name]))      # the layer as a previous run left it
        else:
            plan.weight_align(w)
        align_ms.append(1e3 * (time.perf_counter() - ta))
        plans.append(plan)
        weights.append(w)
    torch.cuda.synchronize()
    total_ms = 1e3 * (time.perf_counter() - t0 - t_gen)
    how = ("from %s: %d code objects loaded as persisted" % (args.load_aligned, n_fast)) if persisted is not None else \
          "one-time, per weight load: net.cpp:819"
    print("[cxh] WeightAlign of %d layers: %.1f ms (%s; slowest: %s)" %
          (len(chain), total_ms, how,
           ", ".join("%s %.0f ms" % (chain[i][0], align_ms[i]) for i in sorted(range(len(chain)), key=lambda i: -align_ms[i])[:3])))
    if args.save_aligned and sparsity_1x1 == 0.0:
        ts = time.perf_counter()
        np.savez(args.save_aligned, **{chain[i][0]: plans[i].export_aligned() for i in range(len(chain))})
        print("[cxh] aligned form of %d layers written to %s (%.1f MB, %.0f ms)" %
              (len(chain), args.save_aligned, os.path.getsize(args.save_aligned if args.save_aligned.endswith(".npz") else args.save_aligned + ".npz") / 1e6,
               1e3 * (time.perf_counter() - ts)))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x0 = torch.rand((batch, 64, 56, 56), device=dev, generator=g) * 2 - 1
    oracle = ge.load_oracle() if args.check else None
    worst = 0.0

    def conv(i, x, bucket):
        nonlocal worst
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = plans[i].forward(x)
        b.record()
        bucket.append((chain[i][1], a, b, i))
        if oracle is not None:
            s = chain[i][2]
            geom = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                               s.dil_h, s.dil_w, s.group)
            want = oracle.conv_forward(geom, x[:1].cpu().numpy(), weights[i], None, relu=chain[i][3], gate=False)
            got = y[:1].cpu().numpy()
            err = float(np.abs(got.astype(np.float64) - want).max() / max(1e-6, np.abs(want).max()))
            if err > 1e-4:
                raise SystemExit("%s: relative error %.3g vs the oracle" % (chain[i][0], err))
            worst = max(worst, err)
        return y

    def other(fn, bucket):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = fn()
        b.record()
        bucket.append(("other", a, b, -1))
        return y

    print("Running for %d iterations." % args.iterations)
    sums = {"sparse": 0.0, "dense": 0.0, "other": 0.0}
    per_layer = np.zeros(len(chain))
    for it in range(args.iterations + 1):          # iteration 0 is an untimed warm-up
        bucket = []
        x = x0
        i = 0
        while i < len(chain):
            shortcut = x
            if chain[i][4] == "1":
                shortcut = conv(i, x, bucket)
                i += 1
            y = conv(i, x, bucket)        # branch2a (+ReLU)
            y = conv(i + 1, y, bucket)    # branch2b (+ReLU): the sparse 3x3
            y = conv(i + 2, y, bucket)    # branch2c
            # Eltwise + ReLU (+ a BatchNorm-like rescale): not the hot path, torch stand-in
            x = other(lambda: torch.relu_(y.add_(shortcut)).mul_(0.7071), bucket)
            i += 3
        torch.cuda.synchronize()
        if oracle is not None and it == 0:
            print("[cxh] oracle check: worst relative error %.3g over %d conv layers (image 0)" % (worst, len(chain)))
            oracle = None
        if it == 0:
            continue
        t = {"sparse": 0.0, "dense": 0.0, "other": 0.0}
        for kind, a, b, li in bucket:
            t[kind] += a.elapsed_time(b)
            if li >= 0:
                per_layer[li] += a.elapsed_time(b)
        for k in t:
            sums[k] += t[k]
        print("[cxh] Total CONV time: %.2f ms" % (t["sparse"] + t["dense"]))
        print("[cxh] Total forwarding time: %.2f ms" % sum(t.values()))
    n = max(1, args.iterations)
    conv_t = (sums["sparse"] + sums["dense"]) / n
    print("[cxh] Average over %d iterations (batch %d): CONV %.3f ms (sparse 3x3 %.3f ms in 16 layers, "
          "%s 1x1 %.3f ms in %d layers), other %.3f ms, total %.3f ms; conv / total = %.1f %%" %
          (n, batch, conv_t, sums["sparse"] / n, "pruned" if sparsity_1x1 > 0 else "dense", sums["dense"] / n,
           len(chain) - 16, sums["other"] / n,
           (conv_t + sums["other"] / n), 100.0 * conv_t / max(1e-9, conv_t + sums["other"] / n)))
    print("kernels: sparse -> %s, dense -> %s" % (plans[2].kernel_name, plans[0].kernel_name))
    if args.per_layer:
        print("%-22s %-44s %9s %9s %8s" % ("layer", "kernel", "us", "TFLOP/s", "GB/s"))
        for li, (name, kind, s, relu, role) in enumerate(chain):
            us = 1e3 * per_layer[li] / n
            dense_fl = 2.0 * batch * s.M * (s.C // s.group) * s.KH * s.KW * synth.out_hw(s)[0] * synth.out_hw(s)[1]
            fl = dense_fl if "dense_mfma" in plans[li].kernel_name else synth.flops(s)
            print("%-22s %-44s %9.1f %9.2f %8.0f" % (name, plans[li].kernel_name[:44], us, fl / us * 1e-6,
                                                    synth.algorithmic_bytes(s) / us * 1e-3))
    out = {"sparse": sums["sparse"] / n, "dense": sums["dense"] / n, "other": sums["other"] / n,
           "kernels": [(chain[i][0], plans[i].kernel_name) for i in range(len(chain))], "align_ms": total_ms}
    for p in plans:
        p.close()
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--weights", default=None, help=".caffemodel with pruned weights (by layer name)")
    ap.add_argument("--export", default=None, help="write the synthetic pruned model here and exit")
    ap.add_argument("--conv_mode", type=int, default=3, choices=[0, 1, 2, 3],
                    help="0 = LOWERED_GEMM (dense MFMA kernel), 1 = LOWERED_SPARSE comparator "
                         "(im2col + csrmm), 2/3 = direct sparse convolution")
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--sparsity", type=float, default=None)
    ap.add_argument("--dense-gate", action="store_true",
                    help="honour the reference's density > 0.2 -> dense GEMM gate")
    ap.add_argument("--check", action="store_true", help="compare with the CPU oracle (first 2 images)")
    ap.add_argument("--prune-1x1", type=float, default=None, metavar="PCT",
                    help="resnet50_chain: a second pass with the 1x1 weights pruned at PCT %% and KERNEL_AUTO per "
                         "layer; both `[cxh] Total CONV time` buckets side by side")
    ap.add_argument("--save-aligned", default=None, metavar="FILE.npz",
                    help="resnet50_chain: persist every layer's aligned form (escoin_plan_export_aligned: CSR, "
                         "channel deal, unit table, code object)")
    ap.add_argument("--load-aligned", default=None, metavar="FILE.npz",
                    help="resnet50_chain: restore the layers from --save-aligned's file instead of WeightAlign")
    ap.add_argument("--per-layer", action="store_true", help="resnet50_chain: per-layer table")
    ap.add_argument("--tilings", action="store_true", help="print how every layer's plan tiles it (escoin_plan_tiling_info)")
    args = ap.parse_args()

    pkg = ge.load_package()
    from caffe_escoin_amd import caffemodel as cm
    synth = pkg.synth
    if args.model == "resnet50_chain":
        return run_chain(args, pkg, synth)
    layers = model_layers(synth, args.model, args.batch, args.sparsity)

    weights = {}
    for i, (name, s) in enumerate(layers):
        weights[name] = (synth.pruned_weights(s, 1000 + 31 * i), synth.bias_vector(s, 2000 + 31 * i))
    if args.export:
        cl = [cm.CaffeLayer(name, "Convolution", [w] + ([b] if b is not None else []), cm.conv_param_of(s))
              for (name, s) in layers for (w, b) in [weights[name]]]
        cm.write_caffemodel(args.export, args.model + "_pruned", cl)
        print("wrote %s: %d convolution layers" % (args.export, len(cl)))
        return 0
    if args.weights:
        _, file_layers = cm.read_caffemodel(args.weights)
        got = cm.conv_weights(file_layers)
        used = 0
        for name, s in layers:
            if name not in got:
                continue
            w, b = got[name]
            want = (s.M, s.C // s.group, s.KH, s.KW)
            if tuple(w.shape) != want:
                raise SystemExit("%s: weight blob %s, layer expects %s" % (name, w.shape, want))
            weights[name] = (w, b if s.bias else None)
            used += 1
        print("[cxh] %s: weights of %d / %d layers taken from the file" % (args.weights, used, len(layers)))

    import torch
    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("caffe_test.py needs a HIP device: the product path has no CPU fallback")
    dev = torch.device("cuda", 0)
    print("[cxh] GPU device name: %s" % torch.cuda.get_device_name(0))

    plans, bottoms, tops, biases = [], {}, {}, []
    t0 = time.perf_counter()
    align_ms = []
    for name, s in layers:
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), conv_mode=args.conv_mode)
        if args.dense_gate:
            plan.set_option("dense_gate", 1)
        w, b = weights[name]
        ta = time.perf_counter()
        plan.weight_align(w)
        align_ms.append(1e3 * (time.perf_counter() - ta))
        plans.append(plan)
        biases.append(torch.from_numpy(b).to(dev) if b is not None else None)
        key = (s.C, s.H, s.W, s.M, s.KH, s.stride_h, s.pad_h)
        if key not in bottoms:
            g = torch.Generator(device=dev)
            g.manual_seed(len(bottoms) + 1)
            bottoms[key] = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=g) * 2 - 1
            oh, ow = synth.out_hw(s)
            tops[key] = torch.empty((s.N, s.M, oh, ow), device=dev)
    torch.cuda.synchronize()
    print("[cxh] WeightAlign of %d layers: %.1f ms" % (len(layers), 1e3 * (time.perf_counter() - t0)))

    def key_of(s):
        return (s.C, s.H, s.W, s.M, s.KH, s.stride_h, s.pad_h)

    per_layer = np.zeros(len(layers))
    conv_total = 0.0
    print("Running for %d iterations." % args.iterations)
    for it in range(args.iterations + 1):          # iteration 0 is an untimed warm-up
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in layers]
        for li, (name, s) in enumerate(layers):
            ev[li][0].record()
            plans[li].forward(bottoms[key_of(s)], biases[li], tops[key_of(s)])
            ev[li][1].record()
        torch.cuda.synchronize()
        if it == 0:
            continue
        ms = np.array([a.elapsed_time(b) for a, b in ev])
        per_layer += ms
        conv_total += ms.sum()
        print("[cxh] Total CONV time: %.2f ms" % ms.sum())
    print("[cxh] Average CONV time: %.3f ms over %d iterations (batch %d)" %
          (conv_total / args.iterations, args.iterations, layers[0][1].N))
    print("%-26s %-34s %9s %9s %8s %14s" % ("layer", "kernel", "us", "TFLOP/s", "GB/s", "WeightAlign ms"))
    for li, (name, s) in enumerate(layers):
        us = 1e3 * per_layer[li] / args.iterations
        print("%-26s %-34s %9.1f %9.2f %8.0f %14.1f" % (name, plans[li].kernel_name[:34], us,
                                                        synth.flops(s) / us * 1e-6,
                                                        synth.algorithmic_bytes(s) / us * 1e-3, align_ms[li]))
    if args.tilings:
        for li, (name, s) in enumerate(layers):
            print("%-26s %s" % (name, plans[li].tiling_info or "(no tiled plan: %s)" % plans[li].kernel_name))

    if args.check:
        oracle = ge.load_oracle()
        worst = 0.0
        for li, (name, s) in enumerate(layers):
            n = min(2, s.N)
            x = bottoms[key_of(s)][:n].contiguous()
            top = plans[li].forward(x, biases[li]).cpu().numpy()
            w, b = weights[name]
            geom = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                               s.dil_h, s.dil_w, s.group)
            want = oracle.conv_forward(geom, x.cpu().numpy(), w, b, gate=args.dense_gate)
            err = float(np.abs(top.astype(np.float64) - want).max() / max(1e-6, np.abs(want).max()))
            worst = max(worst, err)
            if err > 1e-4:
                raise SystemExit("%s: relative error %.3g vs the oracle" % (name, err))
        print("[cxh] oracle check: worst relative error %.3g over %d layers" % (worst, len(layers)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
