#!/usr/bin/env python3
"""GPU box: seeded random convolution geometries through every kernel family against the oracle -- wider than the
suite's randomised tests (strides, dilations, non-square kernels and pads, conv groups, up to 600 channels, skewed
sparsity, fused ReLU, tilings of batches the inputs do not have).  The oracle (oracle/, the CPU restatement of the
reference's loop nest) is the checker, as in tests/.
    python tools/fuzz_parity.py [cases] [seed] > gpurun_out/fuzz.txt
Prints one line per failure (everything needed to rebuild the case) and a summary; exit code 1 on any failure."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

TOL = 5e-5


def rel_err(got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64)).max() if got.size else 0.0
    return d / max(1.0, float(np.abs(want).max()) if want.size else 1.0)


def dropin_chain(pkg, oracle, synth, dev, s, w, x, k):
    """The math_functions-level drop-ins on the reference's own layouts, as base_conv_layer.cpp calls them: device
    dense -> CSR, stretch, padded copy, caffe_gpu_sconv's replacement.  Returns None or what differed."""
    import ctypes as C
    L = pkg.lib()
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w)
    kdim = s.C * s.KH * s.KW
    P = lambda t: C.c_void_p(t.data_ptr())
    A = torch.from_numpy(np.ascontiguousarray(w.reshape(s.M, kdim))).to(dev)
    vals = torch.zeros(s.M * kdim, device=dev)
    cols = torch.zeros(s.M * kdim, dtype=torch.int32, device=dev)
    rowp = torch.zeros(s.M + 1, dtype=torch.int32, device=dev)
    perrow = torch.zeros(s.M, dtype=torch.int32, device=dev)
    nnz = C.c_int()
    if L.escoin_gpu_sparse_dense2csr(s.M, kdim, P(A), P(perrow), P(vals), P(rowp), P(cols), C.byref(nnz), None) != 0:
        return "dense2csr failed"
    orp, oci, ova = oracle.dense2csr(w.reshape(s.M, kdim))
    if nnz.value != len(oci) or not np.array_equal(rowp.cpu().numpy(), orp) or \
            not np.array_equal(cols.cpu().numpy()[:nnz.value], oci) or not np.array_equal(vals.cpu().numpy()[:nnz.value], ova):
        return "dense2csr differs from the oracle's"
    if L.escoin_gpu_stretch(P(rowp), P(cols), s.M, s.H, s.W, s.pad_h, s.pad_w, s.KH, s.KW, None) != 0:
        return "stretch failed"
    if not np.array_equal(cols.cpu().numpy()[:nnz.value], oracle.stretch(orp, oci, s.KH, s.KW, s.H, s.W, s.pad_h, s.pad_w)):
        return "stretch differs from the oracle's"
    N = x.shape[0]
    ifmap = s.C * (s.H + s.pad_h) * (s.W + s.pad_w)
    padded = torch.zeros(N * ifmap + oracle.padded_len(g), device=dev)
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for n in range(N):
        if L.escoin_copy_input_data(C.c_void_p(padded.data_ptr() + 4 * n * ifmap), C.c_void_p(xd.data_ptr() + 4 * n * s.C * s.H * s.W),
                                    s.C, s.H, s.W, s.pad_h, s.pad_w, None) != 0:
            return "copy_input_data failed"
    oh, ow = oracle.out_hw(g)
    out = torch.zeros(N, s.M, oh, ow, device=dev)
    bias = synth.uniform(7000 + k, s.M, -0.1, 0.1)
    bd = torch.from_numpy(bias).to(dev)
    base = oracle.conv_forward(g, x, w, None, gate=False)
    for relu in (0, 1):
        if L.escoin_gpu_sconv(relu, N, P(padded), ifmap, P(rowp), P(cols), P(vals), P(bd), s.H, s.W, s.pad_h, s.pad_w, s.stride_h,
                              s.stride_w, s.dil_h, s.dil_w, s.KH, s.KW, P(out), s.M, 1, None) != 0:
            return "gpu_sconv failed"
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        if relu == 0 and not np.array_equal(got, base):
            return "gpu_sconv not bit-exact (rel err %.3g)" % rel_err(got, base)
        if relu == 1 and rel_err(got, np.maximum(base + bias[None, :, None, None], 0)) > 1e-6:
            return "gpu_sconv with bias + ReLU off by %.3g" % rel_err(got, np.maximum(base + bias[None, :, None, None], 0))
    return None


def fuzz(cases, seed, out=sys.stdout):
    """Runs `cases` random geometries; returns (kernel runs, failure lines, runs per kernel name)."""
    pkg, oracle = ge.load_package(), ge.load_oracle()
    synth = pkg.synth
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda:0")
    kernels = {"generic": pkg.KERNEL_GENERIC, "auto": pkg.KERNEL_AUTO, "tiled": pkg.KERNEL_TILED, "jit": pkg.KERNEL_JIT,
               "dense": pkg.KERNEL_DENSE}
    ran, failed, by_name, lines, imports, chains, partial, modes = 0, 0, {}, [], 0, 0, 0, 0

    def report(line):
        lines.append(line)
        print(line, file=out, flush=True)
    t0 = time.time()
    for k in range(cases):
        cls = k % 5          # 0: anything goes, 1: stride 1 (the LDS-tiled families), 2: pointwise, 3: many channels, 4: many images
        KH = int(rng.choice([1, 1, 2, 3, 3, 3, 4, 5, 7]))
        KW = KH if rng.randint(4) else int(rng.choice([1, 2, 3, 5]))
        if cls == 2:
            KH = KW = 1
        sh, sw = (1, 1) if cls in (1, 2, 3, 4) and rng.randint(4) else (int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 3])))
        if cls == 2 and rng.randint(3) == 0:
            sh = sw = 2
        dh, dw = (1, 1) if cls != 0 or rng.randint(3) else (int(rng.choice([1, 2])), int(rng.choice([1, 2, 3])))
        ph = int(rng.randint(0, KH)) if KH > 1 else 0
        pw = int(rng.randint(0, KW)) if KW > 1 else 0
        if rng.randint(8) == 0:
            ph, pw = ph + 1, pw + 2         # more padding than the kernel reaches
        eh, ew = dh * (KH - 1) + 1, dw * (KW - 1) + 1
        H = int(rng.randint(max(eh - 2 * ph, 1), 34))
        W = int(rng.randint(max(ew - 2 * pw, 1), 66 if cls != 3 else 30))
        group = int(rng.choice([1, 1, 1, 2, 3, 4]))
        if cls == 3:
            C, M = group * int(rng.randint(40, 200)), group * int(rng.randint(30, 200))
            N = int(rng.randint(1, 5))
        elif cls == 4:       # several tiles per workgroup, tile slots past the batch, the persistent loop's rotation
            C, M = group * int(rng.randint(1, 12)), group * int(rng.randint(1, 40))
            N = int(rng.randint(100, 700))
            H, W = min(H, int(rng.randint(max(eh - 2 * ph, 1), 16))), min(W, int(rng.randint(max(ew - 2 * pw, 1), 20)))
        else:
            C, M = group * int(rng.randint(1, 40)), group * int(rng.randint(1, 70))
            N = int(rng.randint(1, 20))
        sp = float(rng.choice([0.0, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95, 0.99, 1.0]))
        dist = str(rng.choice(["uniform", "uniform", "channel", "zero_inputs", "filters_tail"]))
        bias, relu = bool(rng.randint(2)), bool(rng.randint(3) == 0)
        s = synth.shape("fz%d" % k, N, C, H, W, M, KH, KW=KW, pad=ph, pad_w=pw, stride=sh, stride_w=sw, dil=dh, dil_w=dw,
                        group=group, sparsity=sp, bias=bias)
        try:
            w = synth.pruned_weights(s, 1000 + k, dist=dist)
        except Exception:
            w = synth.pruned_weights(s, 1000 + k)
        b, x = synth.bias_vector(s, 2000 + k), synth.activations(s, 3000 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        if relu:
            want = np.maximum(want, 0.0)
        if group == 1 and k % 3 == 0 and N * C * H * W < 400000:
            why = dropin_chain(pkg, oracle, synth, dev, s, w, x, k)
            chains += 1
            if why:
                report("FAIL drop-in chain %s seed=%d k=%d: %s" % (tuple(s), seed, k, why))
                failed += 1
        tb = int(rng.choice([0, 0, 64, 256, 257, 1000])) if cls != 4 else 0
        mlb = int(rng.choice([0, 0, 1])) * (4 * C * H * W * int(rng.randint(3, 60)) + 100)     # sub-batch launches
        desc = pkg.ConvDesc.from_shape(s, fuse_relu=relu)
        xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        bd = torch.from_numpy(b).to(dev) if b is not None else None
        for kn, kernel in kernels.items():
            opts = {"tiling_batch": tb}
            if mlb and cls == 4:
                opts["max_launch_bytes"] = mlb
            if kn == "auto" and rng.randint(2):
                opts["dense_threshold_pct"] = 100         # keep AUTO on the sparse kernels
            # Caffe::conv_mode (caffe.cpp -conv_mode): the four modes on the plan's own pick, one case in three
            cmode = pkg.CONV_MODE_SCONV_PAR
            if kn == "auto" and rng.randint(3) == 0:
                cmode = int(rng.choice([pkg.CONV_MODE_LOWERED_GEMM, pkg.CONV_MODE_LOWERED_SPARSE, pkg.CONV_MODE_SCONV]))
                modes += 1
            try:
                plan = pkg.Plan(desc, kernel=kernel, conv_mode=cmode, **opts)
            except Exception as e:      # a forced kernel that does not cover the geometry says so at plan creation
                if kn in ("tiled", "jit"):
                    continue
                report("FAIL create %s %s: %s" % (kn, tuple(s), e))
                failed += 1
                continue
            try:
                plan.weight_align(w)
                got = plan.forward(xd, bd).cpu().numpy()
                name = plan.kernel_name
                # ... and a call on FEWER images than the plan was created for (the reference's SCONV mode hands the layer one
                # image at a time, conv_layer.cu:16-26): the same values for those images, bit for bit
                if N > 1 and rng.randint(3) == 0:
                    n_part = int(rng.randint(1, N))
                    part = plan.forward(xd[:n_part].contiguous(), bd).cpu().numpy()
                    partial += 1
                    if part.shape[0] != n_part or not np.array_equal(part, got[:n_part]):
                        report("FAIL partial batch %s via %s %s: %d of %d images differ from the full call (rel err %.3g) seed=%d k=%d" %
                               (kn, name, tuple(s), n_part, N, rel_err(part, got[:n_part]), seed, k))
                        failed += 1
            except Exception as e:
                if kn in ("tiled", "jit") and "requested" in str(e):      # a forced kernel that does not cover the geometry says so
                    plan.close()
                    continue
                report("FAIL run %s %s tb=%d dist=%s relu=%d: %s" % (kn, tuple(s), tb, dist, relu, e))
                failed += 1
                plan.close()
                continue
            # one plan in four hands its aligned form (CSR + channel deal + code object) to a fresh plan, as a rank does to
            # the others (shard.py), and one in four of those receivers has another tiling batch: the receiver's result
            # must be the sender's, bit for bit, when it runs the same kernel
            if kn in ("auto", "jit") and rng.randint(4) == 0:
                try:
                    blob = plan.export_aligned()
                    ropts = dict(opts)
                    if rng.randint(4) == 0:
                        ropts["tiling_batch"] = 64 if tb != 64 else 256
                    recv = pkg.Plan(desc, kernel=kernel, **ropts)
                    recv.import_aligned(blob)
                    got2 = recv.forward(xd, bd).cpu().numpy()
                    name2 = recv.kernel_name
                    recv.close()
                    imports += 1
                    if rel_err(got2, want) > TOL or (name2 == name and ropts == opts and not np.array_equal(got2, got)):
                        report("FAIL import %s via %s -> %s %s tb=%d dist=%s seed=%d k=%d: rel err %.3g, same bits %s" %
                               (kn, name, name2, tuple(s), tb, dist, seed, k, rel_err(got2, want), np.array_equal(got2, got)))
                        failed += 1
                except Exception as e:
                    report("FAIL import %s %s tb=%d seed=%d k=%d: %s" % (kn, tuple(s), tb, seed, k, e))
                    failed += 1
            plan.close()
            ran += 1
            by_name[name] = by_name.get(name, 0) + 1
            err = rel_err(got, want)
            exact = "generic" in name and not np.array_equal(got, want)
            if err > TOL or got.shape != want.shape or exact:
                report("FAIL parity %s via %s %s tb=%d dist=%s relu=%d seed=%d k=%d: rel err %.3g%s" %
                       (kn, name, tuple(s), tb, dist, relu, seed, k, err, " (generic kernel not bit-exact)" if exact else ""))
                failed += 1
        if k % 50 == 49:
            print("# %d cases, %d runs, %d failures, %.0f s" % (k + 1, ran, failed, time.time() - t0), file=out, flush=True)
    by_name["(aligned forms handed to a fresh plan)"] = imports
    by_name["(math_functions-level drop-in chains)"] = chains
    by_name["(calls on fewer images than the plan's batch)"] = partial
    by_name["(plans in another Caffe::conv_mode)"] = modes
    return ran, lines, by_name


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
    ran, lines, by_name = fuzz(cases, seed)
    print("cases %d seed %d runs %d failures %d" % (cases, seed, ran, len(lines)))
    for n in sorted(by_name):
        print("  %-60s %d" % (n, by_name[n]))
    sys.exit(1 if lines else 0)


if __name__ == "__main__":
    main()
