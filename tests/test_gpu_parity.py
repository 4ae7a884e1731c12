"""Parity of the HIP path (through the C ABI) against the CPU oracle -- needs an MI355X.

Tolerance (BASELINE.json north_star): 1e-4 relative fp32, defined as
    max|got - want| / max(1e-6, max|want|) <= 1e-4        
The generic kernel keeps the reference's summation order and must be BIT-EXACT.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import Golden, golden_params, naive_conv, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert pkg.device_count() >= 1
    return torch


def _run(pkg, torch, desc, w, x, bias, kernel, **options):
    plan = pkg.Plan(desc, kernel=kernel, **options)
    plan.weight_align(w)
    dev = torch.device("cuda:0")
    top = plan.forward(torch.from_numpy(np.ascontiguousarray(x)).to(dev),
                       torch.from_numpy(bias).to(dev) if bias is not None else None)
    torch.cuda.synchronize()
    name = plan.kernel_name
    out = top.cpu().numpy()
    plan.close()
    return out, name


def _kernels(pkg, desc=None):
    """Every kernel family: generic (reference order), AUTO (generated code where the geometry allows,
    jit_codegen.h), the LDS-staged stream kernel where the geometry allows, dense MFMA."""
    ks = [pkg.KERNEL_GENERIC, pkg.KERNEL_AUTO, pkg.KERNEL_DENSE]
    if desc is not None and _tiled_ok(desc):
        ks[2:2] = [pkg.KERNEL_TILED, pkg.KERNEL_JIT]
    return ks


def _tiled_ok(d):
    return (d.stride_h == 1 and d.stride_w == 1 and d.dil_h == 1 and d.dil_w == 1 and d.KW <= 5 and
            max(d.W, d.W + 2 * d.pad_w - d.KW + 1) <= 256 and d.pad_w <= 4 and d.KW - 1 - d.pad_w <= 4)


def _fast(name):
    """One of the two LDS-tiled kernel families (weight walk generated / LDS-staged stream)."""
    return "tiled" in name or "jit" in name


@pytest.mark.parametrize("path", golden_params())
def test_golden_fixtures(pkg, torch_cuda, path):
    gd = Golden(path)
    for kernel in _kernels(pkg, gd.desc(pkg)):
        got, name = _run(pkg, torch_cuda, gd.desc(pkg), gd.w, gd.x, gd.bias, kernel)
        assert got.shape == gd.top.shape
        if "generic" in name:
            assert np.array_equal(got, gd.top), "%s: generic kernel must be bit-exact" % gd.name
        assert rel_err(got, gd.top) <= TOL, "%s via %s" % (gd.name, name)
    # ... and through the tiling (channels per wave, images per tile, blocks) a batch of 256 / 64
    # gets: the weight stream and tile shapes of the benchmarked configurations
    for tb in (256, 64):
        for kernel in [pkg.KERNEL_AUTO] + ([pkg.KERNEL_TILED, pkg.KERNEL_JIT] if _tiled_ok(gd.desc(pkg)) else []):
            got, name = _run(pkg, torch_cuda, gd.desc(pkg), gd.w, gd.x, gd.bias, kernel, tiling_batch=tb)
            assert rel_err(got, gd.top) <= TOL, "%s via %s, tiling_batch %d" % (gd.name, name, tb)


def _config_shapes(synth):
    return (synth.lenet_conv2(N=3) + synth.alexnet(N=3) + synth.resnet50_3x3(N=3) +
            [synth.googlenet_1x1(N=3)[i] for i in (0, 1, 30, 36, 38)])


def test_config_layers_small_batch_vs_oracle(pkg, oracle, synth, torch_cuda):
    """Every layer shape of BASELINE.json's configs at its real channel counts, batch 3."""
    for k, s in enumerate(_config_shapes(synth)):
        w = synth.pruned_weights(s, 100 + k)
        b = synth.bias_vector(s, 200 + k)
        x = synth.activations(s, 300 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                        s.dil_h, s.dil_w, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False, threads=4)
        for kernel in _kernels(pkg, pkg.ConvDesc.from_shape(s)):
            got, name = _run(pkg, torch_cuda, pkg.ConvDesc.from_shape(s), w, x, b, kernel)
            if "generic" in name:
                assert np.array_equal(got, want), s.name
            assert rel_err(got, want) <= TOL, "%s via %s: %g" % (s.name, name, rel_err(got, want))


def _config_sets(synth):
    """BASELINE.json's configurations at their own batch sizes: the tilings bench.py times."""
    return [("lenet", synth.lenet_conv2(N=64)), ("alexnet", synth.alexnet(N=128)),
            ("resnet50", synth.resnet50_3x3(N=256)), ("googlenet", synth.googlenet_1x1(N=256))]


def _generic_reference(pkg, torch, s, seed, relu=False, oracle=None):
    """The whole config batch through the generic kernel -- one lane per output pixel, the reference's
    loop nest and summation order -- on the same device-generated input _check_full_batch uses: the on-device
    yardstick for all N images of a fast kernel's output.  With `oracle`: EVERY image of that yardstick is
    checked against the oracle first, bit for bit (the oracle on all host threads: 0.1-0.3 s per layer), so the
    fast kernels' all-N comparison is anchored to the oracle image by image, not to a kernel of this library."""
    synth_mod = pkg.synth
    dev = torch.device("cuda:0")
    w, b = synth_mod.pruned_weights(s, seed), synth_mod.bias_vector(s, seed + 1)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=pkg.KERNEL_GENERIC)
    plan.weight_align(w)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 2)
    x = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1
    top = plan.forward(x, torch.from_numpy(b).to(dev) if b is not None else None)
    torch.cuda.synchronize()
    assert "generic" in plan.kernel_name
    plan.close()
    if oracle is not None:
        import os
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
        want = oracle.conv_forward(g, x.cpu().numpy(), w, b, relu=relu, gate=False, threads=max(1, os.cpu_count() or 1))
        assert np.array_equal(top.cpu().numpy(), want), "%s: the generic kernel differs from the oracle on the full batch of %d" % (s.name, s.N)
    return top


def _check_full_batch(pkg, oracle, synth, torch, s, seed, plan=None, images=None, kernel=None, relu=False,
                      full_ref=None):
    """Forward of the WHOLE config batch on device-generated input; images {0, 1 and 3 (inside the
    first multi-image tile), N/2, N-2, N-1} are checked against the oracle (<= 1e-4).  full_ref: the
    generic kernel's output for the same input (_generic_reference): ALL N images are compared with it
    on the device, same tolerance."""
    dev = torch.device("cuda:0")
    w, b = synth.pruned_weights(s, seed), synth.bias_vector(s, seed + 1)
    own = plan is None
    if own:
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_AUTO if kernel is None else kernel)
        plan.weight_align(w)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 2)
    x = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1
    top = plan.forward(x, torch.from_numpy(b).to(dev) if b is not None else None)
    torch.cuda.synchronize()
    if full_ref is not None:
        assert full_ref.shape == top.shape
        scale = max(1e-6, float(full_ref.abs().max()))
        full_err = float((top - full_ref).abs().max()) / scale
        assert full_err <= TOL, "%s via %s: all %d images vs the generic kernel: %g" % (s.name, plan.kernel_name, s.N, full_err)
    imgs = sorted(set(i for i in (images or (0, 1, 3, s.N // 2, s.N - 2, s.N - 1)) if 0 <= i < s.N))
    idx = torch.tensor(imgs, device=dev)
    xs, got = x[idx].cpu().numpy(), top[idx].cpu().numpy()
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                    s.dil_h, s.dil_w, s.group)
    want = oracle.conv_forward(g, xs, w, b, relu=relu, gate=False, threads=4)
    err = rel_err(got, want)
    name = plan.kernel_name
    if own:
        plan.close()
    return err, name


@pytest.mark.parametrize("which", ["lenet", "alexnet", "resnet50", "googlenet"])
def test_config_layers_at_config_batch_vs_oracle(pkg, oracle, synth, torch_cuda, which):
    """Every layer of every BASELINE.json configuration AT ITS CONFIG BATCH (LeNet 64, AlexNet 128,
    ResNet-50 256, all 39 GoogLeNet 1x1 layers at 256): the tiling the plan picks depends on the
    batch (channels per wave, images per tile, blocks per pass), so this is the weight stream and
    the kernel path the headline numbers are measured on."""
    shapes = dict(_config_sets(synth))[which]
    for k, s in enumerate(shapes):
        # all N images of the generic kernel are checked against the oracle bit for bit, all N images of both fast
        # kernels against the generic kernel's on the device, six of them with the oracle directly as well
        ref = _generic_reference(pkg, torch_cuda, s, 7000 + 10 * k, oracle=oracle)
        # the walk as code WeightAlign generated (jit_codegen.h) ...
        err, name = _check_full_batch(pkg, oracle, synth, torch_cuda, s, 7000 + 10 * k, kernel=pkg.KERNEL_JIT, full_ref=ref)
        assert "escoin_sconv_jit_kernel" in name, (s.name, name)
        assert err <= TOL, "%s @N=%d via %s: %g" % (s.name, s.N, name, err)
        # ... and the LDS-staged stream kernel on the same layer
        err, name = _check_full_batch(pkg, oracle, synth, torch_cuda, s, 7000 + 10 * k, kernel=pkg.KERNEL_TILED, full_ref=ref)
        assert "tiled" in name, (s.name, name)
        assert err <= TOL, "%s @N=%d via %s: %g" % (s.name, s.N, name, err)
        # the 3x3 layers whose planes are whole 1 KiB pieces take the instantiation that issues the
        # next block's plane DMA from inside the stream walk
        if s.name.startswith(("res3", "res4", "alex_conv3", "alex_conv4", "alex_conv5")):
            assert "tiled_dma_kernel" in name, (s.name, name)
        if s.name.startswith(("res2", "res5", "alex_conv2")):
            assert "tiled_dma_kernel" not in name, (s.name, name)


def test_per_group_dense_selection_and_conv_mode_0(pkg, oracle, synth, torch_cuda):
    """KERNEL_AUTO decides dense (fp32 MFMA) vs sparse per conv group from each group's own density
    (the reference gates the whole layer on group 0, base_conv_layer.cpp:750-755); conv_mode 0
    (LOWERED_GEMM, forward_gpu_gemm :713-746) sends every group to the dense kernel; the mode may
    be flipped on an aligned plan.  Same numbers on every route."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    s = synth.shape("mixed", 5, 24, 13, 13, 36, 3, pad=1, group=3, sparsity=0.9)
    w = synth.pruned_weights(s, 4)
    w[12:24] = synth.uniform(9, w[12:24].size).reshape(w[12:24].shape)      # group 1 fully dense
    b, x = synth.bias_vector(s, 5), synth.activations(s, 6)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, group=3)
    want = oracle.conv_forward(g, x, w, b, gate=False)
    xd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    plan.weight_align(w)
    assert _fast(plan.kernel_name) and "dense_mfma" in plan.kernel_name, plan.kernel_name
    assert rel_err(plan.forward(xd, bd).cpu().numpy(), want) <= TOL
    for mode, expect in ((pkg.CONV_MODE_LOWERED_GEMM, "escoin_dense_mfma_kernel"),
                         (pkg.CONV_MODE_LOWERED_SPARSE, "escoin_csrmm_kernel"),
                         (pkg.CONV_MODE_SCONV, None), (pkg.CONV_MODE_LOWERED_GEMM, "escoin_dense_mfma_kernel"),
                         (pkg.CONV_MODE_SCONV_PAR, None)):
        plan.set_option("conv_mode", mode)
        if expect:
            assert plan.kernel_name == expect
        else:
            assert _fast(plan.kernel_name) and "dense_mfma" in plan.kernel_name
        assert rel_err(plan.forward(xd, bd).cpu().numpy(), want) <= TOL, mode
    plan.close()
    # thresholds: 100 = never dense, 0 = any nonzero density is dense; generic kernel in a mixed layer
    for pct, kernel, has_dense in ((100, pkg.KERNEL_AUTO, False), (0, pkg.KERNEL_AUTO, True)):
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel, dense_threshold_pct=pct)
        plan.weight_align(w)
        assert ("dense_mfma" in plan.kernel_name) == has_dense
        assert (plan.kernel_name == "escoin_dense_mfma_kernel") == has_dense
        assert rel_err(plan.forward(xd, bd).cpu().numpy(), want) <= TOL
        plan.close()
    # a plan aligned in LOWERED_GEMM mode (what `caffe test -conv_mode 0` does) and flipped to SCONV
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), conv_mode=pkg.CONV_MODE_LOWERED_GEMM)
    plan.weight_align(w)
    assert plan.kernel_name == "escoin_dense_mfma_kernel"
    assert rel_err(plan.forward(xd, bd).cpu().numpy(), want) <= TOL
    plan.set_option("conv_mode", pkg.CONV_MODE_SCONV_PAR)
    assert rel_err(plan.forward(xd, bd).cpu().numpy(), want) <= TOL
    plan.close()


def test_sparsity_sweep_60_to_95(pkg, oracle, synth, torch_cuda):
    for sp in (0.6, 0.7, 0.8, 0.9, 0.95, 1.0, 0.0):
        s = synth.shape("sweep", 2, 32, 14, 14, 48, 3, pad=1, sparsity=sp)
        w, b, x = synth.pruned_weights(s, 1), synth.bias_vector(s, 2), synth.activations(s, 3)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        for kernel in _kernels(pkg, pkg.ConvDesc.from_shape(s)):
            got, name = _run(pkg, torch_cuda, pkg.ConvDesc.from_shape(s), w, x, b, kernel)
            assert rel_err(got, want) <= TOL, "sparsity %g via %s" % (sp, name)


def test_odd_geometries(pkg, oracle, synth, torch_cuda):
    S = synth.shape
    cases = [
        S("s2", 2, 8, 15, 13, 8, 3, pad=1, stride=2, sparsity=0.7),
        S("d2", 2, 8, 15, 13, 8, 3, pad=2, dil=2, sparsity=0.7),
        S("k1s2", 2, 8, 9, 9, 8, 1, stride=2, sparsity=0.5),
        S("k7p3", 1, 3, 20, 20, 8, 7, pad=3, sparsity=0.5),
        S("k3p0", 2, 8, 10, 10, 8, 3, sparsity=0.8),
        S("k3p2", 2, 4, 6, 6, 4, 3, pad=2, sparsity=0.5),
        S("g4", 2, 16, 9, 9, 12, 3, pad=1, group=4, sparsity=0.75),
        S("w1", 2, 8, 5, 1, 8, 3, KW=1, pad=1, pad_w=0, sparsity=0.5),
        S("h1", 2, 8, 1, 9, 8, 1, KW=3, pad=0, pad_w=1, sparsity=0.5),
        S("wide", 1, 4, 3, 300, 4, 3, pad=1, sparsity=0.6),
        S("n1c1m1", 1, 1, 4, 4, 1, 3, pad=1, sparsity=0.0),
        S("w5", 3, 16, 5, 5, 16, 3, pad=1, sparsity=0.8),
        S("w20k5", 2, 8, 20, 20, 8, 5, pad=2, sparsity=0.8),
        S("w33", 2, 8, 33, 33, 8, 3, pad=1, sparsity=0.8),
    ]
    for k, s in enumerate(cases):
        w, b, x = synth.pruned_weights(s, k), synth.bias_vector(s, k + 50), synth.activations(s, k + 90)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                        s.dil_h, s.dil_w, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        assert rel_err(want, naive_conv(x, w, b, s)) <= 1e-5
        for kernel in _kernels(pkg, pkg.ConvDesc.from_shape(s)):
            got, name = _run(pkg, torch_cuda, pkg.ConvDesc.from_shape(s), w, x, b, kernel)
            assert rel_err(got, want) <= TOL, "%s via %s: %g" % (s.name, name, rel_err(got, want))


def test_bias_null_relu_and_partial_batch(pkg, oracle, synth, torch_cuda):
    torch = torch_cuda
    s = synth.shape("b", 5, 16, 14, 14, 24, 3, pad=1, sparsity=0.85)
    w, b, x = synth.pruned_weights(s, 1), synth.bias_vector(s, 2), synth.activations(s, 3)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w)
    dev = torch.device("cuda:0")
    for kernel in _kernels(pkg):
        # has_bias set but bias pointer NULL is legal (reference quirk 2 is not copied)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel)
        plan.weight_align(w)
        xd = torch.from_numpy(x).to(dev)
        nb = plan.forward(xd, None).cpu().numpy()
        assert rel_err(nb, oracle.conv_forward(g, x, w, None, gate=False)) <= TOL
        # n_images < desc.N: only the first images are touched
        top = torch.full((5, s.M, 14, 14), 7.0, device=dev)
        plan.forward_ptr(xd.data_ptr(), 0, top.data_ptr(), 2,
                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        t = top.cpu().numpy()
        assert rel_err(t[:2], nb[:2]) <= 1e-6 and (t[2:] == 7.0).all()
        # ... and an empty batch is a legal no-op
        top.fill_(9.0)
        plan.forward_ptr(xd.data_ptr(), 0, top.data_ptr(), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert (top.cpu().numpy() == 9.0).all()
        with pytest.raises(pkg.EscoinError):
            plan.forward_ptr(xd.data_ptr(), 0, top.data_ptr(), 6, None)   # > desc.N
        plan.close()
        # fused bias + ReLU (ConvolutionReLU)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=True), kernel=kernel)
        plan.weight_align(w)
        r = plan.forward(xd, torch.from_numpy(b).to(dev)).cpu().numpy()
        assert rel_err(r, oracle.conv_forward(g, x, w, b, relu=True, gate=False)) <= TOL
        assert (r >= 0).all()
        plan.close()


def test_tile_store_epilogue_bias_relu_ragged_channels(pkg, oracle, synth, torch_cuda):
    """3x3 / pad 1 layers whose rows are whole quads take the one-block asm epilogue (stores straight
    from the accumulators): bias, ReLU, both, neither; channel counts that leave the last wave with
    fewer than 8 channels; widths with and without padding quads; batches that leave tile slots empty."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    S = synth.shape
    k = 0
    cases = [(3, 16, 8, 8, 20, 3), (2, 12, 28, 28, 13, 3), (2, 8, 56, 56, 9, 3), (5, 24, 12, 16, 64, 3), (1, 8, 4, 32, 7, 3)]
    # 5x5 / pad 2: every OW % 4, rows whose partial quad is / is not the last one of the LDS row
    cases += [(2, 8, 27, 27, 10, 5), (2, 8, 9, 13, 6, 5), (3, 8, 6, 16, 5, 5), (2, 8, 7, 30, 9, 5), (2, 6, 10, 22, 4, 5),
              (2, 6, 5, 21, 7, 5), (1, 4, 6, 64, 3, 5), (2, 6, 8, 31, 8, 5), (2, 6, 8, 15, 8, 5)]
    for (N, Cc, H, W, M, K) in cases:
        s = S("epi%d" % k, N, Cc, H, W, M, K, pad=K // 2, sparsity=0.8)
        w, b, x = synth.pruned_weights(s, 700 + k), synth.bias_vector(s, 710 + k), synth.activations(s, 720 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, K, K, K // 2, K // 2)
        xd = torch.from_numpy(x).to(dev)
        for relu in (False, True):
            for tb, kernel in ((0, pkg.KERNEL_TILED), (256, pkg.KERNEL_TILED), (0, pkg.KERNEL_JIT), (256, pkg.KERNEL_JIT)):
                plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=kernel, tiling_batch=tb)
                plan.weight_align(w)
                for bias in (None, b):
                    got = plan.forward(xd, None if bias is None else torch.from_numpy(bias).to(dev)).cpu().numpy()
                    want = oracle.conv_forward(g, x, w, bias, relu=relu, gate=False)
                    assert rel_err(got, want) <= TOL, (s.name, relu, tb, plan.kernel_name, bias is not None, rel_err(got, want))
                plan.close()
        k += 1


def test_uneven_channel_densities(pkg, oracle, synth, torch_cuda):
    """A pruned model's output channels differ in density (here 0 .. 60 % nonzeros, some channels
    empty): WeightAlign re-deals them over the waves block by block (stream_builder.h
    balance_channels), so slot order != channel order in the stream, the bias lanes and the store
    addresses.  3x3 (asm epilogue) and 5x5 (C++ epilogue), bias + ReLU, two conv groups."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    S = synth.shape
    rng = np.random.RandomState(7)
    for k, (N, Cc, H, W, M, K, pad, group) in enumerate([(3, 32, 14, 14, 96, 3, 1, 1), (2, 24, 28, 28, 64, 3, 1, 1),
                                                          (2, 16, 13, 13, 48, 5, 2, 2), (2, 64, 7, 7, 128, 3, 1, 1)]):
        s = S("uneven%d" % k, N, Cc, H, W, M, K, pad=pad, group=group, sparsity=0.0)
        w = synth.pruned_weights(s, 800 + k)
        keep = rng.uniform(0.0, 0.6, size=M)
        keep[rng.randint(0, M, size=3)] = 0.0
        mask = rng.uniform(size=w.shape) < keep[:, None, None, None]
        w = (w * mask).astype(np.float32)
        b, x = synth.bias_vector(s, 810 + k), synth.activations(s, 820 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, K, K, pad, pad, 1, 1, 1, 1, group)
        want = oracle.conv_forward(g, x, w, b, relu=True, gate=False)
        for kernel in (pkg.KERNEL_TILED, pkg.KERNEL_JIT):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=True), kernel=kernel, tiling_batch=256)
            plan.weight_align(w)
            got = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
            assert rel_err(got, want) <= TOL, (s.name, plan.kernel_name, rel_err(got, want))
            plan.close()


def test_weight_align_from_device_and_csr_roundtrip(pkg, oracle, synth, torch_cuda):
    torch = torch_cuda
    s = synth.alexnet(N=2)[0]                       # group = 2
    w, b, x = synth.pruned_weights(s, 1), synth.bias_vector(s, 2), synth.activations(s, 3)
    dev = torch.device("cuda:0")
    p1 = pkg.Plan(pkg.ConvDesc.from_shape(s))
    p1.weight_align(torch.from_numpy(w).to(dev))    # device blob, like blobs_[0]->gpu_data()
    rp, ci, va, ng = p1.get_csr()
    assert ng.tolist() == [synth.nnz_of(s) // 2] * 2 and p1.nnz() == synth.nnz_of(s)
    # CSR equals the reference's dense2csr + stretch (WeightAlign) bit for bit
    mg, cg = s.M // s.group, s.C // s.group
    _, ci_s, _, _ = p1.get_csr(stretched=True)
    off = 0
    for grp in range(s.group):
        orp, oci, ova = oracle.dense2csr(w[grp * mg:(grp + 1) * mg].reshape(mg, cg * s.KH * s.KW))
        n = len(oci)
        assert np.array_equal(rp[grp * (mg + 1):(grp + 1) * (mg + 1)], orp)
        assert np.array_equal(ci[off:off + n], oci) and np.array_equal(va[off:off + n], ova)
        assert np.array_equal(ci_s[off:off + n],
                              oracle.stretch(orp, oci, s.KH, s.KW, s.H, s.W, s.pad_h, s.pad_w))
        off += n
    # a second plan fed only the CSR (what a broadcast receiver does) gives identical output
    p2 = pkg.Plan(pkg.ConvDesc.from_shape(s))
    p2.set_csr(rp, ci, va, ng)
    xd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
    a = p1.forward(xd, bd).cpu().numpy()
    c = p2.forward(xd, bd).cpu().numpy()
    assert np.array_equal(a, c)
    assert p1.workspace_bytes > 0
    p1.close()
    p2.close()


def test_math_functions_level_dropins(pkg, oracle, synth, torch_cuda):
    """escoin_gpu_sparse_dense2csr / gpu_stretch / copy_input_data / gpu_sconv on the
    reference's own layouts (padded input, stretched CSR), as base_conv_layer.cpp calls them."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    L = pkg.lib()
    for s in (synth.shape("m", 3, 6, 9, 8, 5, 3, pad=1, sparsity=0.6),
              synth.shape("md", 2, 4, 9, 8, 5, 3, pad=2, dil=2, sparsity=0.5),
              synth.shape("ms", 2, 4, 9, 8, 5, 3, pad=1, stride=2, sparsity=0.5)):
        w, x = synth.pruned_weights(s, 1), synth.activations(s, 2)
        bias = synth.uniform(3, s.M, -0.1, 0.1)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                        s.dil_h, s.dil_w)
        kdim = s.C * s.KH * s.KW
        A = torch.from_numpy(w.reshape(s.M, kdim)).to(dev)
        vals = torch.zeros(s.M * kdim, device=dev)
        cols = torch.zeros(s.M * kdim, dtype=torch.int32, device=dev)
        rowp = torch.zeros(s.M + 1, dtype=torch.int32, device=dev)
        perrow = torch.zeros(s.M, dtype=torch.int32, device=dev)
        nnz = C.c_int()
        P = lambda t: C.c_void_p(t.data_ptr())
        assert L.escoin_gpu_sparse_dense2csr(s.M, kdim, P(A), P(perrow), P(vals), P(rowp), P(cols),
                                             C.byref(nnz), None) == 0
        orp, oci, ova = oracle.dense2csr(w.reshape(s.M, kdim))
        assert nnz.value == len(oci)
        assert np.array_equal(rowp.cpu().numpy(), orp)
        assert np.array_equal(cols.cpu().numpy()[:nnz.value], oci)
        assert np.array_equal(vals.cpu().numpy()[:nnz.value], ova)
        assert np.array_equal(perrow.cpu().numpy(), np.diff(orp))
        assert L.escoin_gpu_stretch(P(rowp), P(cols), s.M, s.H, s.W, s.pad_h, s.pad_w, s.KH, s.KW,
                                    None) == 0
        ocs = oracle.stretch(orp, oci, s.KH, s.KW, s.H, s.W, s.pad_h, s.pad_w)
        assert np.array_equal(cols.cpu().numpy()[:nnz.value], ocs)
        plen = oracle.padded_len(g)
        N = x.shape[0]
        ifmap = s.C * (s.H + s.pad_h) * (s.W + s.pad_w)
        padded = torch.zeros(N * ifmap + plen, device=dev)       # SCONV_PAR stride, quirk 9
        xd = torch.from_numpy(x).to(dev)
        for n in range(N):
            assert L.escoin_copy_input_data(C.c_void_p(padded.data_ptr() + 4 * n * ifmap),
                                            C.c_void_p(xd.data_ptr() + 4 * n * s.C * s.H * s.W),
                                            s.C, s.H, s.W, s.pad_h, s.pad_w, None) == 0
        torch.cuda.synchronize()
        pn = padded.cpu().numpy()
        for n in range(N):      # image n's tail overlaps image n+1's leading halo (quirk 9)
            assert np.array_equal(pn[n * ifmap:(n + 1) * ifmap], oracle.pad_input(g, x[n])[:ifmap])
        assert not pn[N * ifmap:].any()
        oh, ow = oracle.out_hw(g)
        out = torch.zeros(N, s.M, oh, ow, device=dev)
        bd = torch.from_numpy(bias).to(dev)
        for relu in (0, 1):
            assert L.escoin_gpu_sconv(relu, N, P(padded), ifmap, P(rowp), P(cols), P(vals), P(bd),
                                      s.H, s.W, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h,
                                      s.dil_w, s.KH, s.KW, P(out), s.M, 1, None) == 0
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            base = oracle.conv_forward(g, x, w, None, gate=False)
            if relu:
                want = np.maximum(base + bias[None, :, None, None], 0)
                assert rel_err(got, want) <= 1e-6
            else:
                assert np.array_equal(got, base)      # non-ReLU kernels ignore bias (quirk 10)


def test_linearity_and_batch_independence_full_size(pkg, synth, torch_cuda):
    """Size-independent properties at BASELINE's full batch (res3 N=256): conv(a*x + y) ==
    a*conv(x) + conv(y) and image n's output does not depend on its batch neighbours."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    s = synth.resnet50_3x3(N=256)[1]
    w = synth.pruned_weights(s, 1)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    plan.weight_align(w)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1
    y = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1
    fx, fy = plan.forward(x), plan.forward(y)
    fz = plan.forward(2.0 * x + y)
    scale = float(fz.abs().max())
    assert float((fz - (2.0 * fx + fy)).abs().max()) / scale <= TOL
    # batch independence: a shuffled batch gives the shuffled outputs, bit for bit
    perm = torch.randperm(s.N, device=dev, generator=gen)
    assert torch.equal(plan.forward(x[perm].contiguous()), fx[perm])
    # and a generic-kernel plan agrees on a slice at full channel count
    ref = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_GENERIC)
    ref.weight_align(w)
    fr = ref.forward(x[:8].contiguous())
    assert float((fr - fx[:8]).abs().max()) / scale <= TOL
    plan.close()
    ref.close()


def test_dense_gate_follows_the_reference_threshold(pkg, oracle, synth, torch_cuda):
    """dense_gate=1 reproduces forward_gpu_sconv's gate (base_conv_layer.cpp:750-755): density of
    group 0 > 0.2 -> dense (MFMA) path, otherwise the sparse kernels; same numbers either way."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    for sparsity, want_dense in ((0.70, True), (0.80, False), (0.90, False), (0.0, True)):
        s = synth.shape("gate", 3, 32, 14, 14, 40, 3, pad=1, sparsity=sparsity)
        w, b, x = synth.pruned_weights(s, 1), synth.bias_vector(s, 2), synth.activations(s, 3)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        plan.set_option("dense_gate", 1)
        plan.weight_align(w)
        assert ("dense_mfma" in plan.kernel_name) == want_dense, (sparsity, plan.kernel_name)
        got = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
        assert rel_err(got, want) <= TOL
        plan.close()
    # the gate looks at group 0 only (reference quirk 5): dense group 0, sparse group 1
    s = synth.shape("gate2", 2, 16, 9, 9, 16, 3, pad=1, group=2, sparsity=0.9)
    w = synth.pruned_weights(s, 4)
    w[:8] = synth.uniform(9, w[:8].size).reshape(w[:8].shape)      # group 0 fully dense
    x = synth.activations(s, 5)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, group=2)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    plan.set_option("dense_gate", 1)
    plan.weight_align(w)
    assert "dense_mfma" in plan.kernel_name
    got = plan.forward(torch.from_numpy(x).to(dev), None).cpu().numpy()
    assert rel_err(got, oracle.conv_forward(g, x, w, None, gate=False)) <= TOL
    plan.close()


@pytest.mark.parametrize("path", golden_params())
def test_lowered_sparse_comparator_is_bit_exact(pkg, torch_cuda, path):
    """conv_mode LOWERED_SPARSE (im2col + CSR x dense, base_conv_layer.cpp:724-736): one fmaf per
    nonzero in CSR order from 0, bias added once -- the oracle's arithmetic, so bit-exact."""
    torch = torch_cuda
    gf = Golden(path)
    dev = torch.device("cuda:0")
    plan = pkg.Plan(gf.desc(pkg), conv_mode=pkg.CONV_MODE_LOWERED_SPARSE)
    plan.weight_align(gf.w)
    assert plan.kernel_name == "escoin_csrmm_kernel"
    top = plan.forward(torch.from_numpy(gf.x).to(dev),
                       torch.from_numpy(gf.bias).to(dev) if gf.bias is not None else None)
    torch.cuda.synchronize()
    got = top.cpu().numpy()
    assert got.shape == gf.top.shape
    assert np.array_equal(got.view(np.uint32), gf.top.view(np.uint32)), rel_err(got, gf.top)
    # the mode can be flipped on an aligned plan: back to the direct path
    plan.set_option("conv_mode", pkg.CONV_MODE_SCONV_PAR)
    assert plan.kernel_name != "escoin_csrmm_kernel"
    top2 = plan.forward(torch.from_numpy(gf.x).to(dev),
                        torch.from_numpy(gf.bias).to(dev) if gf.bias is not None else None)
    torch.cuda.synchronize()
    assert rel_err(top2.cpu().numpy(), gf.top) <= TOL
    plan.close()


def test_gpu_sparse_csrmm_entry_point(pkg, synth, torch_cuda):
    """escoin_gpu_sparse_csrmm (caffe_gpu_sparse_csrmm, math_functions.cu:48-62) against numpy,
    ragged sizes, alpha / beta."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5)
    for (M, N, K, dens) in [(7, 13, 9, 0.5), (64, 784, 576, 0.1), (3, 1, 4, 1.0), (5, 260, 33, 0.0), (16, 1024, 64, 0.3)]:
        A = (rng.rand(M, K) < dens) * rng.uniform(-1, 1, (M, K))
        A = A.astype(np.float32)
        rowptr = np.zeros(M + 1, np.int32)
        cols, vals = [], []
        for m in range(M):
            nz = np.nonzero(A[m])[0]
            cols.extend(nz.tolist())
            vals.extend(A[m, nz].tolist())
            rowptr[m + 1] = len(cols)
        B = rng.uniform(-1, 1, (K, N)).astype(np.float32)
        C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
        d_rp, d_ci, d_va = t(rowptr, np.int32), t(np.array(cols + [0], np.int32), np.int32), t(np.array(vals + [0], np.float32), np.float32)
        d_B, d_C = t(B, np.float32), t(C0, np.float32)
        for (alpha, beta) in [(1.0, 0.0), (0.5, 2.0)]:
            d_C.copy_(t(C0, np.float32))
            pkg.check(pkg.lib().escoin_gpu_sparse_csrmm(M, N, K, len(cols), alpha, C.c_void_p(d_va.data_ptr()),
                                                        C.c_void_p(d_rp.data_ptr()), C.c_void_p(d_ci.data_ptr()),
                                                        C.c_void_p(d_B.data_ptr()), beta, C.c_void_p(d_C.data_ptr()), None),
                      "escoin_gpu_sparse_csrmm")
            torch.cuda.synchronize()
            want = alpha * (A.astype(np.float64) @ B.astype(np.float64)) + beta * C0
            assert rel_err(d_C.cpu().numpy(), want.astype(np.float32)) <= 1e-5, (M, N, K, alpha, beta)


@pytest.mark.parametrize("seed", [20260117, 7, 424242])
def test_randomised_tiled_geometries(pkg, oracle, synth, torch_cuda, seed):
    """Seeded random shapes for the tiled kernel's stream walk: dense rows (groups of 4-6 records,
    rows split over several groups), all-pruned channel blocks (empty units), channel counts that
    leave waves idle or half full, batches that leave tile slots empty, every kernel width."""
    rng = np.random.RandomState(seed)
    S = synth.shape
    checked = 0
    for k in range(48):
        K = int(rng.choice([1, 1, 2, 3, 3, 3, 4, 5]))
        pad = int(rng.randint(0, K)) if K > 1 else 0
        H = int(rng.randint(max(K - 2 * pad, 1), 40))
        W = int(rng.randint(max(K - 2 * pad, 1), 70))
        group = int(rng.choice([1, 1, 1, 2, 3]))
        C = group * int(rng.randint(1, 24))
        M = group * int(rng.randint(1, 40))
        N = int(rng.randint(1, 12))
        sp = float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.97, 1.0]))
        s = S("rnd%d" % k, N, C, H, W, M, K, pad=pad, group=group, sparsity=sp, bias=bool(rng.randint(2)))
        w, b, x = synth.pruned_weights(s, 300 + k), synth.bias_vector(s, 400 + k), synth.activations(s, 500 + k)
        if sp == 0.9 and C // group > 2:
            w[:, : (C // group) // 2] = 0.0      # whole input-channel blocks without a nonzero
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        # a third of the cases each: the tiling of their own (small) batch, of a batch of 64 and of
        # a batch of 256 -- the last two give channel groups of 4-24 per wave, several images per
        # tile and second payload quads, i.e. the streams of the benchmarked configurations
        tb = (0, 64, 256)[k % 3]
        for kernel in (pkg.KERNEL_JIT, pkg.KERNEL_TILED, pkg.KERNEL_AUTO):      # generated code / LDS-staged stream / the plan's own pick
            if kernel != pkg.KERNEL_AUTO and not _tiled_ok(pkg.ConvDesc.from_shape(s)):
                continue
            got, name = _run(pkg, torch_cuda, pkg.ConvDesc.from_shape(s), w, x, b, kernel,
                             tiling_batch=tb, dense_threshold_pct=100)
            assert rel_err(got, want) <= TOL, "%s %s tb=%d via %s: %g" % (s.name, (N, C, H, W, M, K, pad, group, sp), tb, name, rel_err(got, want))
            checked += "jit" in name
    assert checked >= 40      # nearly all of these must have gone down the generated-code path at least once


def test_batches_beyond_one_buffer_descriptor(pkg, oracle, synth):
    """The plane DMA addresses the input through a 32-bit buffer descriptor; larger batches are
    served by consecutive sub-batch launches.  The plan option "max_launch_bytes" lowers the limit to a
    few images' worth of bytes so that the split is exercised on a small problem (in a child process, as
    before the limit became a plan option)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package(); oracle = ge.load_oracle(); synth = pkg.synth
s = synth.shape("chunked", 11, 6, 9, 10, 8, 3, pad=1, sparsity=0.7)
w, b, x = synth.pruned_weights(s, 1), synth.bias_vector(s, 2), synth.activations(s, 3)
dev = torch.device("cuda:0")
g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
want = oracle.conv_forward(g, x, w, b, gate=False)
err = 0.0
for kernel in (pkg.KERNEL_TILED, pkg.KERNEL_JIT):
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel, max_launch_bytes=3 * 6 * 9 * 10 * 4 + 100)   # three images per launch
    plan.weight_align(w)
    top = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    err = max(err, float(np.abs(top - want).max() / max(1e-6, np.abs(want).max())))
print("REL_ERR %%g" %% err)
sys.exit(0 if err <= 1e-4 else 1)
""" % root
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=600)
    assert out.returncode == 0, out.stdout.decode()
    assert "REL_ERR" in out.stdout.decode()


def test_dense_pointwise_k_tail_never_reads_its_neighbours(pkg, oracle, synth, torch_cuda):
    """Dense (fp32 MFMA) kernel on a pointwise layer whose channels per group are NOT a multiple of
    the 32-wide k-step (GoogLeNet 4e: 528): the last k-step's rows >= K are channels of the next
    conv group, of the next image, or memory past the blob.  The reference GEMM never reads them
    (base_conv_layer.cpp:713-746); a zero weight does not make them harmless (0 * NaN = NaN), so
    the staging must deliver zeros there.  NaNs are planted in all three places."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    # two groups of 40 channels (tail of 24 rows in the second k-step), 8 x 8 = whole quads of pixels
    s = synth.shape("k_tail", 3, 80, 8, 8, 32, 1, group=2, sparsity=0.3, bias=True)
    w, b = synth.pruned_weights(s, 91), synth.bias_vector(s, 92)
    x = synth.activations(s, 93)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                    s.dil_h, s.dil_w, s.group)
    want = oracle.conv_forward(g, x, w, b, gate=False)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_DENSE)
    plan.weight_align(w)
    assert "dense_mfma" in plan.kernel_name
    bias = torch.from_numpy(b).to(dev)
    per_img = s.C * s.H * s.W
    # (1) after the blob: the bottom is a window of a NaN-filled allocation
    big = torch.full((5 * per_img,), float("nan"), device=dev)
    big[per_img:4 * per_img] = torch.from_numpy(x).to(dev).reshape(-1)
    bottom = big[per_img:4 * per_img].view(3, s.C, s.H, s.W)
    got = plan.forward(bottom, bias).cpu().numpy()
    assert np.isfinite(got).all(), "rows past K reached memory after the blob"
    assert rel_err(got, want) <= TOL
    # (2) the next image: only image 0 is computed, image 1 is all NaN
    xs = torch.from_numpy(x).to(dev).clone()
    xs[1:] = float("nan")
    top = torch.zeros((3, s.M, s.H, s.W), device=dev)
    plan.forward_ptr(xs.data_ptr(), bias.data_ptr(), top.data_ptr(), 1,
                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got0 = top[0].cpu().numpy()
    assert np.isfinite(got0).all(), "rows past K of the last group reached the next image"
    assert rel_err(got0, want[0]) <= TOL
    # (3) the next conv group: group 1's channels are NaN, group 0's outputs must not notice
    xg = torch.from_numpy(x).to(dev).clone()
    xg[:, s.C // 2:] = float("nan")
    gotg = plan.forward(xg, bias).cpu().numpy()[:, :s.M // 2]
    assert np.isfinite(gotg).all(), "rows past K of group 0 reached group 1's channels"
    assert rel_err(gotg, want[:, :s.M // 2]) <= TOL
    plan.close()


def test_global_batch_2048_on_one_gpu(pkg, oracle, synth, torch_cuda):
    """BASELINE.json configs[3] at its strong-scaling N = 1 point: the whole batch of 2048 on ONE
    GPU, real size (res3 shape: 822 MB bottom, 822 MB top; res2 shape: 1.6 GB each, with the tiling
    chosen for N = 2048).  Images 0, 1023 and 2047 are checked against the oracle, and the batch
    independence of the rest through per-image checksums of a repeated image."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    for k, idx in enumerate((1, 0)):
        s = synth.resnet50_3x3(N=2048)[idx]
        w = synth.pruned_weights(s, 8100 + k)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        plan.weight_align(w)
        err, name = _check_full_batch(pkg, oracle, synth, torch, s, 8100 + k, plan=plan,
                                      images=(0, 1023, 2047))
        assert _fast(name) and err <= TOL, "%s @N=2048 via %s: %g" % (s.name, name, err)
        # every image through the same arithmetic: a batch made of ONE image repeated 2048 times
        # must give 2048 identical outputs (bitwise), wherever the image falls in a tile
        gen = torch.Generator(device=dev)
        gen.manual_seed(8200 + k)
        one = torch.rand((1, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1
        x = one.expand(s.N, -1, -1, -1).contiguous()
        top = plan.forward(x)
        torch.cuda.synchronize()
        ref = top[0:1]
        same = (top == ref).all(dim=(1, 2, 3))
        assert bool(same.all()), "%s: images %s differ from image 0" % (
            s.name, torch.nonzero(~same).flatten()[:8].tolist())
        del x, top
        plan.close()
    torch.cuda.empty_cache()


def test_pointwise_layers_with_one_quad_per_lane(pkg, oracle, synth, torch_cuda):
    """Pointwise layers with 193 .. 384 output channels on small images: generated code gives a lane ONE
    quad and the whole accumulator file to it (up to 48 channels per wave, one workgroup column instead of
    two; Tiling::tpl, stream_builder.h).  Slots 24 .. 47 of a wave live in what are tile B's registers
    elsewhere and leave through their own epilogue paths: the asm one (output rows of whole quads,
    14 x 14 walked as 1 x 196) and the element-wise one (13 x 13 as 1 x 169), with bias, with fused ReLU,
    with a ragged last slot range -- whole batch on the GPU, six images each against the oracle."""
    cases = [synth.shape("pw14_256", 256, 512, 14, 14, 256, 1, sparsity=0.95),
             synth.shape("pw13_256", 256, 256, 13, 13, 256, 1, sparsity=0.95),
             synth.shape("pw13_300", 256, 832, 13, 13, 300, 1, sparsity=0.95, bias=False),
             synth.shape("pw14_200", 256, 512, 14, 14, 200, 1, sparsity=0.95)]
    for k, s in enumerate(cases):
        for relu in (False, True):
            # (the ReLU runs also write the top blob with non-temporal stores: option "stream_stores")
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=pkg.KERNEL_JIT, stream_stores=int(relu))
            plan.weight_align(synth.pruned_weights(s, 9100 + k))
            info = plan.tiling_info
            assert "generated-code" in info and "tpl=1" in info and "columns=1" in info, (s.name, info)
            err, name = _check_full_batch(pkg, oracle, synth, torch_cuda, s, 9100 + k, plan=plan, relu=relu)
            assert err <= TOL, "%s via %s: %g (%s)" % (s.name, name, err, info)
            if k == 0:      # a partial batch on the same plan: tiles past the last image store nothing
                dev = torch_cuda.device("cuda:0")
                x5 = torch_cuda.from_numpy(synth.activations(s._replace(N=5), 9200)).to(dev)
                b = synth.bias_vector(s, 9100 + k + 1)
                guard = torch_cuda.full((6, s.M, s.H, s.W), 7.0, device=dev)
                plan.forward(x5, torch_cuda.from_numpy(b).to(dev), guard[:5])
                g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0, 1, 1, 1, 1, 1)
                want = oracle.conv_forward(g, x5.cpu().numpy(), synth.pruned_weights(s, 9100 + k), b, relu=relu, gate=False)
                assert rel_err(guard[:5].cpu().numpy(), want) <= TOL
                assert float(guard[5].min()) == 7.0 and float(guard[5].max()) == 7.0
            plan.close()


def test_stream_stores_option_changes_nothing_but_the_store_instruction(pkg, oracle, synth, torch_cuda):
    """Plan option "stream_stores" (non-temporal stores of a pointwise layer's top blob): same results, bit
    for bit, on the asm epilogue (rows of whole quads), the element-wise one (7 x 7) and both tiles."""
    dev = torch_cuda.device("cuda:0")
    for k, s in enumerate([synth.shape("ss28", 8, 48, 28, 28, 40, 1, sparsity=0.9),
                           synth.shape("ss7", 40, 64, 7, 7, 72, 1, sparsity=0.9),
                           synth.shape("ss56", 3, 16, 56, 56, 24, 1, sparsity=0.8, bias=False)]):
        w, b = synth.pruned_weights(s, 9300 + k), synth.bias_vector(s, 9400 + k)
        x = torch_cuda.from_numpy(synth.activations(s, 9500 + k)).to(dev)
        bias = torch_cuda.from_numpy(b).to(dev) if b is not None else None
        outs = []
        for ss in (0, 1):
            for kernel in (pkg.KERNEL_JIT, pkg.KERNEL_TILED):
                plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel, stream_stores=ss)
                plan.weight_align(w)
                outs.append(plan.forward(x, bias).cpu().numpy())
                plan.close()
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
        want = oracle.conv_forward(g, synth.activations(s, 9500 + k), w, b, gate=False)
        assert rel_err(outs[0], want) <= TOL
        assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3]), s.name


def test_padding_in_one_direction_only(pkg, oracle, synth, torch_cuda):
    """pad_h == 0 < pad_w (and the reverse), where the reference's padded buffer ends short of its own reads
    (oracle/sconv_oracle.c oracle_padded_len): every kernel family pads with zeros, the generic one bit for bit."""
    for k, (KH, KW, ph, pw, st) in enumerate([(3, 3, 0, 2, 1), (1, 5, 0, 4, 1), (7, 7, 0, 6, 1), (2, 5, 0, 4, 2), (4, 4, 0, 3, 1),
                                              (3, 3, 2, 0, 1), (5, 1, 3, 0, 1), (1, 3, 0, 2, 3)]):
        s = synth.shape("p1_%d" % k, 5, 12, 9, 23, 20, KH, KW=KW, pad=ph, pad_w=pw, stride=st, sparsity=0.8)
        w, b, x = synth.pruned_weights(s, 7100 + k), synth.bias_vector(s, 7200 + k), synth.activations(s, 7300 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, 1, 1, 1)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        assert rel_err(want, naive_conv(x, w, b, s)) <= 1e-5
        for kernel in _kernels(pkg, pkg.ConvDesc.from_shape(s)):
            got, name = _run(pkg, torch_cuda, pkg.ConvDesc.from_shape(s), w, x, b, kernel, tiling_batch=(0, 256)[k & 1])
            if "generic" in name:
                assert np.array_equal(got, want), (s.name, KH, KW, ph, pw)
            assert rel_err(got, want) <= TOL, "%s via %s" % ((KH, KW, ph, pw, st), name)


def test_forward_is_capturable_into_a_hip_graph(pkg, oracle, synth, torch_cuda):
    """escoin_forward launches on the stream it is given and neither synchronises nor allocates: a host framework may
    capture its forward pass (every kernel family here) into a HIP graph and replay it on new bottom data
    (tools/graph_step.py times a whole step that way: within 1 % of launching layer by layer)."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    cases = [(synth.shape("g3", 5, 24, 14, 14, 40, 3, pad=1, sparsity=0.9), pkg.KERNEL_JIT, {"tiling_batch": 256}),
             (synth.shape("g1", 6, 64, 7, 7, 72, 1, sparsity=0.95), pkg.KERNEL_AUTO, {"tiling_batch": 256}),
             (synth.shape("gs", 4, 16, 12, 12, 24, 3, pad=1, sparsity=0.8), pkg.KERNEL_TILED, {}),
             (synth.shape("gd", 3, 32, 14, 14, 64, 1, sparsity=0.0), pkg.KERNEL_DENSE, {}),
             # a dense pointwise layer whose launch splits K across workgroups (stream-K: 392 tiles on 512 slots); its
             # workspace is allocated at WeightAlign, so even its FIRST forward may be the captured one
             (synth.shape("gk", 64, 1024, 7, 7, 512, 1, bias=True, sparsity=0.0), pkg.KERNEL_DENSE, {}),
             (synth.shape("gg", 3, 8, 9, 9, 12, 3, pad=1, stride=2, sparsity=0.9), pkg.KERNEL_GENERIC, {})]
    layers = []
    for k, (s, kernel, opts) in enumerate(cases):
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel, **opts)
        w, b = synth.pruned_weights(s, 8100 + k), synth.bias_vector(s, 8200 + k)
        plan.weight_align(w)
        x = torch.zeros((s.N, s.C, s.H, s.W), device=dev)
        y = torch.zeros((s.N, s.M) + tuple(plan.out_hw), device=dev)
        layers.append((s, plan, w, b, x, torch.from_numpy(b).to(dev) if b is not None else None, y))

    def step():
        for s, plan, w, b, x, bd, y in layers:
            plan.forward(x, bd, y)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s, plan, w, b, x, bd, y in layers:
            if s.name != "gk":      # warm-up outside the capture -- except the stream-K layer: its first forward IS the captured one
                plan.forward(x, bd, y)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    assert [p_.stat("streamk") for s_, p_, *_ in layers if s_.name == "gk"] == [1]
    for rnd in range(2):            # two replays on different bottom data
        for k, (s, plan, w, b, x, bd, y) in enumerate(layers):
            x.copy_(torch.from_numpy(synth.activations(s, 8300 + 10 * rnd + k)).to(dev))
            y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        for k, (s, plan, w, b, x, bd, y) in enumerate(layers):
            geo = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, 1, 1, 1)
            want = oracle.conv_forward(geo, synth.activations(s, 8300 + 10 * rnd + k), w, b, gate=False)
            assert rel_err(y.cpu().numpy(), want) <= TOL, (s.name, plan.kernel_name, rnd)
            # (a stream-K give-up inside a replay is not seen by escoin_forward -- the host replays, it does not call;
            #  the sticky word is what a replaying host polls)
            assert plan.stat("streamk_gave_up") == 0
    for s, plan, *_ in layers:
        plan.close()


def test_seeded_slice_of_the_parity_fuzzer(pkg, oracle, synth, torch_cuda):
    """tools/fuzz_parity.py (strides, dilations, non-square kernels and pads, conv groups, many channels, skewed
    sparsity, fused ReLU, foreign tiling batches; every kernel family against the oracle, the generic kernel bit for
    bit): 240 cases of one seed here, tens of thousands per round on the GPU box (profiles/r05_fuzz.md)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fuzz_parity
    with open(os.devnull, "w") as sink:
        ran, lines, by_name = fuzz_parity.fuzz(240, 5, out=sink)
    assert not lines, "\n".join(lines[:10])
    assert ran >= 800 and sum(v for n, v in by_name.items() if "jit" in n) >= 150


def test_randomised_pointwise_layers_with_many_output_channels(pkg, oracle, synth, torch_cuda):
    """Seeded random pointwise layers with 130 .. 400 output channels on small images, tiled as for a
    batch of 256 (one image or a few per workgroup, packed single-row planes, no tile B, one quad per
    lane with up to 48 channels per wave where it saves a workgroup column) but run on 1 .. 9 images:
    every image of the small batch is checked, with and without bias, ReLU and streaming stores."""
    rng = np.random.RandomState(77)
    S = synth.shape
    tpl1 = 0
    for k in range(24):
        H = int(rng.choice([3, 4, 5, 6, 7, 9, 12, 13, 14, 15]))
        W = H if rng.randint(3) else int(rng.randint(2, 17))
        C = int(rng.randint(40, 200))
        M = int(rng.randint(130, 401))
        N = int(rng.randint(1, 10))
        sp = float(rng.choice([0.9, 0.95, 0.97]))
        relu = bool(rng.randint(2))
        s = S("pwr%d" % k, N, C, H, W, M, 1, sparsity=sp, bias=bool(rng.randint(2)))
        w, b, x = synth.pruned_weights(s, 9600 + k), synth.bias_vector(s, 9700 + k), synth.activations(s, 9800 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0, 1, 1, 1, 1, 1)
        want = oracle.conv_forward(g, x, w, b, relu=relu, gate=False)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=pkg.KERNEL_JIT, tiling_batch=256,
                        stream_stores=int(rng.randint(2)))
        plan.weight_align(w)
        info = plan.tiling_info
        tpl1 += "tpl=1" in info
        dev = torch_cuda.device("cuda:0")
        got = plan.forward(torch_cuda.from_numpy(x).to(dev), torch_cuda.from_numpy(b).to(dev) if b is not None else None).cpu().numpy()
        plan.close()
        assert rel_err(got, want) <= TOL, "%s %s: %g (%s)" % (s.name, (N, C, H, W, M, sp, relu), rel_err(got, want), info)
    assert tpl1 >= 4, tpl1        # some of them must have taken one quad per lane


def test_dense_stream_k_fixup(pkg, oracle, synth, torch_cuda):
    """The dense MFMA kernel splits K across workgroups (stream-K) where whole output tiles would leave more than
    10 % of its 512 slots idle -- the ResNet-50 chain's 14 x 14 and 7 x 7 1x1 layers (784 / 392 tiles) -- and fixes
    the cut tiles up inside the launch (partial sums handed over write-through, flags, agent-scope acquire).  All N
    images against the generic kernel (bit-exact to the oracle) on the device, image 0 / N-1 against the oracle;
    pointwise and gathered (3x3) operand paths, ragged tile edges, repeated launches (the flags are re-armed)."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    cases = [synth.shape("res5_2a_like", 256, 2048, 7, 7, 512, 1, bias=False, sparsity=0.0),
             synth.shape("res4_2a_like", 256, 1024, 14, 14, 256, 1, bias=True, sparsity=0.0),
             synth.shape("res4_3x3_dense", 256, 256, 14, 14, 256, 3, pad=1, bias=False, sparsity=0.0),
             synth.shape("ragged", 80, 320, 13, 13, 200, 1, bias=True, sparsity=0.0)]
    for k, s in enumerate(cases):
        w = (synth.pruned_weights(s, 900 + k) * np.float32(0.05)).astype(np.float32)
        b = synth.bias_vector(s, 950 + k)
        gen = torch.Generator(device=dev)
        gen.manual_seed(970 + k)
        x = torch.rand((s.N, s.C, s.H, s.W), device=dev, generator=gen) * 2 - 1
        bd = torch.from_numpy(b).to(dev) if b is not None else None
        ref_plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=True), kernel=pkg.KERNEL_GENERIC)
        ref_plan.weight_align(w)
        ref = ref_plan.forward(x, bd)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=True), kernel=pkg.KERNEL_DENSE)
        plan.weight_align(w)
        for rep in range(3):
            top = plan.forward(x, bd)
            torch.cuda.synchronize()
            if s.KH == 1:
                assert plan.stat("streamk") == 1, s.name             # the schedule this test is about
            # (the 3x3 case runs whole tiles unless ESCOIN_DENSE_STREAMK=1 forces the split: its operand path gathers)
            assert plan.stat("streamk_gave_up") == 0, s.name
            scale = max(1e-6, float(ref.abs().max()))
            err = float((top - ref).abs().max()) / scale
            assert err <= TOL, "%s launch %d: all %d images vs the generic kernel: %g" % (s.name, rep, s.N, err)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, group=s.group)
        idx = torch.tensor([0, s.N - 1], device=dev)
        want = oracle.conv_forward(g, x[idx].cpu().numpy(), w, b, relu=True, gate=False, threads=2)
        assert rel_err(top[idx].cpu().numpy(), want) <= TOL, s.name
        # a partial batch changes the tile count: another cut, the same numbers
        n2 = s.N // 2 + 3
        top2 = plan.forward(x[:n2].contiguous(), bd)
        torch.cuda.synchronize()
        assert float((top2 - ref[:n2]).abs().max()) / scale <= TOL, s.name
        assert plan.stat("streamk_gave_up") == 0
        plan.close()
        ref_plan.close()


def test_kernel_auto_follows_the_measured_crossover(pkg, synth, torch_cuda):
    """What KERNEL_AUTO resolves to, per BASELINE shape (profiles/r04_crossover.md, r05_crossover.md): generated code at every point of
    the 50-95 % sparsity sweep of the ResNet-50, AlexNet and GoogLeNet sets (it is ahead of the stream kernel and of
    the dense kernel at each of them); the dense MFMA kernel for unpruned layers; above 50 % density whatever the
    two-kernel cost model says (res2 stays sparse down to 10 % sparsity, AlexNet conv3 goes dense at 30 %); layers the
    tiled kernels do not cover (stride 2) go dense above 4 % density, to the generic kernel below."""
    gl = synth.googlenet_1x1(N=256)
    layers = synth.resnet50_3x3(N=256) + synth.alexnet(N=128) + [gl[0], gl[1], gl[5], gl[9], gl[25], gl[33], gl[37]]
    for s in layers:
        for sp in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
            ss = s._replace(sparsity=sp, N=2)
            plan = pkg.Plan(pkg.ConvDesc.from_shape(ss), tiling_batch=s.N)
            plan.weight_align(synth.pruned_weights(ss, 3))
            assert plan.stat("kernel_choice") == pkg.KERNEL_JIT, (s.name, sp, plan.kernel_name)
            plan.close()
        ss = s._replace(sparsity=0.0, N=2)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(ss), tiling_batch=s.N)
        plan.weight_align(synth.pruned_weights(ss, 3))
        assert plan.stat("kernel_choice") == pkg.KERNEL_DENSE, (s.name, plan.kernel_name)
        plan.close()
    # above the 50 % density cut the model decides, shape by shape
    def choice(s, sp):
        ss = s._replace(sparsity=sp, N=2)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(ss), tiling_batch=s.N)
        plan.weight_align(synth.pruned_weights(ss, 3))
        c = plan.stat("kernel_choice")
        plan.close()
        return c
    rn, al = synth.resnet50_3x3(N=256), synth.alexnet(N=128)
    # (profiles/r05_crossover.md: the model's constants were re-fitted to round 5's generated code)
    assert choice(rn[0], 0.1) == pkg.KERNEL_JIT          # res2: 588 us against 782 dense
    assert choice(rn[2], 0.2) == pkg.KERNEL_JIT          # res4: 583 against 688
    assert choice(rn[3], 0.1) == pkg.KERNEL_DENSE        # res5: 682 against 636 (the model's worst cell: 7 % behind)
    assert choice(al[1], 0.3) == pkg.KERNEL_DENSE        # AlexNet conv3: 369 against 413
    assert choice(al[2], 0.1) == pkg.KERNEL_JIT          # AlexNet conv4: 405 against 434
    assert choice(gl[5], 0.3) == pkg.KERNEL_DENSE        # inception_3b 256@28x28 -> 128: 139 against 149
    assert choice(gl[5], 0.4) == pkg.KERNEL_JIT          # ... 137 against 139
    # stride 2, 1x1 (ResNet-50's res3a_branch2a, pruned): the pointwise path over a strided view of the bottom blob
    s2 = synth.shape("res3a_branch2a", 2, 256, 56, 56, 128, 1, stride=2, bias=False, sparsity=0.9)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s2), tiling_batch=256)
    plan.weight_align(synth.pruned_weights(s2, 3))
    assert plan.stat("kernel_choice") == pkg.KERNEL_JIT, plan.kernel_name
    plan.close()
    # a geometry no tiled kernel covers (3x3, stride 2): dense above 4 % density, the generic kernel below
    s2 = synth.shape("k3s2", 2, 64, 28, 28, 64, 3, stride=2, pad=1, bias=False, sparsity=0.9)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s2), tiling_batch=256)
    plan.weight_align(synth.pruned_weights(s2, 3))
    assert plan.stat("kernel_choice") == pkg.KERNEL_DENSE, plan.kernel_name
    plan.close()
    s3 = s2._replace(sparsity=0.98)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s3), tiling_batch=256)
    plan.weight_align(synth.pruned_weights(s3, 3))
    assert plan.stat("kernel_choice") == pkg.KERNEL_GENERIC, plan.kernel_name
    plan.close()


def test_strided_pointwise_layers(pkg, oracle, synth, torch_cuda):
    """1x1 convolutions with stride 2 (ResNet-50's res{3,4,5}a_branch1 / branch2a) on the tiled path: even input rows
    staged whole, elements 0 and 2 of a lane's quad stored.  Generated code and the stream kernel against the oracle
    at small batch through the tiling of batch 256; odd heights, bias + ReLU, few channels (waves without an
    oc-group: unchained units), conv groups; widths that are not whole quads fall back (dense / generic kernel)."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    cases = [synth.shape("res3a_branch2a", 5, 256, 56, 56, 128, 1, stride=2, bias=False, sparsity=0.9),
             synth.shape("res4a_branch1", 5, 512, 28, 28, 1024, 1, stride=2, bias=False, sparsity=0.9),
             synth.shape("res5a_branch2a", 7, 1024, 14, 14, 512, 1, stride=2, bias=True, sparsity=0.9),
             synth.shape("odd_h", 3, 24, 13, 20, 40, 1, stride=2, bias=True, sparsity=0.8),
             synth.shape("w54", 3, 24, 9, 54, 40, 1, stride=2, bias=True, sparsity=0.8),
             synth.shape("few_ch", 3, 16, 12, 12, 5, 1, stride=2, bias=True, sparsity=0.5),
             synth.shape("groups", 3, 32, 8, 16, 48, 1, stride=2, group=2, bias=True, sparsity=0.85)]
    for k, s in enumerate(cases):
        w, b = synth.pruned_weights(s, 500 + k), synth.bias_vector(s, 520 + k)
        x = synth.activations(s, 540 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0, 2, 2, 1, 1, s.group)
        for relu in (False, True):
            want = oracle.conv_forward(g, x, w, b, relu=relu, gate=False, threads=4)
            for kernel in (pkg.KERNEL_AUTO, pkg.KERNEL_JIT, pkg.KERNEL_TILED):
                for tb in (0, 256):
                    plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=kernel, tiling_batch=tb)
                    plan.weight_align(w)
                    if kernel == pkg.KERNEL_AUTO and tb == 0 and plan.stat("small_launch_rule") == 2:
                        # (a launch this small: KERNEL_AUTO's rule may prefer the generic kernel)
                        assert "generic" in plan.kernel_name, (s.name, plan.kernel_name)
                    else:
                        assert _fast(plan.kernel_name), (s.name, plan.kernel_name)
                    top = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev) if b is not None else None)
                    torch.cuda.synchronize()
                    err = rel_err(top.cpu().numpy(), want)
                    assert err <= TOL, "%s via %s (tiling_batch %d, relu %d): %g" % (s.name, plan.kernel_name, tb, relu, err)
                    plan.close()
    # an odd input width: no strided view (rows would not start on 16-byte boundaries) -- the layer still computes
    # (dense MFMA / generic kernel)
    s = synth.shape("w55", 2, 16, 10, 55, 8, 1, stride=2, bias=True, sparsity=0.5)
    w, b, x = synth.pruned_weights(s, 7), synth.bias_vector(s, 8), synth.activations(s, 9)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    plan.weight_align(w)
    assert not _fast(plan.kernel_name), plan.kernel_name
    g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0, 2, 2, 1, 1, 1)
    top = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev))
    assert rel_err(top.cpu().numpy(), oracle.conv_forward(g, x, w, b, gate=False)) <= TOL
    plan.close()


def test_plan_churn_gives_its_device_memory_back(pkg, synth, torch_cuda):
    """Plans created, aligned (generated code: a code object loaded per layer), run once and destroyed, round after
    round: what a plan takes on the device -- CSR, unit tables, the loaded code object -- goes back when it is
    destroyed (the reference frees its layer-private buffers in the destructor, base_conv_layer.cpp:16-42)."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    shapes = synth.resnet50_3x3(N=4) + synth.googlenet_1x1(N=4)[:4] + synth.alexnet(N=4)[:1]
    free = []
    for it in range(5):
        for s in shapes:
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_AUTO)
            plan.weight_align(synth.pruned_weights(s, 10 + it))
            x = torch.rand((s.N, s.C, s.H, s.W), device=dev)
            plan.forward(x, None)
            del plan
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    # (the first round warms torch's caching allocator and the HIP runtime; after it nothing may accumulate)
    assert max(free[1:]) - min(free[1:]) <= 8 << 20, free


def test_code_address_reuse_runs_the_new_code(pkg, oracle, synth, torch_cuda):
    """Generated code lives in executable device memory the library fills itself (csrc/code_memory.h): a destroyed plan's
    address range comes back for the next plan's code, and nothing but the kernel's own invalidation stands between that
    plan and the previous tenant's instructions / weight lines in the CUs' caches (the first build of this path returned
    NaN on exactly this pattern).  Same shape -- same code size, so the allocator hands the same range back --, other
    weights every round, both code loaders, small and chip-filling batches: every round against the oracle."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    for s in (synth.shape("r14", 40, 40, 14, 14, 96, 1, sparsity=0.95, group=2),
              synth.shape("r7k3", 24, 64, 7, 7, 64, 3, pad=1, sparsity=0.9, bias=False)):
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
        x = synth.activations(s, 31)
        xd = torch.from_numpy(x).to(dev)
        for it in range(6):
            w, b = synth.pruned_weights(s, 200 + it), synth.bias_vector(s, 300 + it)
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256, code_loader=it % 3 == 2)
            plan.weight_align(w)
            assert plan.stat("code_direct") == (0 if it % 3 == 2 else 1)
            bd = torch.from_numpy(b).to(dev) if b is not None else None
            want = oracle.conv_forward(g, x, w, b, gate=False, threads=4)
            for n in (s.N, 3):
                got = plan.forward(xd[:n], bd).cpu().numpy()
                assert rel_err(got, want[:n]) <= 1e-4, (s.name, it, n)
            plan.close()


def test_plans_of_two_host_threads_on_two_streams(pkg, oracle, synth, torch_cuda):
    """a layer instance belongs to one host thread; several threads, each with its own plans and its own
    stream, share the device.  Two threads WeightAlign (code generation, the process-wide code object template, module
    loads) and run their layers at the same time, repeatedly; every output is checked against the oracle."""
    import threading
    torch = torch_cuda
    dev = torch.device("cuda:0")
    sets = [synth.resnet50_3x3(N=3) + synth.alexnet(N=3)[:2], [synth.googlenet_1x1(N=3)[i] for i in (0, 5, 9, 25, 33)] + synth.lenet_conv2(N=3)]
    jobs = []
    for t, shapes in enumerate(sets):
        for k, s in enumerate(shapes):
            w, b, x = synth.pruned_weights(s, 500 + 10 * t + k), synth.bias_vector(s, 600 + k), synth.activations(s, 700 + k)
            g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
            jobs.append((t, s, w, b, x, oracle.conv_forward(g, x, w, b, gate=False, threads=4)))
    errors = []
    start = threading.Barrier(2)

    def worker(t):
        try:
            stream = torch.cuda.Stream(device=dev)
            start.wait()
            with torch.cuda.stream(stream):
                for rep in range(3):
                    for (tt, s, w, b, x, want) in jobs:
                        if tt != t:
                            continue
                        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_AUTO)
                        plan.weight_align(w)
                        xd = torch.from_numpy(x).to(dev, non_blocking=False)
                        bd = torch.from_numpy(b).to(dev) if b is not None else None
                        got = plan.forward(xd, bd)
                        stream.synchronize()
                        e = rel_err(got.cpu().numpy(), want)
                        if e > TOL:
                            errors.append("%s (thread %d, round %d): %g" % (s.name, t, rep, e))
        except Exception as exc:      # noqa: BLE001 -- reported to the main thread
            errors.append("thread %d: %r" % (t, exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors


def test_kernel_auto_small_launch_rule(pkg, oracle, synth, torch_cuda):
    """Below 64 MFLOP per launch, on a pointwise layer whose tiles fit one round of workgroups, KERNEL_AUTO's RULE
    (escoin_capi.hip; fitted to profiles/r05_small_launch_fit.md) may keep the generic kernel: one image of
    inception_4e/1x1 takes 8.7 us there and 18.8 us as a chain of nine blocks of generated code -- the reference's
    SCONV mode calls the layer image by image, conv_layer.cu:16-26.  A rule, not a measurement: the same options
    and weights give the same kernel every time.  The config batch is never considered; a plan whose tiling is
    asked for another batch is not either; 3x3 layers keep generated code; results are right whichever kernel runs."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    s = synth.googlenet_1x1(N=1)[25]
    w, b, x = synth.pruned_weights(s, 1), synth.bias_vector(s, 2), synth.activations(s, 3)
    g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0)
    want = oracle.conv_forward(g, x, w, b, gate=False)
    names = set()
    for _ in range(5):
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        plan.weight_align(w)
        assert plan.stat("small_launch_rule") == 2 and "generic" in plan.kernel_name, (plan.stat("small_launch_rule"), plan.kernel_name)
        assert plan.stat("kernel_choice") == pkg.KERNEL_GENERIC
        names.add(plan.kernel_name)
        got = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
        assert np.array_equal(got, want)          # (the generic kernel is bit-exact to the oracle)
        # the aligned form carries no code for such a plan, and a receiver resolves to the same kernel
        blob = plan.export_aligned()
        other = pkg.Plan(pkg.ConvDesc.from_shape(s))
        other.import_aligned(blob)
        assert other.kernel_name == plan.kernel_name and other.stat("small_launch_rule") == 2
        other.close()
        plan.close()
    assert len(names) == 1
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), tiling_batch=256)
    plan.weight_align(w)
    assert plan.stat("small_launch_rule") == 0 and _fast(plan.kernel_name)
    assert rel_err(plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy(), want) <= TOL
    plan.close()
    # a 3x3 layer at one image: never considered, generated code stays (12 us against 44)
    s3 = synth.resnet50_3x3(N=1)[2]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s3))
    plan.weight_align(synth.pruned_weights(s3, 4))
    assert plan.stat("small_launch_rule") == 0 and _fast(plan.kernel_name), (plan.stat("small_launch_rule"), plan.kernel_name)
    plan.close()
    # the config batch: far above the threshold
    s256 = synth.googlenet_1x1(N=256)[25]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s256))
    plan.weight_align(synth.pruned_weights(s256, 1))
    assert plan.stat("small_launch_rule") == 0 and _fast(plan.kernel_name)
    plan.close()


def test_kernel_auto_is_the_same_in_fresh_processes(pkg, synth):
    """The same plan options + weights give the same kernel_choice in 20 fresh processes (round 4's KERNEL_AUTO timed
    two kernels at WeightAlign and a layer's bits depended on the box's noise): five small pointwise launches around
    the rule's boundary, resolved by 20 children, one answer each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package(); synth = pkg.synth
out = []
for idx, n in ((0, 4), (5, 4), (5, 8), (9, 8), (25, 8), (25, 16), (33, 2)):
    s = synth.googlenet_1x1(N=n)[idx]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    plan.weight_align(synth.pruned_weights(s, 1))
    out.append("%%d:%%d" %% (plan.stat("kernel_choice"), plan.stat("small_launch_rule")))
    plan.close()
print("CHOICES " + " ".join(out))
""" % root
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for _ in range(4)]
    seen = set()
    done = 0
    while done < 20:
        for i, pr in enumerate(procs):
            out = pr.communicate(timeout=600)[0].decode()
            assert pr.returncode == 0, out
            seen.add([l for l in out.splitlines() if l.startswith("CHOICES")][0])
            done += 1
            procs[i] = None
        procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for _ in range(min(4, 20 - done))]
    assert len(seen) == 1, seen


def test_blobs_that_start_off_a_16_byte_boundary(pkg, oracle, synth, torch_cuda):
    """A caller may hand in any float-aligned pointer (the reference's SCONV mode passes `bottom_data + n * bottom_dim_`,
    conv_layer.cu:16-26; a 13 x 13 x 3 image is 2028 bytes): bottom and top one, two and three floats past a 16-byte
    boundary, on every kernel family, with widths that are and are not whole quads."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    cases = [synth.shape("k3", 3, 16, 14, 14, 24, 3, pad=1, sparsity=0.85), synth.shape("k1", 3, 40, 14, 14, 32, 1, sparsity=0.9),
             synth.shape("k3w13", 2, 8, 13, 13, 16, 3, pad=1, sparsity=0.8), synth.shape("k5", 2, 8, 27, 27, 16, 5, pad=2, sparsity=0.8)]
    for k, s in enumerate(cases):
        w, b, x = synth.pruned_weights(s, 40 + k), synth.bias_vector(s, 50 + k), synth.activations(s, 60 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        for kernel in _kernels(pkg, pkg.ConvDesc.from_shape(s)) + [pkg.KERNEL_DENSE]:
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel)
            plan.weight_align(w)
            bd = torch.from_numpy(b).to(dev)
            for off in (1, 2, 3):
                xin = torch.zeros(x.size + 8, device=dev)
                xin[off:off + x.size] = torch.from_numpy(x).to(dev).reshape(-1)
                out = torch.full((want.size + 8,), 5.0, device=dev)
                plan.forward_ptr(xin.data_ptr() + 4 * off, bd.data_ptr(), out.data_ptr() + 4 * off, s.N,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                o = out.cpu().numpy()
                got = o[off:off + want.size].reshape(want.shape)
                assert rel_err(got, want) <= TOL, "%s via %s, %d floats off: %g" % (s.name, plan.kernel_name, off, rel_err(got, want))
                assert (o[:off] == 5.0).all() and (o[off + want.size:] == 5.0).all(), (s.name, plan.kernel_name, off)
            plan.close()


@pytest.mark.parametrize("dist", ["channel", "zero_inputs", "filters_tail"])
def test_skewed_sparsity_distributions(pkg, oracle, synth, torch_cuda, dist):
    """A pruned model's weights are not uniformly sparse (the reference's nets are SkimCaffe-pruned, run.sh:14): per-output-
    channel densities from U(0, 2d), a fifth of the input channels all zero, a tenth of the filters all zero with a heavy
    tail of rows at 4d (synth.pruned_weights(..., dist); the same total count as the uniform case).  Every BASELINE 3x3 /
    5x5 shape and two pointwise ones, tiled as for their config batch, on generated code and on the stream kernel, against
    the oracle; the channel deal must stay a permutation (every output channel written: the comparison covers them all) and
    generated code must not fall off its size limit."""
    torch = torch_cuda
    shapes = list(synth.resnet50_3x3(N=3)) + list(synth.alexnet(N=3)) + [synth.googlenet_1x1(N=3)[5], synth.googlenet_1x1(N=3)[25]]
    for k, s in enumerate(shapes):
        w = synth.pruned_weights(s, 700 + k, dist)
        assert int((w != 0).sum()) == synth.nnz_of(s)
        b, x = synth.bias_vector(s, 720 + k), synth.activations(s, 740 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False, threads=4)
        tb = 256 if s.name.startswith(("res", "incep")) else 128
        for kernel in (pkg.KERNEL_JIT, pkg.KERNEL_TILED):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel, tiling_batch=tb)
            plan.weight_align(w)
            if kernel == pkg.KERNEL_JIT:
                assert "jit" in plan.kernel_name and plan.stat("code_bytes") > 0, (s.name, plan.kernel_name)
                assert plan.stat("deal_slowest_over_mean_x1000") >= 1000
            dev = torch.device("cuda:0")
            got = plan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev) if b is not None else None).cpu().numpy()
            plan.close()
            assert rel_err(got, want) <= TOL, "%s %s via kernel %d: %g" % (s.name, dist, kernel, rel_err(got, want))


@pytest.mark.parametrize("n", [257, 293])
def test_batches_off_the_tilings_grid(pkg, oracle, synth, torch_cuda, n):
    """A drop-in sees arbitrary batches (the reference calls the layer image by image in SCONV mode, conv_layer.cu:19-26;
    shard.py allows uneven shards): 257 and 293 images of res4 / res5 -- one more tile than the 256 CUs' round, a last tile
    that is not full -- through KERNEL_AUTO; every image against the generic kernel (bit-exact to the reference's order),
    the last three against the oracle."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    for s in (synth.resnet50_3x3(N=n)[2], synth.resnet50_3x3(N=n)[3]):
        w = synth.pruned_weights(s, 31)
        x = torch.rand((n, s.C, s.H, s.W), device=dev) * 2 - 1
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        plan.weight_align(w)
        assert _fast(plan.kernel_name), plan.kernel_name
        got = plan.forward(x, None)
        ref_plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_GENERIC)
        ref_plan.weight_align(w)
        ref = ref_plan.forward(x, None)
        torch.cuda.synchronize()
        scale = max(1e-6, float(ref.abs().max()))
        assert float((got - ref).abs().max()) / scale <= TOL, (s.name, n, plan.tiling_info)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
        want = oracle.conv_forward(g, x[n - 3:].cpu().numpy(), w, None, gate=False, threads=3)
        assert rel_err(got[n - 3:].cpu().numpy(), want) <= TOL, (s.name, n)
        plan.close(); ref_plan.close()


def test_dense_strided_pointwise_through_registers(pkg, oracle, synth, torch_cuda):
    """Dense (fp32 MFMA) kernel, 1x1 convolutions with stride > 1 and no padding (ResNet-50's res{3,4,5}a_branch1 /
    branch2a; the reference: forward_gpu_gemm + im2col, base_conv_layer.cpp:713-746).  The product gathers their B tile 4
    bytes per LDS-DMA lane; the experiments flavour (ESCOIN_LIB=tools/ab/libescoin_exp.so ESCOIN_DENSE_S2=1) stages it
    through registers instead -- one aligned 16-byte load per lane and k-row where the output width is even and the stride
    2, two 4-byte loads otherwise: built for VERDICT r4 item 4, measured 0-9 % slower, not shipped (profiles/r05_dense.md).
    Either way the same cases must hold: even and odd output widths, an odd input
    width, stride 3, conv groups, a channel count that is not whole k-steps (rows past K must read as zeros: NaNs are
    planted behind the blob), fewer output channels than a tile, a batch whose last tile is ragged, fused ReLU."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    cases = [synth.shape("s2_56", 3, 64, 56, 56, 128, 1, stride=2, bias=True, sparsity=0.2),          # pair path, 128-row tiles
             synth.shape("s2_28", 5, 96, 28, 28, 64, 1, stride=2, bias=True, sparsity=0.3),           # pair path, 64-row tiles
             synth.shape("s2_14", 7, 128, 14, 14, 160, 1, stride=2, bias=False, sparsity=0.1),        # odd OW = 7: 4-byte loads
             synth.shape("s2_odd_w", 4, 40, 13, 13, 48, 1, stride=2, bias=True, sparsity=0.5),        # odd input width, K = 40: a k tail
             synth.shape("s3", 3, 36, 16, 16, 40, 1, stride=3, bias=True, sparsity=0.4),              # stride 3
             synth.shape("s2_groups", 3, 80, 12, 12, 96, 1, stride=2, group=2, bias=True, sparsity=0.3),  # K = 40 per group
             synth.shape("s2_one", 1, 32, 6, 6, 8, 1, stride=2, bias=True, sparsity=0.0)]             # 9 outputs in all
    for k, s in enumerate(cases):
        w, b = synth.pruned_weights(s, 800 + k), synth.bias_vector(s, 820 + k)
        x = synth.activations(s, 840 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0, s.stride_h, s.stride_w, 1, 1, s.group)
        for relu in (False, True):
            want = oracle.conv_forward(g, x, w, b, relu=relu, gate=False, threads=4)
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), kernel=pkg.KERNEL_DENSE)
            plan.weight_align(w)
            assert "dense_mfma" in plan.kernel_name
            # the bottom blob as a window of a NaN-filled allocation: nothing outside it may reach a product
            per = x.size
            big = torch.full((3 * per,), float("nan"), device=dev)
            big[per:2 * per] = torch.from_numpy(x).to(dev).reshape(-1)
            bottom = big[per:2 * per].view(s.N, s.C, s.H, s.W)
            got = plan.forward(bottom, torch.from_numpy(b).to(dev) if b is not None else None).cpu().numpy()
            plan.close()
            assert np.isfinite(got).all(), s.name
            assert rel_err(got, want) <= TOL, "%s relu=%d: %g" % (s.name, relu, rel_err(got, want))


def test_half_workgroups_on_hbm_bound_pointwise_layers(pkg, oracle, synth, torch_cuda):
    """Two 4-wave workgroups per CU (sconv_tiled.hip, half-workgroup rule): what KERNEL_AUTO tiles GoogLeNet's 28 x 28 1x1 layers
    with up to 96 output channels as at batch 256.  The tiling of batch 256 on small inputs (tiling_batch), output channel
    counts that do and do not fill the four waves, bias, fused ReLU, a partial batch, and the cases the rule must NOT take
    (128 channels; 14 x 14: one tile per workgroup; 56 x 56: seven)."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    taken = 0
    for k, (C, HW, M, expect) in enumerate([(192, 28, 64, True), (192, 28, 96, True), (192, 28, 16, True), (192, 28, 20, True),
                                            (256, 28, 32, True), (256, 28, 90, True), (256, 28, 128, False), (480, 14, 64, False),
                                            (64, 56, 64, False)]):
        s = synth.shape("half_wg", 5, C, HW, HW, M, 1, bias=True, sparsity=0.95)
        w, b, x = synth.pruned_weights(s, 900 + k), synth.bias_vector(s, 920 + k), synth.activations(s, 940 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, 1, 1, 0, 0)
        for relu in (False, True):
            want = oracle.conv_forward(g, x, w, b, relu=relu, gate=False, threads=4)
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s, fuse_relu=relu), tiling_batch=256)
            plan.weight_align(w)
            half = "oc_waves=4 pix_waves=1" in plan.tiling_info
            assert half == expect, (C, HW, M, plan.tiling_info)
            taken += half
            xd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
            got = plan.forward(xd, bd).cpu().numpy()
            assert rel_err(got, want) <= TOL, (C, HW, M, relu, rel_err(got, want))
            # a partial batch through the same plan
            top = torch.full((3, s.M, HW, HW), float("nan"), device=dev)
            plan.forward(xd[:3].contiguous(), bd, top)
            assert rel_err(top.cpu().numpy(), want[:3]) <= TOL
            plan.close()
    assert taken == 12
