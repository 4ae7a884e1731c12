"""Caffe::CPU mode of the product library (csrc/sconv_cpu.cpp, sconv_cpu_kernel.cpp) -- no GPU needed.

The reference's layer runs in Caffe::CPU mode (conv_layer.cpp:25-63 -> base_conv_layer.cpp:569-661 ->
math_functions.cpp:128-176) for float and double (conv_layer.cpp:102).  escoin_forward_cpu is the product's
own host implementation of that path; the oracle stays on the other side of the comparison (the library neither
links nor loads it: test_product_library_does_not_know_the_oracle).  Bar: BIT-EXACT -- same summation order as the
reference's loop nest, whatever the tile shape or the thread count."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import Golden, golden_params, naive_conv, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True, params=["avx2", "avx512"])
def cpu_flavour(request, pkg):
    """Every test of this file runs once per flavour of the host kernel this machine can run."""
    try:
        pkg.cpu_kernel_select(request.param)
    except pkg.EscoinError:
        pytest.skip("this CPU has no %s" % request.param)
    assert pkg.cpu_kernel_name().endswith(request.param)
    yield request.param
    pkg.cpu_kernel_select("auto")


def _plan_cpu(pkg, g, w, **kw):
    plan = pkg.Plan(g.desc(pkg, **kw))
    plan.weight_align_cpu(w)
    return plan


@pytest.mark.parametrize("path", golden_params())
def test_cpu_forward_float_is_bit_equal_to_the_goldens_and_the_oracle(pkg, oracle, path):
    g = Golden(path)
    plan = _plan_cpu(pkg, g, g.w)
    for threads in (1, 3, 8):
        got = plan.forward_cpu(g.x, g.bias, n_threads=threads)
        # the committed fixture (made by the reference's own kernel, tests/golden/make_golden.py) ...
        assert np.array_equal(got, g.top), "%s, %d threads: differs from the golden" % (g.name, threads)
    # ... and the oracle restatement on the same inputs
    want = oracle.conv_forward(g.geom(oracle), g.x, g.w, g.bias, gate=False)
    assert np.array_equal(got, want)
    # the CSR WeightAlign built on the host is the fixture's, stretched indices included
    rp, ci, va, ng = plan.get_csr(stretched=True)
    assert np.array_equal(ci, g.colidx_stretched) and np.array_equal(va, g.values)
    plan.close()


@pytest.mark.parametrize("path", golden_params())
def test_cpu_forward_double_is_bit_equal_to_the_f64_oracle(pkg, oracle, path):
    g = Golden(path)
    rng = np.random.RandomState(7)
    # genuinely double data: the float fixtures' inputs scaled by something fp32 cannot hold
    x = g.x.astype(np.float64) * (1.0 + 1e-9) + rng.uniform(-1e-9, 1e-9, g.x.shape)
    w = g.w.astype(np.float64) * (1.0 / 3.0)
    b = None if g.bias is None else g.bias.astype(np.float64) * 1.000000001
    plan = _plan_cpu(pkg, g, w)
    assert plan.stat("is_f64") == 1
    want = oracle.conv_forward_f64(g.geom(oracle), x, w, b)
    for threads in (1, 5):
        got = plan.forward_cpu(x, b, n_threads=threads)
        assert got.dtype == np.float64 and np.array_equal(got, want), g.name
    # independent check of the f64 oracle itself: a dense direct convolution in the caffe_conv() style
    s = g
    ref = naive_conv(x, w, b, s)
    assert rel_err(want, ref) <= 1e-12
    # a float forward on a double plan is a state error, not a silent conversion
    with pytest.raises(pkg.EscoinError):
        plan.forward_cpu(g.x, g.bias)
    plan.close()


def test_f64_and_f32_oracles_agree_where_both_are_exact(oracle):
    # integer-valued data: every product and partial sum is exact in fp32 and fp64 alike
    rng = np.random.RandomState(3)
    g = oracle.geom(6, 9, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1, 2)
    x = rng.randint(-4, 5, (3, 6, 9, 8)).astype(np.float64)
    w = rng.randint(-3, 4, (8, 3, 3, 3)).astype(np.float64) * (rng.uniform(size=(8, 3, 3, 3)) < 0.4)
    b = rng.randint(-2, 3, 8).astype(np.float64)
    a32 = oracle.conv_forward(g, x.astype(np.float32), w.astype(np.float32), b.astype(np.float32), gate=False)
    a64 = oracle.conv_forward_f64(g, x, w, b)
    assert np.array_equal(a32.astype(np.float64), a64)


@pytest.mark.parametrize("geom", [
    # (N, C, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, group)   -- the shapes the vector kernel has to get right:
    (3, 4, 7, 7, 6, 3, 3, 1, 1, 1, 1, 1, 1, 1),        # 7x7: 55 virtual pixels, rows shorter than a vector
    (2, 3, 56, 56, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1),      # many tiles, last one partial
    (2, 4, 13, 13, 4, 3, 3, 1, 1, 1, 1, 1, 1, 2),      # groups
    (2, 2, 9, 10, 3, 1, 1, 0, 0, 1, 1, 1, 1, 1),       # pointwise, unpadded: reads the caller's blob directly
    (2, 2, 12, 11, 3, 3, 3, 0, 0, 1, 1, 2, 2, 1),      # dilation 2, unpadded
    (2, 3, 10, 9, 4, 3, 5, 1, 2, 1, 1, 1, 1, 1),       # non-square kernel and pads
    (2, 3, 11, 10, 4, 3, 3, 1, 1, 2, 2, 1, 1, 1),      # stride 2: the any-stride path
    (2, 3, 11, 10, 4, 3, 3, 0, 2, 1, 2, 1, 1, 1),      # pad_h == 0 < pad_w (the geometry escoin_padded_len adds slack for), stride_w 2
    (1, 2, 6, 9, 2, 3, 3, 0, 1, 1, 1, 1, 1, 1),        # pad_h == 0 < pad_w, stride 1
    (5, 1, 1, 1, 2, 1, 1, 0, 0, 1, 1, 1, 1, 1),        # 1 x 1 images
    (2, 2, 5, 40, 2, 5, 5, 2, 2, 1, 1, 1, 1, 1),       # wide rows
    (4, 1, 12, 29, 3, 4, 1, 2, 3, 1, 1, 2, 1, 1),      # pad_w > (KW - 1) / 2: MORE outputs per row than the padded pitch (found by the seeded sweep below)
    (2, 3, 6, 5, 4, 3, 3, 3, 3, 1, 1, 1, 1, 1),        # the same with a 3x3 kernel and pad 3
])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cpu_forward_geometries_vs_oracle(pkg, oracle, geom, dtype):
    N, Cc, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp = geom
    rng = np.random.RandomState(hash(geom) & 0xffff)
    x = rng.uniform(-1, 1, (N, Cc, H, W)).astype(dtype)
    w = (rng.uniform(-1, 1, (M, Cc // grp, KH, KW)) * (rng.uniform(size=(M, Cc // grp, KH, KW)) < 0.35)).astype(dtype)
    w[0] = 0                                            # an empty CSR row
    b = rng.uniform(-0.1, 0.1, M).astype(dtype)
    g = oracle.geom(Cc, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp)
    fwd = oracle.conv_forward_f64 if dtype == np.float64 else oracle.conv_forward
    for relu in (False, True):
        want = fwd(g, x, w, b, relu=relu, gate=False)
        desc = pkg.ConvDesc(N, Cc, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp, 1, int(relu))
        plan = pkg.Plan(desc)
        plan.weight_align_cpu(w)
        for threads in (1, 4, 11):                      # 11 > N: output channels are split too
            got = plan.forward_cpu(x, b, n_threads=threads)
            assert np.array_equal(got, want), (geom, relu, threads)
        got = plan.forward_cpu(x, None, n_threads=2)    # bias = NULL is legal
        assert np.array_equal(got, fwd(g, x, w, None, relu=relu, gate=False))
        plan.close()


def test_cpu_forward_batch_larger_than_desc_and_repeated_calls(pkg, oracle, synth):
    s = synth.resnet50_3x3(N=2)[2]                      # res4: 14 x 14
    w = synth.pruned_weights(s, 5)
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    plan.weight_align_cpu(w)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
    x = synth.activations(synth.resnet50_3x3(N=9)[2], 6)
    want = oracle.conv_forward(g, x, w, None, gate=False)
    buf = np.empty_like(x)
    for rep in range(3):                                # the same host buffer holding other data each call
        buf[...] = np.roll(x, rep, axis=0)
        got = plan.forward_cpu(buf, None, n_threads=4)
        assert np.array_equal(got, np.roll(want, rep, axis=0)), rep
    plan.close()


def test_cpu_sconv_dropin_on_the_reference_layout(pkg, oracle):
    """escoin_cpu_sconv = caffe_cpu_sconv<Dtype> (math_functions.cpp:128-176): padded image of EXACTLY the
    reference's length, stretched CSR, one group of one image; both Dtypes; dilated branch included."""
    L = pkg.lib()
    rng = np.random.RandomState(11)
    for (Cc, H, W, M, K, p, s, d) in [(5, 9, 8, 7, 3, 1, 1, 1), (3, 12, 13, 4, 3, 2, 1, 2), (4, 10, 10, 5, 5, 2, 2, 1),
                                      (6, 7, 7, 8, 3, 1, 1, 1)]:
        g = oracle.geom(Cc, H, W, M, K, K, p, p, s, s, d, d, 1)
        oh, ow = oracle.out_hw(g)
        wd = (rng.uniform(-1, 1, (M, Cc * K * K)) * (rng.uniform(size=(M, Cc * K * K)) < 0.3)).astype(np.float32)
        rp, ci, va = oracle.dense2csr(wd)
        cs = oracle.stretch(rp, ci, K, K, H, W, p, p)
        img = rng.uniform(-1, 1, (Cc, H, W)).astype(np.float32)
        for dt, suffix, sconv, pad in ((np.float32, "", oracle.sconv, oracle.pad_input),
                                       (np.float64, "_f64", oracle.sconv_f64, oracle.pad_input_f64)):
            padded = pad(g, img.astype(dt))
            plen = (Cc * (H + p) * (W + p) + p * (W + 2 * p))     # base_conv_layer.cpp:71, no slack
            exact = np.ascontiguousarray(padded[:plen])
            want = sconv(g, padded, Cc, rp, cs, va.astype(dt), M)
            out = np.full((M, oh, ow), 7, dt)
            fn = getattr(L, "escoin_cpu_sconv" + suffix)
            v = va.astype(dt)
            rc = fn(exact.ctypes.data, Cc, H, W, p, p, s, s, d, d, rp.ctypes.data, cs.ctypes.data, v.ctypes.data, K, K,
                    None, out.ctypes.data, M, plen)
            assert rc == 0, L.escoin_last_error()
            assert np.array_equal(out, want), (Cc, H, W, K, p, s, d, suffix)
            # a buffer shorter than what the last output reads is refused (the reference asserts, :168)
            rc = fn(exact.ctypes.data, Cc, H, W, p, p, s, s, d, d, rp.ctypes.data, cs.ctypes.data, v.ctypes.data, K, K,
                    None, out.ctypes.data, M, max(1, int(cs.max()) if len(cs) else 1))
            assert rc == -1


def test_cpu_dense2csr_dropin(pkg, oracle):
    L = pkg.lib()
    rng = np.random.RandomState(2)
    A = (rng.uniform(-1, 1, (9, 37)) * (rng.uniform(size=(9, 37)) < 0.2)).astype(np.float32)
    A[4] = 0
    rp, ci, va = oracle.dense2csr(A)
    for dt, suffix in ((np.float32, ""), (np.float64, "_f64")):
        a = A.astype(dt)
        vals, cols, ptr = np.zeros(A.size, dt), np.zeros(A.size, np.int32), np.zeros(10, np.int32)
        rc = getattr(L, "escoin_cpu_sparse_dense2csr" + suffix)(9, 37, a.ctypes.data, vals.ctypes.data,
                                                                cols.ctypes.data, ptr.ctypes.data)
        assert rc == 0
        assert np.array_equal(ptr, rp) and np.array_equal(cols[:ptr[-1]], ci) and np.array_equal(vals[:ptr[-1]], va.astype(dt))


def test_forward_cpu_state_and_argument_errors(pkg, synth):
    s = synth.lenet_conv2(N=1)[0]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    x = synth.activations(s, 1)
    with pytest.raises(pkg.EscoinError) as e:          # before WeightAlign
        plan.forward_cpu(x)
    assert "before weight_align" in str(e.value)
    plan.weight_align_cpu(synth.pruned_weights(s, 2))
    assert plan.stat("host_aligned") == 1 and plan.stat("is_f64") == 0
    with pytest.raises(pkg.EscoinError):               # double data on a float plan
        plan.forward_cpu(x.astype(np.float64))
    assert pkg.lib().escoin_forward_cpu(plan._h, None, None, None, 1, 1) == -1
    assert plan.forward_cpu(x[:0]).shape[0] == 0       # empty batch
    # re-align with double weights: the plan changes type
    plan.weight_align_cpu(synth.pruned_weights(s, 2).astype(np.float64))
    assert plan.stat("is_f64") == 1
    plan.forward_cpu(x.astype(np.float64))
    plan.close()


def test_padded_len_covers_the_pad_h_zero_case(pkg):
    # base_conv_layer.cpp:71 is pad_w floats short when pad_h == 0 < pad_w: the library's helper now
    # returns what its own kernels read
    d = pkg.ConvDesc(1, 3, 6, 9, 2, 3, 3, 0, 2, 1, 1, 1, 1, 1, 0, 0)
    assert pkg.lib().escoin_padded_len(C.byref(d)) == 3 * 6 * (9 + 2) + 2
    d = pkg.ConvDesc(1, 3, 6, 9, 2, 3, 3, 1, 2, 1, 1, 1, 1, 1, 0, 0)
    assert pkg.lib().escoin_padded_len(C.byref(d)) == 3 * 7 * 11 + 1 * (9 + 4)


def test_product_library_does_not_know_the_oracle(pkg):
    """The CPU mode is product code: nothing in libescoin_hip.so or its sources names, links or loads oracle/."""
    out = subprocess.run(["ldd", pkg.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "oracle" not in out
    strings = subprocess.run(["strings", "-n", "5", pkg.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "liboracle" not in strings and "oracle_" not in strings and "dlopen" not in strings.split("GLIBC")[0]
    csrc = os.path.join(ROOT, "caffe-escoin_amd", "csrc")
    for fn in ("sconv_cpu.cpp", "sconv_cpu_kernel.cpp", "sconv_cpu.h"):
        text = open(os.path.join(csrc, fn)).read()
        for line in text.splitlines():
            if "#include" in line or "dlopen" in line:
                assert "oracle" not in line, (fn, line)


def test_channel_blocking_keeps_every_bit(pkg, oracle, synth):
    """Channel blocking (sconv_cpu.h GroupJob::blk_ptr; the reference's register-blocked kernel parks partial sums the same
    way, sconv.hpp:57-589): the full-size layers pick a block size by themselves, and any forced block size on small and
    awkward geometries -- one channel per block, blocks that leave rows empty, groups, dilation, masked tails, fewer
    images than threads -- gives the bits of the unblocked run and of the oracle."""
    # (1) the BASELINE shapes block by themselves where a tile's window exceeds L1
    for s, expect_blocked in ((synth.resnet50_3x3(N=2)[0], True), (synth.resnet50_3x3(N=2)[2], True), (synth.lenet_conv2(N=2)[0], False)):
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        w, b = synth.pruned_weights(s, 3), synth.bias_vector(s, 4)
        plan.weight_align_cpu(w)
        x = synth.activations(s, 5, 0, 2)
        got = plan.forward_cpu(x, b, n_threads=2)
        assert (plan.stat("cpu_channel_block") > 0) == expect_blocked, (s.name, plan.stat("cpu_channel_block"))
        plan.set_option("cpu_channel_block", 10 ** 6)            # more channels than the layer has: unblocked
        assert np.array_equal(plan.forward_cpu(x, b, n_threads=2), got) and plan.stat("cpu_channel_block") == 0
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
        assert np.array_equal(got, oracle.conv_forward(g, x, w, b, gate=False, threads=4)), s.name
        plan.close()
    # (2) forced block sizes on small geometries
    rng = np.random.RandomState(11)
    for (C_, H, W, M, KH, KW, ph, pw, dh, dw, grp) in ((12, 9, 11, 7, 3, 3, 1, 1, 1, 1, 1), (16, 6, 5, 10, 1, 1, 0, 0, 1, 1, 2),
                                                       (9, 13, 17, 6, 3, 2, 2, 3, 2, 1, 3), (20, 7, 7, 5, 5, 5, 2, 2, 1, 1, 1),
                                                       (6, 30, 33, 4, 3, 3, 0, 1, 1, 2, 1)):
        for dtype in (np.float32, np.float64):
            Cg = C_ // grp
            x = rng.uniform(-1, 1, (3, C_, H, W)).astype(dtype)
            w = (rng.uniform(-1, 1, (M * grp, Cg, KH, KW)) * (rng.uniform(size=(M * grp, Cg, KH, KW)) < 0.35)).astype(dtype)
            w[0] = 0                                                   # an empty row
            b = rng.uniform(-0.1, 0.1, M * grp).astype(dtype)
            g = oracle.geom(C_, H, W, M * grp, KH, KW, ph, pw, 1, 1, dh, dw, grp)
            fwd = oracle.conv_forward_f64 if dtype == np.float64 else oracle.conv_forward
            want = fwd(g, x, w, b) if dtype == np.float64 else fwd(g, x, w, b, gate=False)
            desc = pkg.ConvDesc(N=3, C=C_, H=H, W=W, M=M * grp, KH=KH, KW=KW, pad_h=ph, pad_w=pw, stride_h=1, stride_w=1,
                                dil_h=dh, dil_w=dw, group=grp, has_bias=1, fuse_relu=0)
            plan = pkg.Plan(desc)
            plan.weight_align_cpu(w)
            for cb in (0, 1, 2, 3, Cg - 1, Cg):
                if cb < 0:
                    continue
                plan.set_option("cpu_channel_block", cb)
                for threads in (1, 5):
                    got = plan.forward_cpu(x, b, n_threads=threads)
                    assert np.array_equal(got, want), (C_, H, W, M, KH, KW, grp, dtype.__name__, cb, threads)
                if 0 < cb < Cg:
                    assert plan.stat("cpu_channel_block") == cb
            plan.close()


def test_several_small_images_per_job_keep_every_bit(pkg, oracle, synth):
    """Small images travel two or three to a job (sconv_cpu.h GroupJob::n_img: one broadcast weight feeds every image's
    accumulators): forced 1 / 2 / 3 images per job on 7 x 7, 4 x 4 and 5 x 9 geometries -- padded and not, groups, dilation,
    with forced channel blocks, batches that are not a multiple of the group, fewer images than threads -- give the bits of
    the oracle; and the full-size res5 shape takes three by itself."""
    s = synth.resnet50_3x3(N=12)[3]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    w = synth.pruned_weights(s, 3)
    plan.weight_align_cpu(w)
    x = synth.activations(s, 5, 0, 12)
    got = plan.forward_cpu(x, None, n_threads=2)
    # (seven 8-lane vectors per 7 x 7 image leave no room for a second image in the AVX2 flavour's twelve accumulators)
    assert plan.stat("cpu_images_per_job") >= (2 if pkg.cpu_kernel_name().endswith("avx512") else 1) and plan.stat("cpu_channel_block") > 0
    plan.set_option("cpu_images_per_job", 1)
    assert np.array_equal(plan.forward_cpu(x, None, n_threads=2), got) and plan.stat("cpu_images_per_job") == 1
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
    assert np.array_equal(got, oracle.conv_forward(g, x, w, None, gate=False, threads=4))
    plan.close()
    rng = np.random.RandomState(13)
    for (C_, H, W, M, KH, KW, ph, pw, dh, dw, grp) in ((10, 7, 7, 6, 3, 3, 1, 1, 1, 1, 1), (12, 4, 4, 9, 1, 1, 0, 0, 1, 1, 3),
                                                       (8, 5, 9, 5, 3, 3, 2, 0, 1, 2, 2), (16, 7, 7, 4, 1, 1, 0, 0, 1, 1, 1)):
        for dtype in (np.float32, np.float64):
            Cg = C_ // grp
            N = 11
            x = rng.uniform(-1, 1, (N, C_, H, W)).astype(dtype)
            w = (rng.uniform(-1, 1, (M * grp, Cg, KH, KW)) * (rng.uniform(size=(M * grp, Cg, KH, KW)) < 0.4)).astype(dtype)
            b = rng.uniform(-0.1, 0.1, M * grp).astype(dtype)
            g = oracle.geom(C_, H, W, M * grp, KH, KW, ph, pw, 1, 1, dh, dw, grp)
            want = oracle.conv_forward_f64(g, x, w, b) if dtype == np.float64 else oracle.conv_forward(g, x, w, b, gate=False)
            desc = pkg.ConvDesc(N=N, C=C_, H=H, W=W, M=M * grp, KH=KH, KW=KW, pad_h=ph, pad_w=pw, stride_h=1, stride_w=1,
                                dil_h=dh, dil_w=dw, group=grp, has_bias=1, fuse_relu=0)
            plan = pkg.Plan(desc)
            plan.weight_align_cpu(w)
            seen = set()
            for ni in (0, 1, 2, 3):
                for cb in (0, 2):
                    plan.set_option("cpu_images_per_job", ni)
                    plan.set_option("cpu_channel_block", cb)
                    for threads, n in ((1, N), (2, N), (4, 5), (16, 3)):
                        got = plan.forward_cpu(x[:n], b, n_threads=threads)
                        assert np.array_equal(got, want[:n]), (C_, H, W, M, KH, KW, grp, dtype.__name__, ni, cb, threads, n)
                        seen.add(plan.stat("cpu_images_per_job"))
            # (the multi-image tiles really ran wherever two images fit the flavour's accumulators)
            wide = pkg.cpu_kernel_name().endswith("avx512")
            lanes = (16 if wide else 8) // (2 if dtype == np.float64 else 1)
            OH, OW = H + 2 * ph - (dh * (KH - 1) + 1) + 1, W + 2 * pw - (dw * (KW - 1) + 1) + 1
            vecs = -(-((OH - 1) * (W + pw) + OW) // lanes)
            assert (max(seen) >= 2) == (2 * vecs <= (14 if wide else 12)), (seen, vecs, lanes)
            plan.close()


def test_pool_threads_are_placed_not_pinned_and_out_is_written_in_place(pkg, oracle, synth):
    """The pool's workers are moved to a core of their own when they start and get the mask they inherited back at once
    (sconv_cpu.cpp, place_on_own_core): after a team call every thread of the process still has the caller's mask.  And
    `out=` writes the caller's top blob (a Caffe top is allocated at Reshape, not per Forward)."""
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity on this platform")
    s = synth.googlenet_1x1(N=8)[3]
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
    w = synth.pruned_weights(s, 5)
    plan.weight_align_cpu(w)
    x = synth.activations(s, 6, 0, 8)
    mine = os.sched_getaffinity(0)
    top = np.full((8, s.M) + tuple(plan.out_hw), np.nan, np.float32)
    got = plan.forward_cpu(x, None, n_threads=6, out=top)
    assert got is top and not np.isnan(top).any()
    assert np.array_equal(top, plan.forward_cpu(x, None, n_threads=1))
    for tid in os.listdir("/proc/self/task"):
        try:
            assert os.sched_getaffinity(int(tid)) == mine, "thread %s was left pinned" % tid
        except ProcessLookupError:
            pass
    with pytest.raises(AssertionError):
        plan.forward_cpu(x, None, out=top[:, :, ::-1])           # not C-contiguous
    plan.close()


@pytest.mark.parametrize("seed", [20261004, 7])
def test_cpu_forward_seeded_random_geometries(pkg, oracle, seed):
    """A seeded slice of random geometries (kernel 1..5 per axis, strides 1..3, pads 0..3, dilation 1..2, groups, odd
    widths, batches below and above the thread count) through the product's CPU mode against the oracle: bit-equal,
    float and double, with and without bias / ReLU."""
    rng = np.random.RandomState(seed)
    done = 0
    while done < 60:
        grp = int(rng.choice([1, 1, 2, 3]))
        Cg, Mg = int(rng.randint(1, 7)), int(rng.randint(1, 9))
        KH, KW = int(rng.randint(1, 6)), int(rng.randint(1, 6))
        sh, sw = int(rng.choice([1, 1, 1, 2, 3])), int(rng.choice([1, 1, 1, 2, 3]))
        dh, dw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2]))
        ph, pw = int(rng.randint(0, 4)), int(rng.randint(0, 4))
        H, W = int(rng.randint(1, 24)), int(rng.randint(1, 40))
        if (H + 2 * ph - (dh * (KH - 1) + 1)) < 0 or (W + 2 * pw - (dw * (KW - 1) + 1)) < 0:
            continue
        N = int(rng.randint(1, 7))
        C_, M = Cg * grp, Mg * grp
        dtype = np.float64 if rng.rand() < 0.4 else np.float32
        dens = float(rng.choice([0.05, 0.3, 0.7, 1.0]))
        x = rng.uniform(-1, 1, (N, C_, H, W)).astype(dtype)
        w = (rng.uniform(-1, 1, (M, Cg, KH, KW)) * (rng.uniform(size=(M, Cg, KH, KW)) < dens)).astype(dtype)
        b = rng.uniform(-0.1, 0.1, M).astype(dtype) if rng.rand() < 0.7 else None
        relu = bool(rng.rand() < 0.3)
        g = oracle.geom(C_, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp)
        fwd = oracle.conv_forward_f64 if dtype == np.float64 else oracle.conv_forward
        want = fwd(g, x, w, b, relu=relu, gate=False)
        plan = pkg.Plan(pkg.ConvDesc(N, C_, H, W, M, KH, KW, ph, pw, sh, sw, dh, dw, grp, int(b is not None), int(relu)))
        plan.weight_align_cpu(w)
        got = plan.forward_cpu(x, b, n_threads=int(rng.choice([1, 2, 5])))
        assert np.array_equal(got, want), dict(N=N, C=C_, H=H, W=W, M=M, KH=KH, KW=KW, ph=ph, pw=pw, sh=sh, sw=sw, dh=dh, dw=dw,
                                                grp=grp, dtype=str(dtype), relu=relu, bias=b is not None)
        plan.close()
        done += 1
