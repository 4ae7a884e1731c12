"""Dtype = double on the device (conv_layer.cu:75, math_functions.cu:696-704,765-766) and CPU mode next to GPU mode
on one plan -- needs an MI355X.

Bar for double: <= 1e-12 relative against the fp64 oracle.  The device kernel keeps the
reference's summation order with fp64 fused multiply-adds, so the tests assert the stronger BIT-EXACT equality."""
import ctypes as C

import numpy as np
import pytest

from conftest import Golden, golden_params, naive_conv, rel_err

pytestmark = pytest.mark.gpu
TOL64 = 1e-12


@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert pkg.device_count() >= 1
    return torch


def _double_inputs(g):
    rng = np.random.RandomState(17)
    x = g.x.astype(np.float64) * (1.0 + 1e-9) + rng.uniform(-1e-9, 1e-9, g.x.shape)
    w = g.w.astype(np.float64) / 3.0
    b = None if g.bias is None else g.bias.astype(np.float64) * 1.000000001
    return x, w, b


@pytest.mark.parametrize("path", golden_params())
def test_double_forward_on_the_golden_geometries(pkg, oracle, torch_cuda, path):
    torch = torch_cuda
    dev = torch.device("cuda:0")
    g = Golden(path)
    x, w, b = _double_inputs(g)
    want = oracle.conv_forward_f64(g.geom(oracle), x, w, b)
    xd = torch.from_numpy(x).to(dev)
    bd = None if b is None else torch.from_numpy(b).to(dev)
    for relu in (False, True):
        for src in ("host", "device"):
            plan = pkg.Plan(g.desc(pkg, fuse_relu=relu))
            plan.weight_align(w if src == "host" else torch.from_numpy(w).to(dev))
            assert plan.stat("is_f64") == 1 and "f64" in plan.kernel_name
            assert plan.stat("kernel_choice") == pkg.KERNEL_GENERIC
            got = plan.forward(xd, bd).cpu().numpy()
            ref = np.maximum(want, 0) if relu else want
            assert rel_err(got, ref) <= TOL64
            assert np.array_equal(got, ref), (g.name, relu, src)
            # the same plan in Caffe::CPU mode: the host kernel gives the same bits as the device kernel
            assert np.array_equal(plan.forward_cpu(x, b, n_threads=3), got)
            # a float forward on a double plan is refused
            with pytest.raises(pkg.EscoinError):
                plan.forward(xd.float(), None if bd is None else bd.float())
            plan.close()


def test_double_plan_in_every_conv_mode_and_csr_round_trip(pkg, oracle, synth, torch_cuda):
    torch = torch_cuda
    dev = torch.device("cuda:0")
    s = synth.shape("d", 3, 8, 11, 10, 12, 3, pad=1, group=2, sparsity=0.7)
    w = synth.pruned_weights(s, 4).astype(np.float64) * 0.7
    x = synth.activations(s, 5).astype(np.float64) / 7.0
    b = synth.uniform(6, s.M, -0.1, 0.1).astype(np.float64)
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
    want = oracle.conv_forward_f64(g, x, w, b)
    xd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
    desc = pkg.ConvDesc.from_shape(s)
    desc.has_bias = 1
    plan = pkg.Plan(desc)
    plan.weight_align(w)
    for mode in (pkg.CONV_MODE_SCONV_PAR, pkg.CONV_MODE_SCONV, pkg.CONV_MODE_LOWERED_SPARSE, pkg.CONV_MODE_LOWERED_GEMM,
                 pkg.CONV_MODE_SCONV_PAR):
        plan.set_option("conv_mode", mode)          # (to / from LOWERED_GEMM re-uploads an aligned plan)
        assert np.array_equal(plan.forward(xd, bd).cpu().numpy(), want), mode
    rp, ci, va, ng = plan.get_csr()
    assert va.dtype == np.float64
    # broadcast receiver in double: set_csr_f64 without the dense blob
    other = pkg.Plan(desc)
    other.set_csr(rp, ci, va, ng)
    assert other.stat("is_f64") == 1
    assert np.array_equal(other.forward(xd, bd).cpu().numpy(), want)
    # the aligned form (generated fp32 code) is not defined for a double plan
    with pytest.raises(pkg.EscoinError):
        other.export_aligned()
    # re-aligning the same plan with float weights turns it back into a float plan with the fast kernels
    other.weight_align(w.astype(np.float32))
    assert other.stat("is_f64") == 0
    got = other.forward(xd.float(), bd.float()).cpu().numpy()
    assert rel_err(got, want) <= 1e-4
    plan.close()
    other.close()


def test_double_math_functions_level_dropins(pkg, oracle, synth, torch_cuda):
    """caffe_gpu_sparse_dense2csr<double> / copy_input_data<double> / caffe_gpu_sconv<double> /
    caffe_gpu_sparse_csrmm<double> on the reference's own layouts."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    L = pkg.lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    for s in (synth.shape("m", 3, 6, 9, 8, 5, 3, pad=1, sparsity=0.6),
              synth.shape("md", 2, 4, 9, 8, 5, 3, pad=2, dil=2, sparsity=0.5),
              synth.shape("ms", 2, 4, 9, 8, 5, 3, pad=1, stride=2, sparsity=0.5)):
        w = synth.pruned_weights(s, 1).astype(np.float64) / 3.0
        x = synth.activations(s, 2).astype(np.float64) * (1 + 1e-10)
        bias = synth.uniform(3, s.M, -0.1, 0.1).astype(np.float64)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w)
        kdim = s.C * s.KH * s.KW
        A = torch.from_numpy(w.reshape(s.M, kdim)).to(dev)
        vals = torch.zeros(s.M * kdim, device=dev, dtype=torch.float64)
        cols = torch.zeros(s.M * kdim, dtype=torch.int32, device=dev)
        rowp = torch.zeros(s.M + 1, dtype=torch.int32, device=dev)
        perrow = torch.zeros(s.M, dtype=torch.int32, device=dev)
        nnz = C.c_int()
        assert L.escoin_gpu_sparse_dense2csr_f64(s.M, kdim, P(A), P(perrow), P(vals), P(rowp), P(cols),
                                                 C.byref(nnz), None) == 0
        orp, oci, ova = oracle.dense2csr(w.reshape(s.M, kdim).astype(np.float32))     # (pattern only)
        assert nnz.value == len(oci) and np.array_equal(rowp.cpu().numpy(), orp)
        assert np.array_equal(cols.cpu().numpy()[:nnz.value], oci)
        assert np.array_equal(vals.cpu().numpy()[:nnz.value], w.reshape(-1)[w.reshape(-1) != 0])
        assert L.escoin_gpu_stretch(P(rowp), P(cols), s.M, s.H, s.W, s.pad_h, s.pad_w, s.KH, s.KW, None) == 0
        desc = pkg.ConvDesc.from_shape(s)
        plen = pkg.lib().escoin_padded_len(C.byref(desc))
        N = x.shape[0]
        ifmap = s.C * (s.H + s.pad_h) * (s.W + s.pad_w)
        padded = torch.zeros(N * ifmap + plen, device=dev, dtype=torch.float64)
        xd = torch.from_numpy(x).to(dev)
        for n in range(N):
            assert L.escoin_copy_input_data_f64(C.c_void_p(padded.data_ptr() + 8 * n * ifmap),
                                                C.c_void_p(xd.data_ptr() + 8 * n * s.C * s.H * s.W),
                                                s.C, s.H, s.W, s.pad_h, s.pad_w, None) == 0
        torch.cuda.synchronize()
        pn = padded.cpu().numpy()
        for n in range(N):
            assert np.array_equal(pn[n * ifmap:(n + 1) * ifmap], oracle.pad_input_f64(g, x[n])[:ifmap])
        oh, ow = oracle.out_hw(g)
        out = torch.zeros(N, s.M, oh, ow, device=dev, dtype=torch.float64)
        bd = torch.from_numpy(bias).to(dev)
        base = oracle.conv_forward_f64(g, x, w, None)
        for relu in (0, 1):
            assert L.escoin_gpu_sconv_f64(relu, N, P(padded), ifmap, P(rowp), P(cols), P(vals), P(bd), s.H, s.W,
                                          s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.KH, s.KW,
                                          P(out), s.M, 1, None) == 0
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            if relu:      # FUSE_RELU starts the sum at bias (math_functions.cu:215,421): another order, not bit-equal
                assert rel_err(got, np.maximum(base + bias[None, :, None, None], 0)) <= TOL64
            else:
                assert np.array_equal(got, base)
    # csrmm<double>: C = alpha A B + beta C
    rng = np.random.RandomState(5)
    M, K, N = 13, 29, 70
    A = rng.uniform(-1, 1, (M, K)) * (rng.uniform(size=(M, K)) < 0.3)
    B = rng.uniform(-1, 1, (K, N))
    C0 = rng.uniform(-1, 1, (M, N))
    rp = np.zeros(M + 1, np.int32)
    ci, va = [], []
    for i in range(M):
        nz = np.nonzero(A[i])[0]
        ci += list(nz)
        va += list(A[i][nz])
        rp[i + 1] = len(ci)
    t = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(dev)
    vd, cd, rd, Bd, Cd = t(va, np.float64), t(ci, np.int32), t(rp, np.int32), t(B, np.float64), t(C0, np.float64)
    assert L.escoin_gpu_sparse_csrmm_f64(M, N, K, len(ci), 1.5, P(vd), P(rd), P(cd), P(Bd), -0.25, P(Cd), None) == 0
    torch.cuda.synchronize()
    assert rel_err(Cd.cpu().numpy(), 1.5 * (A @ B) - 0.25 * C0) <= TOL64


def test_cpu_mode_and_gpu_mode_of_one_float_plan(pkg, oracle, synth, torch_cuda):
    """escoin_forward_cpu on a plan that escoin_weight_align prepared for the device: the host kernel is bit-equal to
    the oracle and to the device's order-preserving generic kernel; the fast device kernel is within 1e-4 of both."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    for s in (synth.resnet50_3x3(N=5)[1], synth.alexnet(N=3)[0], synth.googlenet_1x1(N=4)[1]):
        w, x, b = synth.pruned_weights(s, 21), synth.activations(s, 22), synth.bias_vector(s, 23)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False)
        xd = torch.from_numpy(x).to(dev)
        bd = None if b is None else torch.from_numpy(b).to(dev)
        gen = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_GENERIC)
        gen.weight_align(w)
        fast = pkg.Plan(pkg.ConvDesc.from_shape(s))
        fast.weight_align(w)
        cpu = fast.forward_cpu(x, b, n_threads=4)
        assert np.array_equal(cpu, want)
        assert np.array_equal(gen.forward(xd, bd).cpu().numpy(), cpu)
        assert rel_err(fast.forward(xd, bd).cpu().numpy(), cpu) <= 1e-4
        gen.close()
        fast.close()
