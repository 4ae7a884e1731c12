"""SURVEY 8 f2 / f3 on the GPU: the `caffe test` look-alike (tools/caffe_test.py) and the
.caffemodel / persisted-CSR hand-off feeding the HIP path (reference: tools/caffe.cpp:270-370,
Net::CopyTrainedLayersFrom -> WeightAlign, net.cpp:785-822)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "caffe_test.py")


def _tool(*argv):
    out = subprocess.run([sys.executable, TOOL] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0, text
    return text


def test_caffe_test_export_then_weights_from_file(tmp_path):
    """--export writes the pruned AlexNet as a .caffemodel; a second run takes every layer's
    weights and biases from that file (CopyTrainedLayersFrom), WeightAligns them and checks each
    layer against the oracle."""
    path = str(tmp_path / "alexnet_pruned.caffemodel")
    assert "4 convolution layers" in _tool("--model", "alexnet", "--export", path)
    assert os.path.getsize(path) > 1 << 20
    text = _tool("--model", "alexnet", "--weights", path, "--check", "--batch", "6", "--iterations", "1")
    assert "weights of 4 / 4 layers taken from the file" in text
    assert "oracle check: worst relative error" in text and "[cxh] Total CONV time" in text


def test_caffe_test_every_conv_mode(tmp_path):
    for mode in ("0", "1", "2", "3"):
        text = _tool("--model", "lenet", "--conv_mode", mode, "--check", "--batch", "5", "--iterations", "1")
        assert "oracle check: worst relative error" in text, mode


def test_caffe_test_resnet50_chain_with_dense_1x1():
    """All 53 convolutions of ResNet-50's bottleneck blocks chained (dense 1x1 on the MFMA kernel
    around the sparse 3x3), every layer checked against the oracle, per-type buckets printed."""
    text = _tool("--model", "resnet50_chain", "--batch", "4", "--iterations", "1", "--check")
    assert "over 52 conv layers" in text or "over 53 conv layers" in text, text
    assert "sparse 3x3" in text and "dense 1x1" in text and "conv / total" in text
    assert "dense_mfma" in text and ("tiled" in text or "jit" in text)


def test_persisted_csr_feeds_the_hip_path(tmp_path, pkg, oracle, synth):
    """save_aligned -> load_aligned -> escoin_plan_set_csr -> forward: a layer restored from the
    persisted CSR (no dense blob, no dense -> CSR) computes what the oracle computes."""
    import torch
    from caffe_escoin_amd import caffemodel as cm
    dev = torch.device("cuda:0")
    shapes = [synth.alexnet(N=3)[0], synth.resnet50_3x3(N=3)[2], synth.googlenet_1x1(N=3)[4]]
    entries, inputs = {}, {}
    for k, s in enumerate(shapes):
        w, b = synth.pruned_weights(s, 40 + k), synth.bias_vector(s, 50 + k)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        plan.weight_align(w)
        entries[s.name] = (pkg.ConvDesc.from_shape(s), plan.get_csr())
        inputs[s.name] = (s, w, b)
        plan.close()
    path = str(tmp_path / "aligned.npz")
    cm.save_aligned(path, entries)
    for name, (fields, csr) in cm.load_aligned(path).items():
        s, w, b = inputs[name]
        plan = pkg.Plan(pkg.ConvDesc(*fields))
        plan.set_csr(*csr)
        x = synth.activations(s, 60)
        got = plan.forward(torch.from_numpy(x).to(dev),
                           torch.from_numpy(b).to(dev) if b is not None else None).cpu().numpy()
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, s.stride_h, s.stride_w,
                        s.dil_h, s.dil_w, s.group)
        assert rel_err(got, oracle.conv_forward(g, x, w, b, gate=False)) <= 1e-4, name
        plan.close()


def test_pointwise_batches_that_end_inside_a_tile(pkg, oracle, synth):
    """Pointwise layers whose images do not fill their last tile, and partial batches that end inside
    one (ragged N * H * W): both LDS-tiled kernel families against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    for k, s in enumerate([synth.shape("f28", 37, 24, 28, 28, 40, 1, sparsity=0.9),
                           synth.shape("f14", 150, 40, 14, 14, 96, 1, sparsity=0.95, group=2),
                           synth.shape("f56", 9, 16, 56, 56, 16, 1, sparsity=0.8, bias=False)]):
        w, b, x = synth.pruned_weights(s, 70 + k), synth.bias_vector(s, 80 + k), synth.activations(s, 90 + k)
        g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
        want = oracle.conv_forward(g, x, w, b, gate=False, threads=4)
        for kernel in (pkg.KERNEL_TILED, pkg.KERNEL_JIT):
            plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=kernel)
            plan.weight_align(w)
            bias = torch.from_numpy(b).to(dev) if b is not None else None
            top = plan.forward(torch.from_numpy(x).to(dev), bias).cpu().numpy()
            assert rel_err(top, want) <= 1e-4, (s.name, plan.kernel_name)
            part = plan.forward(torch.from_numpy(x[:5]).to(dev), bias).cpu().numpy()
            assert rel_err(part, want[:5]) <= 1e-4, (s.name, plan.kernel_name)
            plan.close()


def test_forward_is_capturable_in_a_hip_graph(pkg, oracle, synth):
    """escoin_forward only enqueues work on the caller's stream (no allocation, no synchronisation
    after WeightAlign and the first launch): a whole chain of layer calls can be captured in a HIP
    graph and replayed, which is how a launch-bound net (LeNet, GoogLeNet's small 1x1 layers) would
    be driven.  Replays must reproduce the eager results on new inputs."""
    import torch
    dev = torch.device("cuda:0")
    shapes = [synth.lenet_conv2(N=8)[0], synth.googlenet_1x1(N=8)[30], synth.resnet50_3x3(N=8)[3]]
    plans, xs, tops, biases, ws = [], [], [], [], []
    for k, s in enumerate(shapes):
        w, b = synth.pruned_weights(s, 11 + k), synth.bias_vector(s, 21 + k)
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s))
        plan.weight_align(w)
        plans.append(plan); ws.append((w, b))
        xs.append(torch.from_numpy(synth.activations(s, 31 + k)).to(dev))
        biases.append(torch.from_numpy(b).to(dev) if b is not None else None)
        oh, ow = synth.out_hw(s)
        tops.append(torch.empty((s.N, s.M, oh, ow), device=dev))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                  # warm-up on the capture stream (first-launch attributes)
        for p, x, b, t in zip(plans, xs, biases, tops):
            p.forward(x, b, t)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for p, x, b, t in zip(plans, xs, biases, tops):
            p.forward(x, b, t)
    for rep in range(2):
        for k, (s, x) in enumerate(zip(shapes, xs)):
            x.copy_(torch.from_numpy(synth.activations(s, 41 + 10 * rep + k)).to(dev))
        for t in tops:
            t.fill_(-7.0)
        graph.replay()
        torch.cuda.synchronize()
        for k, s in enumerate(shapes):
            g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, 1, 1, 1, 1, s.group)
            want = oracle.conv_forward(g, xs[k].cpu().numpy(), ws[k][0], ws[k][1], gate=False, threads=4)
            assert rel_err(tops[k].cpu().numpy(), want) <= 1e-4, (s.name, rep)
    for p in plans:
        p.close()


def test_export_import_aligned_round_trip(pkg, oracle, synth):
    """escoin_plan_export_aligned -> escoin_plan_import_aligned: a generated-code plan restored from the
    persisted blob (CSR + channel deal + unit table + code) loads the code as it is, computes
    bit-identical outputs, and costs a fraction of WeightAlign; a blob for another batch / geometry falls back to
    aligning from its CSR; a damaged blob is refused."""
    import torch
    dev = torch.device("cuda:0")
    for s in (synth.resnet50_3x3(N=8)[2], synth.googlenet_1x1(N=8)[13], synth.alexnet(N=4)[0]):
        w, b = synth.pruned_weights(s, 5), synth.bias_vector(s, 6)
        x = torch.from_numpy(synth.activations(s, 7)).to(dev)
        bd = torch.from_numpy(b).to(dev) if b is not None else None
        plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256)
        plan.weight_align(w)
        want = plan.forward(x, bd).cpu().numpy()
        blob = plan.export_aligned()
        assert plan.stat("code_bytes") > 0 and blob.size > plan.stat("code_bytes")
        t_align = plan.align_ms
        # (1) same options: the persisted code object is loaded as it is
        p2 = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256)
        assert p2.import_aligned(blob) is True
        assert p2.kernel_name == plan.kernel_name and p2.tiling_info == plan.tiling_info
        assert np.array_equal(p2.forward(x, bd).cpu().numpy(), want), s.name
        assert p2.stat("code_bytes") == plan.stat("code_bytes")
        rp, ci, va, ng = plan.get_csr()
        rp2, ci2, va2, ng2 = p2.get_csr()
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and np.array_equal(va, va2) and np.array_equal(ng, ng2)
        assert np.array_equal(p2.export_aligned(), blob)          # and it can be persisted again
        print("%s: weight_align %.1f ms, import_aligned %.1f ms" % (s.name, t_align, p2.align_ms))
        # (2) another tiling batch: the code does not fit -- aligned from the CSR instead, same numbers
        p3 = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT)
        assert p3.import_aligned(blob) is False
        got = p3.forward(x, bd).cpu().numpy()
        assert float(np.abs(got - want).max()) <= 1e-4 * max(1e-6, float(np.abs(want).max()))
        # (3) other weights' geometry, a truncated blob, a flipped magic: refused
        other = pkg.Plan(pkg.ConvDesc.from_shape(s._replace(M=s.M * 2)))
        for bad, p in ((blob, other), (blob[:-5], pkg.Plan(pkg.ConvDesc.from_shape(s))),
                       (np.concatenate([blob[:1] ^ 0xFF, blob[1:]]), pkg.Plan(pkg.ConvDesc.from_shape(s)))):
            with pytest.raises(pkg.EscoinError):
                p.import_aligned(bad)
        # (4) the content tags (round 6): one flipped byte in the code section, one in the CSR values, and a blob spliced
        # from the CSR of one export and the code of another (same sizes, other weights, each section's own tag intact)
        # -- all refused: a receiver can never run code that was not generated from the CSR it carries
        import struct
        plan_b = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256)
        wb = w.copy()
        wb.reshape(-1)[np.flatnonzero(wb)] *= np.float32(1.5)    # same pattern, other values: same section sizes
        plan_b.weight_align(wb)
        blob_b = plan_b.export_aligned()
        magic, version, total, nnz, jit_bytes, csr_tag, jit_tag, pair_tag = struct.unpack_from("<IIQQQQQQ", blob.tobytes())
        assert total == blob.size and jit_bytes > 0 and version == 2
        jit_at = total - jit_bytes
        tampered = [blob.copy(), blob.copy()]
        tampered[0][jit_at + jit_bytes // 2] ^= 0x01             # inside the code object
        tampered[1][jit_at - 4 * nnz + 2] ^= 0x40                # inside the CSR values
        tb = struct.unpack_from("<IIQQQQQQ", blob_b.tobytes())
        if blob_b.size == blob.size and tb[4] == jit_bytes:
            spliced = blob.copy()
            spliced[jit_at:] = blob_b[jit_at:]                   # A's CSR, B's code ...
            spliced[40:48] = blob_b[40:48]                       # ... with B's (valid) code tag; the pair tag is still A's
            tampered.append(spliced)
        for bad in tampered:
            q = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256)
            with pytest.raises(pkg.EscoinError) as e:
                q.import_aligned(bad)
            assert "content tag" in str(e.value)
            q.close()
        # the device-buffer import (what an RCCL broadcast hands over): same result as the host import
        p4 = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256)
        assert p4.import_aligned(torch.from_numpy(blob).to(dev)) is True
        assert np.array_equal(p4.forward(x, bd).cpu().numpy(), want), s.name
        # where the code lives (round 6): executable device memory the library fills itself is the default and must be
        # what this box uses; option code_loader = 1 sends the same words through the HIP module loader -- same blob,
        # same results, either way round between exporter and importer
        assert plan.stat("code_direct") == 1 and p2.stat("code_direct") == 1 and p4.stat("code_direct") == 1
        p5 = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256, code_loader=1)
        p5.weight_align(w)
        assert p5.stat("code_direct") == 0 and p5.stat("code_bytes") == plan.stat("code_bytes")
        assert np.array_equal(p5.forward(x, bd).cpu().numpy(), want), s.name
        assert np.array_equal(p5.export_aligned(), blob)
        p6 = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_JIT, tiling_batch=256, code_loader=1)
        assert p6.import_aligned(blob) is True and p6.stat("code_direct") == 0
        assert np.array_equal(p6.forward(x, bd).cpu().numpy(), want), s.name
        print("%s: weight_align %.1f ms direct / %.1f ms code object loader; import %.1f / %.1f ms" % (s.name, t_align, p5.align_ms, p2.align_ms, p6.align_ms))
        for q in (plan, p2, p3, p4, p5, p6, other, plan_b):
            q.close()
    g = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, group=s.group)
    ref = oracle.conv_forward(g, x.cpu().numpy(), w, b, gate=False)
    assert float(np.abs(want - ref).max()) <= 1e-4 * float(np.abs(ref).max())


def test_caffe_test_chain_from_persisted_aligned_form(tmp_path):
    """resnet50_chain: --save-aligned, then a second run --load-aligned: every layer restored without the
    channel deal / generator / assembler, oracle check green."""
    path = str(tmp_path / "chain_aligned.npz")
    text = _tool("--model", "resnet50_chain", "--batch", "4", "--iterations", "1", "--save-aligned", path)
    assert "aligned form of 52 layers written" in text, text
    text = _tool("--model", "resnet50_chain", "--batch", "4", "--iterations", "1", "--load-aligned", path, "--check")
    assert "16 code objects loaded as persisted" in text, text
    assert "oracle check: worst relative error" in text


def test_caffe_test_chain_prune_1x1_what_if():
    text = _tool("--model", "resnet50_chain", "--batch", "4", "--iterations", "1", "--prune-1x1", "90", "--check")
    assert "1x1 pruned @90 %" in text and "with the 1x1 layers on the sparse path" in text, text
    assert text.count("oracle check: worst relative error") == 2
