"""The weight hand-off formats either side of WeightAlign: the .caffemodel
reader is checked against files serialised by the official protobuf runtime (message classes
built from a descriptor that restates the relevant caffe.proto fields), in both the current
LayerParameter and the deprecated V1LayerParameter forms, and the aligned-CSR file round-trips
through the oracle's dense2csr."""
import numpy as np
import pytest



@pytest.fixture(scope="module")
def cm(pkg):
    from caffe_escoin_amd import caffemodel
    return caffemodel


def _proto_classes():
    """NetParameter & friends for the official runtime, fields numbered as in
    /root/reference/src/caffe/proto/caffe.proto (only the ones on this path)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "escoin_test_caffe.proto"
    fd.package = "escoin_test_caffe"
    fd.syntax = "proto2"

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for fname, num, typ, label, extra in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, typ, label
            if "type_name" in extra:
                f.type_name = ".escoin_test_caffe." + extra["type_name"]
            if extra.get("packed"):
                f.options.packed = True

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("BlobShape", [("dim", 1, F.TYPE_INT64, REP, {"packed": True})])
    msg("BlobProto", [("num", 1, F.TYPE_INT32, OPT, {}), ("channels", 2, F.TYPE_INT32, OPT, {}),
                      ("height", 3, F.TYPE_INT32, OPT, {}), ("width", 4, F.TYPE_INT32, OPT, {}),
                      ("data", 5, F.TYPE_FLOAT, REP, {"packed": True}),
                      ("shape", 7, F.TYPE_MESSAGE, OPT, {"type_name": "BlobShape"})])
    msg("ConvolutionParameter", [
        ("num_output", 1, F.TYPE_UINT32, OPT, {}), ("bias_term", 2, F.TYPE_BOOL, OPT, {}),
        ("pad", 3, F.TYPE_UINT32, REP, {}), ("kernel_size", 4, F.TYPE_UINT32, REP, {}),
        ("group", 5, F.TYPE_UINT32, OPT, {}), ("stride", 6, F.TYPE_UINT32, REP, {}),
        ("pad_h", 9, F.TYPE_UINT32, OPT, {}), ("pad_w", 10, F.TYPE_UINT32, OPT, {}),
        ("kernel_h", 11, F.TYPE_UINT32, OPT, {}), ("kernel_w", 12, F.TYPE_UINT32, OPT, {}),
        ("stride_h", 13, F.TYPE_UINT32, OPT, {}), ("stride_w", 14, F.TYPE_UINT32, OPT, {}),
        ("dilation", 18, F.TYPE_UINT32, REP, {})])
    msg("LayerParameter", [
        ("name", 1, F.TYPE_STRING, OPT, {}), ("type", 2, F.TYPE_STRING, OPT, {}),
        ("bottom", 3, F.TYPE_STRING, REP, {}), ("top", 4, F.TYPE_STRING, REP, {}),
        ("blobs", 7, F.TYPE_MESSAGE, REP, {"type_name": "BlobProto"}),
        ("convolution_param", 106, F.TYPE_MESSAGE, OPT, {"type_name": "ConvolutionParameter"})])
    msg("V1LayerParameter", [
        ("bottom", 2, F.TYPE_STRING, REP, {}), ("top", 3, F.TYPE_STRING, REP, {}),
        ("name", 4, F.TYPE_STRING, OPT, {}), ("type", 5, F.TYPE_INT32, OPT, {}),
        ("blobs", 6, F.TYPE_MESSAGE, REP, {"type_name": "BlobProto"}),
        ("blobs_lr", 7, F.TYPE_FLOAT, REP, {}),
        ("convolution_param", 10, F.TYPE_MESSAGE, OPT, {"type_name": "ConvolutionParameter"})])
    msg("NetParameter", [
        ("name", 1, F.TYPE_STRING, OPT, {}),
        ("layers", 2, F.TYPE_MESSAGE, REP, {"type_name": "V1LayerParameter"}),
        ("input", 3, F.TYPE_STRING, REP, {}), ("input_dim", 4, F.TYPE_INT32, REP, {}),
        ("force_backward", 5, F.TYPE_BOOL, OPT, {}),
        ("layer", 100, F.TYPE_MESSAGE, REP, {"type_name": "LayerParameter"})])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("escoin_test_caffe." + n))
            for n in ("NetParameter", "LayerParameter", "BlobProto")}


def _weights(synth, seed=3):
    s1 = synth.shape("conv1", 1, 3, 11, 11, 8, 3, pad=1, sparsity=0.0)
    s2 = synth.shape("conv2", 1, 8, 9, 9, 12, 3, pad=1, group=2, sparsity=0.8)
    return [(s1, synth.pruned_weights(s1, seed), synth.bias_vector(s1, seed)),
            (s2, synth.pruned_weights(s2, seed + 1), synth.bias_vector(s2, seed + 1))]


def test_reads_official_runtime_output_current_format(cm, synth):
    cls = _proto_classes()
    net = cls["NetParameter"]()
    net.name = "tiny"
    net.input.append("data")
    net.input_dim.extend([1, 3, 11, 11])
    for s, w, b in _weights(synth):
        l = net.layer.add()
        l.name, l.type = s.name, "Convolution"
        l.bottom.append("data")
        l.top.append(s.name)
        for arr in (w, b):
            blob = l.blobs.add()
            blob.shape.dim.extend(arr.shape)
            blob.data.extend(arr.reshape(-1).tolist())
        cp = l.convolution_param
        cp.num_output, cp.group = s.M, s.group
        cp.kernel_size.append(s.KH)
        cp.pad.append(s.pad_h)
        cp.stride.append(1)
    relu = net.layer.add()
    relu.name, relu.type = "relu1", "ReLU"
    name, layers = cm.parse_net(net.SerializeToString())
    assert name == "tiny" and [l.name for l in layers] == ["conv1", "conv2", "relu1"]
    got = cm.conv_weights(layers)
    assert sorted(got) == ["conv1", "conv2"]
    for s, w, b in _weights(synth):
        gw, gb = got[s.name]
        assert gw.shape == w.shape and np.array_equal(gw.view(np.uint32), w.view(np.uint32))
        assert np.array_equal(gb.view(np.uint32), b.view(np.uint32))
    cp = layers[1].conv_param
    assert (cp["num_output"], cp["group"], cp["kernel_h"], cp["kernel_w"], cp["pad_h"], cp["pad_w"],
            cp["stride_h"], cp["stride_w"], cp["dilation_h"]) == (12, 2, 3, 3, 1, 1, 1, 1, 1)
    assert cp["bias_term"] is True


def test_reads_official_runtime_output_v1_format(cm, synth):
    cls = _proto_classes()
    net = cls["NetParameter"]()
    net.name = "tiny_v1"
    for s, w, b in _weights(synth):
        l = net.layers.add()
        l.name, l.type = s.name, cm.V1_CONVOLUTION
        l.blobs_lr.extend([1.0, 2.0])
        for arr in (w, b):
            blob = l.blobs.add()
            dims = [1] * (4 - arr.ndim) + list(arr.shape)
            blob.num, blob.channels, blob.height, blob.width = dims
            blob.data.extend(arr.reshape(-1).tolist())
        cp = l.convolution_param
        cp.num_output, cp.group, cp.kernel_h, cp.kernel_w = s.M, s.group, s.KH, s.KW
        cp.pad_h, cp.pad_w = s.pad_h, s.pad_w
    _, layers = cm.parse_net(net.SerializeToString())
    got = cm.conv_weights(layers)
    for s, w, b in _weights(synth):
        gw, gb = got[s.name]
        assert gw.shape == w.shape and np.array_equal(gw, w)
        assert gb.shape == (s.M,) and np.array_equal(gb, b)   # legacy 1x1x1xM dims flattened


@pytest.mark.parametrize("v1", [False, True])
def test_writer_is_read_by_the_official_runtime(tmp_path, v1, cm, synth):
    cls = _proto_classes()
    layers = [cm.CaffeLayer(s.name, "Convolution", [w, b], cm.conv_param_of(s)) for s, w, b in _weights(synth)]
    path = str(tmp_path / "m.caffemodel")
    cm.write_caffemodel(path, "exported", layers, v1=v1)
    net = cls["NetParameter"]()
    with open(path, "rb") as f:
        net.ParseFromString(f.read())
    assert net.name == "exported"
    pl = net.layers if v1 else net.layer
    assert len(pl) == 2
    for (s, w, b), l in zip(_weights(synth), pl):
        assert l.name == s.name
        assert np.array_equal(np.array(l.blobs[0].data, np.float32), w.reshape(-1))
        assert l.convolution_param.kernel_h == s.KH and l.convolution_param.group == s.group
    # and by our own reader
    _, back = cm.read_caffemodel(path)
    for (s, w, b), l in zip(_weights(synth), back):
        assert np.array_equal(l.blobs[0], w) and l.conv_param["pad_w"] == s.pad_w


def test_malformed_input_is_rejected(cm):
    layers = [cm.CaffeLayer("c", "Convolution", [np.ones((2, 1, 3, 3), np.float32)], None)]
    good = cm.serialize_net("n", layers)
    with pytest.raises(ValueError):
        cm.parse_net(good[:-5])                       # truncated blob
    with pytest.raises(ValueError):
        cm.parse_net(b"\x0b\x00")                     # group wire type (3) is not supported
    # shape / data count mismatch
    bad_blob = cm._enc_len(7, cm._enc_len(1, cm._enc_varint(5))) + cm._enc_len(5, b"\0" * 8)
    bad = cm._enc_len(100, cm._enc_len(1, b"c") + cm._enc_len(2, b"Convolution") + cm._enc_len(7, bad_blob))
    with pytest.raises(ValueError):
        cm.parse_net(bad)
    assert cm.parse_net(b"") == ("", [])


def test_aligned_csr_file_roundtrip(tmp_path, cm, pkg, synth, oracle):
    entries = {}
    want = {}
    for s, w, b in _weights(synth):
        desc = pkg.ConvDesc.from_shape(s)
        mg, kd = s.M // s.group, (s.C // s.group) * s.KH * s.KW
        rps, cis, vas, ngs = [], [], [], []
        for g in range(s.group):
            rp, ci, va = oracle.dense2csr(w.reshape(s.group, mg, kd)[g])
            rps.append(rp), cis.append(ci), vas.append(va), ngs.append(len(va))
        csr = (np.concatenate(rps), np.concatenate(cis), np.concatenate(vas), np.array(ngs, np.int32))
        entries[s.name] = (desc, csr)
        want[s.name] = csr
    path = str(tmp_path / "aligned.npz")
    cm.save_aligned(path, entries)
    back = cm.load_aligned(path)
    assert sorted(back) == sorted(want)
    for name, (fields, csr) in back.items():
        d = entries[name][0]
        assert fields == [getattr(d, f) for f, _ in d._fields_]
        for a, b in zip(csr, want[name]):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    with pytest.raises(ValueError):
        np.savez(str(tmp_path / "other.npz"), magic=np.array("nope"), names=np.array([]))
        cm.load_aligned(str(tmp_path / "other.npz"))
