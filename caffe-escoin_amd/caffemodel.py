"""Weight hand-off formats on either side of WeightAlign.

Upstream side: the reference reads its pruned dense weights from a binary ``NetParameter``
(``.caffemodel``) via ``Net::CopyTrainedLayersFrom`` (src/caffe/net.cpp) and hands each
Convolution layer's ``blobs_[0]`` to ``WeightAlign``.  protoc is not part of this image, so
``read_caffemodel`` walks the protobuf wire format directly; only the fields on this path are
decoded (field numbers from src/caffe/proto/caffe.proto):

    NetParameter      .name = 1, .layers = 2 (V1LayerParameter), .layer = 100 (LayerParameter)
    LayerParameter    .name = 1, .type = 2 (string), .blobs = 7, .convolution_param = 106
    V1LayerParameter  .name = 4, .type = 5 (enum, CONVOLUTION = 4), .blobs = 6,
                      .convolution_param = 10
    BlobProto         .num/.channels/.height/.width = 1..4, .data = 5 (packed float),
                      .shape = 7 (BlobShape.dim = 1, packed int64), .double_data = 8
    ConvolutionParameter  num_output 1, bias_term 2, pad 3, kernel_size 4, group 5, stride 6,
                      pad_h 9, pad_w 10, kernel_h 11, kernel_w 12, stride_h 13, stride_w 14,
                      dilation 18

Downstream side: ``save_aligned`` / ``load_aligned`` persist what ``WeightAlign`` produced
(the per-group CSR exactly as ``escoin_plan_get_csr`` returns it) so a deployment can skip the
dense -> CSR step and go straight to ``escoin_plan_set_csr``.

``write_caffemodel`` emits the same subset; it exists for tests and for exporting synthetic
pruned models, not as a general protobuf writer.
"""
import struct

import numpy as np

V1_CONVOLUTION = 4


# ---- wire format ---------------------------------------------------------------------------
def _varint(buf, pos):
    out = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint longer than 10 bytes")


def _fields(buf):
    """Yields (field_number, wire_type, value) for one message; value is an int (varint,
    fixed32/64 raw) or a memoryview (length-delimited)."""
    buf = memoryview(buf)
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > n:
                raise ValueError("truncated fixed64")
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field %d" % fno)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            if pos + 4 > n:
                raise ValueError("truncated fixed32")
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported wire type %d (field %d)" % (wt, fno))
        yield fno, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [v]
    out = []
    pos = 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_key(fno, wt):
    return _enc_varint((fno << 3) | wt)


def _enc_len(fno, payload):
    return _enc_key(fno, 2) + _enc_varint(len(payload)) + payload


def _enc_uint(fno, x):
    return _enc_key(fno, 0) + _enc_varint(int(x))


# ---- messages ------------------------------------------------------------------------------
def _parse_blob(buf):
    dims = None
    legacy = {}
    data = None
    for fno, wt, v in _fields(buf):
        if fno == 7 and wt == 2:
            dims = []
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    dims.extend(_packed_varints(v2, w2))
        elif fno in (1, 2, 3, 4) and wt == 0:
            legacy[fno] = v
        elif fno == 5:
            if wt == 2:
                chunk = np.frombuffer(v, dtype="<f4")
            else:  # unpacked repeated float
                chunk = np.frombuffer(v, dtype="<f4", count=1)
            data = chunk if data is None else np.concatenate([data, chunk])
        elif fno == 8:
            chunk = np.frombuffer(v, dtype="<f8").astype(np.float32)
            data = chunk if data is None else np.concatenate([data, chunk])
    if data is None:
        data = np.zeros(0, np.float32)
    if dims is None:
        if legacy:
            dims = [legacy.get(i, 1) for i in (1, 2, 3, 4)]
        else:
            dims = [data.size]
    count = int(np.prod(dims)) if dims else 1
    if count != data.size:
        raise ValueError("BlobProto: shape %s holds %d values, data has %d" % (dims, count, data.size))
    return np.array(data, np.float32).reshape(dims)


def _parse_conv_param(buf):
    rep = {3: [], 4: [], 6: [], 18: []}
    one = {}
    for fno, wt, v in _fields(buf):
        if fno in rep:
            rep[fno].extend(_packed_varints(v, wt))
        elif wt == 0:
            one[fno] = v

    def hw(rep_f, h_f, w_f, default):
        if h_f in one or w_f in one:
            return one.get(h_f, default), one.get(w_f, default)
        r = rep[rep_f]
        if not r:
            return default, default
        return (r[0], r[0]) if len(r) == 1 else (r[0], r[1])

    kh, kw = hw(4, 11, 12, 0)
    ph, pw = hw(3, 9, 10, 0)
    sh, sw = hw(6, 13, 14, 1)
    d = rep[18]
    dh, dw = (1, 1) if not d else ((d[0], d[0]) if len(d) == 1 else (d[0], d[1]))
    return dict(num_output=one.get(1, 0), bias_term=bool(one.get(2, 1)), group=one.get(5, 1),
                kernel_h=kh, kernel_w=kw, pad_h=ph, pad_w=pw, stride_h=sh, stride_w=sw,
                dilation_h=dh, dilation_w=dw)


class CaffeLayer(object):
    """One layer of a .caffemodel: name, type string, blobs (numpy float32) and, for
    convolutions, the decoded ConvolutionParameter."""

    def __init__(self, name, type_, blobs, conv_param=None):
        self.name = name
        self.type = type_
        self.blobs = blobs
        self.conv_param = conv_param

    @property
    def is_convolution(self):
        return self.type in ("Convolution", "ConvolutionReLU", V1_CONVOLUTION)

    def __repr__(self):
        return "CaffeLayer(%r, %r, blobs=%s)" % (self.name, self.type, [b.shape for b in self.blobs])


def _parse_layer(buf, v1):
    f_name, f_type, f_blobs, f_conv = (4, 5, 6, 10) if v1 else (1, 2, 7, 106)
    name, type_, blobs, conv = "", None, [], None
    for fno, wt, v in _fields(buf):
        if fno == f_name and wt == 2:
            name = bytes(v).decode("utf-8")
        elif fno == f_type:
            type_ = v if v1 else bytes(v).decode("utf-8")
        elif fno == f_blobs and wt == 2:
            blobs.append(_parse_blob(v))
        elif fno == f_conv and wt == 2:
            conv = _parse_conv_param(v)
    return CaffeLayer(name, type_, blobs, conv)


def parse_net(buf):
    """Binary NetParameter -> (net name, [CaffeLayer])."""
    name, layers = "", []
    for fno, wt, v in _fields(buf):
        if fno == 1 and wt == 2:
            name = bytes(v).decode("utf-8")
        elif fno == 100 and wt == 2:
            layers.append(_parse_layer(v, v1=False))
        elif fno == 2 and wt == 2:
            layers.append(_parse_layer(v, v1=True))
    return name, layers


def read_caffemodel(path):
    with open(path, "rb") as f:
        return parse_net(f.read())


def conv_weights(layers):
    """{layer name: (weight M x C/g x KH x KW, bias or None)} for the convolution layers that
    carry blobs -- what Net::CopyTrainedLayersFrom would copy into blobs_[0], blobs_[1]."""
    out = {}
    for l in layers:
        if not l.is_convolution or not l.blobs:
            continue
        w = l.blobs[0]
        if w.ndim != 4:
            raise ValueError("layer %s: weight blob has %d axes" % (l.name, w.ndim))
        b = l.blobs[1].reshape(-1) if len(l.blobs) > 1 else None
        out[l.name] = (w, b)
    return out


# ---- writer (tests / synthetic exports) ------------------------------------------------------
def _enc_blob(a, legacy_dims=False):
    a = np.ascontiguousarray(a, dtype="<f4")
    out = b""
    if legacy_dims:
        dims = ([1] * (4 - a.ndim) + list(a.shape))[-4:]
        for f, d in zip((1, 2, 3, 4), dims):
            out += _enc_uint(f, d)
    else:
        out += _enc_len(7, _enc_len(1, b"".join(_enc_varint(d) for d in a.shape)))
    out += _enc_len(5, a.tobytes())
    return out


def _enc_conv_param(p):
    out = _enc_uint(1, p["num_output"]) + _enc_uint(2, int(p.get("bias_term", True)))
    out += _enc_uint(5, p.get("group", 1))
    out += _enc_uint(11, p["kernel_h"]) + _enc_uint(12, p["kernel_w"])
    out += _enc_uint(9, p.get("pad_h", 0)) + _enc_uint(10, p.get("pad_w", 0))
    out += _enc_uint(13, p.get("stride_h", 1)) + _enc_uint(14, p.get("stride_w", 1))
    dh, dw = p.get("dilation_h", 1), p.get("dilation_w", 1)
    if (dh, dw) != (1, 1):
        out += _enc_uint(18, dh) + _enc_uint(18, dw)
    return out


def serialize_net(name, layers, v1=False):
    """layers: iterable of CaffeLayer.  v1=True writes the deprecated V1LayerParameter form
    (with 4-D legacy blob dims), which old pruned model zoo files still use."""
    out = _enc_len(1, name.encode("utf-8"))
    for l in layers:
        if v1:
            body = _enc_len(4, l.name.encode("utf-8"))
            body += _enc_uint(5, V1_CONVOLUTION if l.is_convolution else 0)
            for b in l.blobs:
                body += _enc_len(6, _enc_blob(b, legacy_dims=True))
            if l.conv_param:
                body += _enc_len(10, _enc_conv_param(l.conv_param))
            out += _enc_len(2, body)
        else:
            body = _enc_len(1, l.name.encode("utf-8")) + _enc_len(2, str(l.type).encode("utf-8"))
            for b in l.blobs:
                body += _enc_len(7, _enc_blob(b))
            if l.conv_param:
                body += _enc_len(106, _enc_conv_param(l.conv_param))
            out += _enc_len(100, body)
    return out


def write_caffemodel(path, name, layers, v1=False):
    with open(path, "wb") as f:
        f.write(serialize_net(name, layers, v1=v1))


def conv_param_of(shape):
    """ConvolutionParameter dict for a synth.ConvShape."""
    return dict(num_output=shape.M, bias_term=bool(shape.bias), group=shape.group,
                kernel_h=shape.KH, kernel_w=shape.KW, pad_h=shape.pad_h, pad_w=shape.pad_w,
                stride_h=shape.stride_h, stride_w=shape.stride_w, dilation_h=shape.dil_h,
                dilation_w=shape.dil_w)


# ---- aligned (post-WeightAlign) form -----------------------------------------------------------
_ALIGNED_MAGIC = "escoin-aligned-csr-v1"


def save_aligned(path, layers):
    """layers: {name: (ConvDesc-like with the 16 int fields, (rowptr, colidx, values, nnz_per_group))}.
    Column indices are the unstretched ones (escoin_plan_get_csr(stretched=0)), so the file does not
    depend on the input geometry a later plan pads for."""
    arrays = {"magic": np.array(_ALIGNED_MAGIC), "names": np.array(sorted(layers))}
    for name, (desc, (rp, ci, va, ng)) in layers.items():
        arrays[name + "/desc"] = np.array([getattr(desc, f) for f, _ in desc._fields_], np.int32)
        arrays[name + "/rowptr"] = np.asarray(rp, np.int32)
        arrays[name + "/colidx"] = np.asarray(ci, np.int32)
        arrays[name + "/values"] = np.asarray(va, np.float32)
        arrays[name + "/nnz"] = np.asarray(ng, np.int32)
    np.savez(path, **arrays)


def load_aligned(path):
    """-> {name: (desc field list, (rowptr, colidx, values, nnz_per_group))}"""
    z = np.load(path, allow_pickle=False)
    if str(z["magic"]) != _ALIGNED_MAGIC:
        raise ValueError("%s is not an aligned-CSR file" % path)
    out = {}
    for name in z["names"].tolist():
        out[name] = (z[name + "/desc"].tolist(),
                     (z[name + "/rowptr"], z[name + "/colidx"], z[name + "/values"], z[name + "/nnz"]))
    return out


def float_bits(x):
    """Helper for tests: the IEEE-754 bits of a float32 scalar."""
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]
