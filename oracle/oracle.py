"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end to the CPU oracle (oracle/liboracle.so, the C restatement of the
reference's CPU sconv path) and, when present, to oracle/_ref/libescoin_ref.so
(the reference's own header kernel compiled in place).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libescoin_ref.so")


class Geom(C.Structure):
    """Mirror of oracle_conv_geom (sconv_oracle.h)."""
    _fields_ = [(n, C.c_int) for n in
                ("C", "H", "W", "M", "KH", "KW", "pad_h", "pad_w",
                 "stride_h", "stride_w", "dil_h", "dil_w", "group")]


def geom(C_, H, W, M, KH, KW, pad_h=0, pad_w=0, stride_h=1, stride_w=1,
         dil_h=1, dil_w=1, group=1):
    return Geom(C_, H, W, M, KH, KW, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, group)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is mounted)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < max(os.path.getmtime(os.path.join(_HERE, f))
                                         for f in ("sconv_oracle.c", "sconv_oracle_f64.c")):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/include/caffe/util/sconv.hpp") and \
            (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.oracle_out_dim.restype = C.c_int
        L.oracle_out_dim.argtypes = [C.c_int] * 5
        L.oracle_padded_len.restype = C.c_long
        L.oracle_padded_len.argtypes = [C.POINTER(Geom)]
        L.oracle_dense2csr.restype = C.c_int
        L.oracle_dense2csr.argtypes = [C.c_int, C.c_int, _fp, _fp, _ip, _ip]
        L.oracle_stretch.restype = None
        L.oracle_stretch.argtypes = [C.c_int, _ip, _ip] + [C.c_int] * 6
        L.oracle_pad_input.restype = None
        L.oracle_pad_input.argtypes = [C.POINTER(Geom), _fp, _fp]
        L.oracle_sconv.restype = None
        L.oracle_sconv.argtypes = [_fp] + [C.c_int] * 9 + [_ip, _ip, _fp, C.c_int, C.c_int, _fp, C.c_int]
        for name in ("oracle_conv_forward", "oracle_conv_forward_nogate"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(Geom), C.c_int, _fp, _fp, C.c_void_p, C.c_int, _fp, C.c_int]
            f = getattr(L, name + "_f64")     # Dtype = double (sconv_oracle_f64.c)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(Geom), C.c_int, _dp, _dp, C.c_void_p, C.c_int, _dp, C.c_int]
        L.oracle_sconv_f64.restype = None
        L.oracle_sconv_f64.argtypes = [_dp] + [C.c_int] * 9 + [_ip, _ip, _dp, C.c_int, C.c_int, _dp, C.c_int]
        L.oracle_dense2csr_f64.restype = C.c_int
        L.oracle_dense2csr_f64.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip, _ip]
        L.oracle_pad_input_f64.restype = None
        L.oracle_pad_input_f64.argtypes = [C.POINTER(Geom), _dp, _dp]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    """The compiled reference kernel (None when oracle/_ref was not built)."""
    global _ref
    if _ref is None and have_ref():
        R = C.CDLL(_REF)
        R.ref_sconv_default.restype = None
        R.ref_sconv_default.argtypes = [_fp] + [C.c_int] * 9 + \
            [_ip, _ip, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int]
        R.ref_conv_forward.restype = C.c_int
        R.ref_conv_forward.argtypes = [C.POINTER(Geom), C.c_int, _fp, _fp, C.c_void_p, _fp, C.c_int]
        if hasattr(R, "ref_plan_create"):
            R.ref_blocked_supported.restype = C.c_int
            R.ref_blocked_supported.argtypes = [C.POINTER(Geom)]
            R.ref_plan_create.restype = C.c_void_p
            R.ref_plan_create.argtypes = [C.POINTER(Geom), _fp]
            R.ref_plan_destroy.restype = None
            R.ref_plan_destroy.argtypes = [C.c_void_p]
            R.ref_plan_has_blocked.restype = C.c_int
            R.ref_plan_has_blocked.argtypes = [C.c_void_p]
            R.ref_plan_ncolblocks.restype = C.c_int
            R.ref_plan_ncolblocks.argtypes = [C.c_void_p]
            R.ref_plan_forward.restype = C.c_int
            R.ref_plan_forward.argtypes = [C.c_void_p, C.c_int, _fp, C.c_void_p, _fp, C.c_int, C.c_int]
            R.ref_thread_partition.restype = None
            R.ref_thread_partition.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _ip]
        _ref = R
    return _ref


def out_hw(g):
    L = lib()
    return (L.oracle_out_dim(g.H, g.KH, g.pad_h, g.stride_h, g.dil_h),
            L.oracle_out_dim(g.W, g.KW, g.pad_w, g.stride_w, g.dil_w))


def padded_len(g):
    return lib().oracle_padded_len(C.byref(g))


def dense2csr(A):
    """A: (M, N) float32 -> rowptr[M+1], colidx[nnz], values[nnz]."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    M, N = A.shape
    values = np.zeros(M * N, np.float32)
    colidx = np.zeros(M * N, np.int32)
    rowptr = np.zeros(M + 1, np.int32)
    nnz = lib().oracle_dense2csr(M, N, A, values, colidx, rowptr)
    return rowptr, colidx[:nnz].copy(), values[:nnz].copy()


def stretch(rowptr, colidx, KH, KW, H, W, pad_h, pad_w):
    out = np.ascontiguousarray(colidx, dtype=np.int32).copy()
    lib().oracle_stretch(len(rowptr) - 1, np.ascontiguousarray(rowptr, np.int32), out,
                         KH, KW, H, W, pad_h, pad_w)
    return out


def pad_input(g, image):
    buf = np.zeros(padded_len(g), np.float32)
    lib().oracle_pad_input(C.byref(g), np.ascontiguousarray(image, np.float32).ravel(), buf)
    return buf


def sconv(g, padded, cin, rowptr, colidx, values, mout):
    """One image / one group through the restated caffe_cpu_sconv."""
    oh, ow = out_hw(g)
    out = np.zeros(mout * oh * ow, np.float32)
    lib().oracle_sconv(padded, cin, g.H, g.W, g.pad_h, g.pad_w, g.stride_h, g.stride_w,
                       g.dil_h, g.dil_w, rowptr, colidx, values, g.KH, g.KW, out, mout)
    return out.reshape(mout, oh, ow)


def ref_sconv(g, padded, cin, rowptr, colidx, values, mout, bias=None, relu=False):
    """Same call through the compiled reference kernel (sconv.hpp:594-678)."""
    oh, ow = out_hw(g)
    out = np.zeros(mout * oh * ow, np.float32)
    b = np.zeros(mout, np.float32) if bias is None else np.ascontiguousarray(bias, np.float32)
    ref().ref_sconv_default(padded, cin, g.H, g.W, g.pad_h, g.pad_w, g.stride_h, g.stride_w,
                            g.dil_h, g.dil_w, rowptr, colidx, values, g.KH, g.KW, b, out,
                            mout, int(relu))
    return out.reshape(mout, oh, ow)


def _bias_ptr(bias):
    if bias is None:
        return None, None
    b = np.ascontiguousarray(bias, np.float32)
    return b, b.ctypes.data_as(C.c_void_p)


def conv_forward_f64(g, bottom, weights, bias=None, relu=False, threads=1, gate=False):
    """conv_forward for Dtype = double (the same source text compiled with float read as double)."""
    bottom = np.ascontiguousarray(bottom, np.float64)
    weights = np.ascontiguousarray(weights, np.float64)
    N = bottom.shape[0]
    oh, ow = out_hw(g)
    top = np.zeros((N, g.M, oh, ow), np.float64)
    b = None if bias is None else np.ascontiguousarray(bias, np.float64)
    bp = None if b is None else b.ctypes.data_as(C.c_void_p)
    fn = lib().oracle_conv_forward_f64 if gate else lib().oracle_conv_forward_nogate_f64
    rc = fn(C.byref(g), N, bottom.ravel(), weights.ravel(), bp, int(relu), top.ravel(), threads)
    if rc != 0:
        raise MemoryError("oracle_conv_forward_f64 failed")
    return top


def sconv_f64(g, padded, cin, rowptr, colidx, values, mout):
    """One image / one group through the restated caffe_cpu_sconv<double>."""
    oh, ow = out_hw(g)
    out = np.zeros(mout * oh * ow, np.float64)
    lib().oracle_sconv_f64(padded, cin, g.H, g.W, g.pad_h, g.pad_w, g.stride_h, g.stride_w,
                           g.dil_h, g.dil_w, rowptr, colidx, values, g.KH, g.KW, out, mout)
    return out.reshape(mout, oh, ow)


def pad_input_f64(g, image):
    buf = np.zeros(padded_len(g), np.float64)
    lib().oracle_pad_input_f64(C.byref(g), np.ascontiguousarray(image, np.float64).ravel(), buf)
    return buf


def conv_forward(g, bottom, weights, bias=None, relu=False, threads=1, gate=True):
    """Whole-batch Forward_cpu in SCONV mode (restatement). bottom: (N,C,H,W)."""
    bottom = np.ascontiguousarray(bottom, np.float32)
    weights = np.ascontiguousarray(weights, np.float32)
    N = bottom.shape[0]
    oh, ow = out_hw(g)
    top = np.zeros((N, g.M, oh, ow), np.float32)
    keep, bp = _bias_ptr(bias)
    fn = lib().oracle_conv_forward if gate else lib().oracle_conv_forward_nogate
    rc = fn(C.byref(g), N, bottom.ravel(), weights.ravel(), bp, int(relu), top.ravel(), threads)
    if rc != 0:
        raise MemoryError("oracle_conv_forward failed")
    return top


def ref_conv_forward(g, bottom, weights, bias=None, threads=1):
    """Whole-batch forward with the compiled reference kernel doing the arithmetic."""
    bottom = np.ascontiguousarray(bottom, np.float32)
    weights = np.ascontiguousarray(weights, np.float32)
    N = bottom.shape[0]
    oh, ow = out_hw(g)
    top = np.zeros((N, g.M, oh, ow), np.float32)
    keep, bp = _bias_ptr(bias)
    rc = ref().ref_conv_forward(C.byref(g), N, bottom.ravel(), weights.ravel(), bp, top.ravel(),
                                threads)
    if rc != 0:
        raise MemoryError("ref_conv_forward failed")
    return top


class RefPlan(object):
    """The reference CPU layer after WeightAlign (oracle/_ref): CSR (and, where the reference's
    switchboard has an instantiation, the column-blocked CSR of its register-blocked kernel) built
    once, so that forward() times only what the reference's Forward_cpu does per call.

    kernel "default" = caffe_cpu_sconv's loop nest (math_functions.cpp:128-176), OpenMP over the
    batch; kernel "blocked" = sconv_unit_stride (sconv.hpp:57-589) behind the reference's thread
    grouping (cpu_info.cpp:483-605)."""

    def __init__(self, g, weights):
        self.g = g
        self._w = np.ascontiguousarray(weights, np.float32)
        self._h = ref().ref_plan_create(C.byref(g), self._w.ravel())
        if not self._h:
            raise MemoryError("ref_plan_create failed")

    @property
    def has_blocked(self):
        return bool(ref().ref_plan_has_blocked(self._h))

    @property
    def ncolblocks(self):
        return ref().ref_plan_ncolblocks(self._h)

    def forward(self, bottom, bias=None, threads=1, kernel="default", top=None):
        bottom = np.ascontiguousarray(bottom, np.float32)
        N = bottom.shape[0]
        oh, ow = out_hw(self.g)
        if top is None:
            top = np.zeros((N, self.g.M, oh, ow), np.float32)
        keep, bp = _bias_ptr(bias)
        rc = ref().ref_plan_forward(self._h, N, bottom.ravel(), bp, top.ravel(), int(threads),
                                    1 if kernel == "blocked" else 0)
        if rc != 0:
            raise RuntimeError("ref_plan_forward failed (%d)" % rc)
        return top

    def close(self):
        if self._h:
            ref().ref_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def have_ref_plan():
    return have_ref() and hasattr(ref(), "ref_plan_create")


def thread_partition(n_threads, batch, work):
    """The reference's thread grouping (cpu_info.cpp:483-540) for a team of n_threads: per thread
    its group id, its group's [begin, end) of the batch and its own [begin, end) of `work`."""
    z = lambda: np.zeros(n_threads, np.int32)
    gid, bb, be, wb, we = z(), z(), z(), z(), z()
    ref().ref_thread_partition(n_threads, batch, work, gid, bb, be, wb, we)
    return gid, bb, be, wb, we
