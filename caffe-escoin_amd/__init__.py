"""caffe-escoin_amd: MI355X-native direct sparse convolution forward (Escoin's SCONV path).

The product is the C-ABI shared library ``libescoin_hip.so`` (hand-written HIP kernels for
gfx950, see ``csrc/`` and ``include/escoin.h``) plus the C++ Caffe-compatible Layer/Blob shim
in ``caffe_shim/``.  This Python module is only a ctypes binding of that C ABI, used by the
tests and ``bench.py``; torch supplies device memory, streams and ``torch.distributed`` --
plumbing, not the product.

No silent CPU fallback: if the library is missing, or no HIP device is visible, the GPU entry
points raise.  Caffe::CPU mode is its own explicit set of entry points (``Plan.weight_align_cpu`` /
``Plan.forward_cpu``), implemented in the same library and usable without a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import shard, synth  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ESCOIN_LIB: A/B experiments against another build of the library
LIB_PATH = os.environ.get("ESCOIN_LIB") or os.path.join(_HERE, "libescoin_hip.so")

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_TILED, KERNEL_DENSE, KERNEL_JIT = 0, 1, 2, 3, 4
CONV_MODE_LOWERED_GEMM, CONV_MODE_LOWERED_SPARSE, CONV_MODE_SCONV, CONV_MODE_SCONV_PAR = 0, 1, 2, 3

# every symbol include/escoin.h declares (tests check the library exports all of them)
API_SYMBOLS = [
    "escoin_last_error", "escoin_device_count", "escoin_out_shape", "escoin_padded_len",
    "escoin_plan_create", "escoin_plan_destroy", "escoin_plan_set_option",
    "escoin_weight_align", "escoin_plan_set_csr", "escoin_plan_nnz", "escoin_plan_get_csr",
    "escoin_plan_workspace_bytes", "escoin_plan_kernel_name", "escoin_plan_tiling_info", "escoin_forward",
    "escoin_plan_export_aligned", "escoin_plan_import_aligned", "escoin_plan_import_aligned_dev", "escoin_plan_stat",
    "escoin_gpu_sconv", "escoin_gpu_stretch", "escoin_copy_input_data",
    "escoin_gpu_sparse_dense2csr", "escoin_gpu_sparse_csrmm",
    # Dtype = double
    "escoin_weight_align_f64", "escoin_plan_set_csr_f64", "escoin_plan_get_csr_f64", "escoin_forward_f64",
    "escoin_gpu_sconv_f64", "escoin_copy_input_data_f64", "escoin_gpu_sparse_csrmm_f64",
    "escoin_gpu_sparse_dense2csr_f64",
    # Caffe::CPU mode
    "escoin_cpu_kernel_name", "escoin_cpu_kernel_select", "escoin_weight_align_cpu", "escoin_weight_align_cpu_f64",
    "escoin_forward_cpu", "escoin_forward_cpu_f64", "escoin_cpu_sconv", "escoin_cpu_sconv_f64",
    "escoin_cpu_sparse_dense2csr", "escoin_cpu_sparse_dense2csr_f64",
]


class EscoinError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """Mirror of escoin_conv_desc (include/escoin.h)."""
    _fields_ = [(n, C.c_int) for n in
                ("N", "C", "H", "W", "M", "KH", "KW", "pad_h", "pad_w", "stride_h", "stride_w",
                 "dil_h", "dil_w", "group", "has_bias", "fuse_relu")]

    @classmethod
    def from_shape(cls, s, N=None, fuse_relu=False):
        """From a synth.ConvShape."""
        return cls(s.N if N is None else N, s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w,
                   s.stride_h, s.stride_w, s.dil_h, s.dil_w, s.group, int(bool(s.bias)),
                   int(bool(fuse_relu)))


def build(verbose=False):
    """Compile libescoin_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=out)
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EscoinError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                          " (there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, cp = C.c_void_p, C.c_int, C.c_char_p
    dp = C.POINTER(ConvDesc)
    L.escoin_last_error.restype = cp
    L.escoin_last_error.argtypes = []
    L.escoin_device_count.restype = ip
    L.escoin_device_count.argtypes = []
    L.escoin_out_shape.restype = ip
    L.escoin_out_shape.argtypes = [dp, C.POINTER(ip), C.POINTER(ip)]
    L.escoin_padded_len.restype = C.c_long
    L.escoin_padded_len.argtypes = [dp]
    L.escoin_plan_create.restype = ip
    L.escoin_plan_create.argtypes = [dp, C.POINTER(vp)]
    L.escoin_plan_destroy.restype = ip
    L.escoin_plan_destroy.argtypes = [vp]
    L.escoin_plan_set_option.restype = ip
    L.escoin_plan_set_option.argtypes = [vp, cp, ip]
    L.escoin_weight_align.restype = ip
    L.escoin_weight_align.argtypes = [vp, vp, ip, vp]
    L.escoin_plan_set_csr.restype = ip
    L.escoin_plan_set_csr.argtypes = [vp, vp, vp, vp, vp, vp]
    L.escoin_plan_nnz.restype = C.c_long
    L.escoin_plan_nnz.argtypes = [vp, ip]
    L.escoin_plan_get_csr.restype = ip
    L.escoin_plan_get_csr.argtypes = [vp, vp, vp, vp, ip]
    L.escoin_plan_workspace_bytes.restype = C.c_size_t
    L.escoin_plan_workspace_bytes.argtypes = [vp]
    L.escoin_plan_kernel_name.restype = cp
    L.escoin_plan_kernel_name.argtypes = [vp]
    L.escoin_plan_tiling_info.restype = cp
    L.escoin_plan_tiling_info.argtypes = [vp]
    L.escoin_forward.restype = ip
    L.escoin_forward.argtypes = [vp, vp, vp, vp, ip, vp]
    L.escoin_plan_export_aligned.restype = ip
    L.escoin_plan_export_aligned.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.escoin_plan_import_aligned.restype = ip
    L.escoin_plan_import_aligned.argtypes = [vp, vp, C.c_size_t, vp]
    L.escoin_plan_import_aligned_dev.restype = ip
    L.escoin_plan_import_aligned_dev.argtypes = [vp, vp, C.c_size_t, vp]
    L.escoin_plan_stat.restype = C.c_long
    L.escoin_plan_stat.argtypes = [vp, cp]
    L.escoin_gpu_sconv.restype = ip
    L.escoin_gpu_sconv.argtypes = [ip, ip, vp, ip, vp, vp, vp, vp] + [ip] * 10 + [vp, ip, ip, vp]
    L.escoin_gpu_stretch.restype = ip
    L.escoin_gpu_stretch.argtypes = [vp, vp] + [ip] * 7 + [vp]
    L.escoin_copy_input_data.restype = ip
    L.escoin_copy_input_data.argtypes = [vp, vp] + [ip] * 5 + [vp]
    L.escoin_gpu_sparse_dense2csr.restype = ip
    L.escoin_gpu_sparse_dense2csr.argtypes = [ip, ip, vp, vp, vp, vp, vp, C.POINTER(ip), vp]
    L.escoin_gpu_sparse_csrmm.restype = ip
    L.escoin_gpu_sparse_csrmm.argtypes = [ip, ip, ip, ip, C.c_float, vp, vp, vp, vp, C.c_float, vp, vp]
    for suffix in ("", "_f64"):
        real = C.c_float if suffix == "" else C.c_double
        f = getattr(L, "escoin_weight_align_cpu" + suffix)
        f.restype, f.argtypes = ip, [vp, vp]
        f = getattr(L, "escoin_forward_cpu" + suffix)
        f.restype, f.argtypes = ip, [vp, vp, vp, vp, ip, ip]
        f = getattr(L, "escoin_cpu_sconv" + suffix)
        f.restype, f.argtypes = ip, [vp] + [ip] * 9 + [vp, vp, vp, ip, ip, vp, vp, ip, ip]
        f = getattr(L, "escoin_cpu_sparse_dense2csr" + suffix)
        f.restype, f.argtypes = ip, [ip, ip, vp, vp, vp, vp]
        if suffix:
            L.escoin_weight_align_f64.restype = ip
            L.escoin_weight_align_f64.argtypes = [vp, vp, ip, vp]
            L.escoin_plan_set_csr_f64.restype = ip
            L.escoin_plan_set_csr_f64.argtypes = [vp, vp, vp, vp, vp, vp]
            L.escoin_plan_get_csr_f64.restype = ip
            L.escoin_plan_get_csr_f64.argtypes = [vp, vp, vp, vp, ip]
            L.escoin_forward_f64.restype = ip
            L.escoin_forward_f64.argtypes = [vp, vp, vp, vp, ip, vp]
            L.escoin_gpu_sconv_f64.restype = ip
            L.escoin_gpu_sconv_f64.argtypes = L.escoin_gpu_sconv.argtypes
            L.escoin_copy_input_data_f64.restype = ip
            L.escoin_copy_input_data_f64.argtypes = L.escoin_copy_input_data.argtypes
            L.escoin_gpu_sparse_dense2csr_f64.restype = ip
            L.escoin_gpu_sparse_dense2csr_f64.argtypes = L.escoin_gpu_sparse_dense2csr.argtypes
            L.escoin_gpu_sparse_csrmm_f64.restype = ip
            L.escoin_gpu_sparse_csrmm_f64.argtypes = [ip, ip, ip, ip, real, vp, vp, vp, vp, real, vp, vp]
    L.escoin_cpu_kernel_name.restype = cp
    L.escoin_cpu_kernel_name.argtypes = []
    L.escoin_cpu_kernel_select.restype = ip
    L.escoin_cpu_kernel_select.argtypes = [cp]
    _lib = L
    return L


def check(rc, what="escoin call"):
    if rc != 0:
        raise EscoinError("%s failed (%d): %s" % (what, rc, lib().escoin_last_error().decode()))


def device_count():
    return lib().escoin_device_count()


def cpu_kernel_select(which):
    """Pin the host kernel's flavour ("avx2", "avx512", "auto"); raises when this CPU lacks it."""
    check(lib().escoin_cpu_kernel_select(which.encode()), "escoin_cpu_kernel_select(%s)" % which)


def cpu_kernel_name():
    """Which flavour of the host kernel Caffe::CPU mode runs on this machine."""
    return lib().escoin_cpu_kernel_name().decode()


def out_shape(desc):
    oh, ow = C.c_int(), C.c_int()
    check(lib().escoin_out_shape(C.byref(desc), C.byref(oh), C.byref(ow)), "escoin_out_shape")
    return oh.value, ow.value


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Plan(object):
    """escoin_plan: one ConvolutionLayer's sparse state (CSR + weight streams) on one device."""

    def __init__(self, desc, kernel=KERNEL_AUTO, conv_mode=CONV_MODE_SCONV_PAR, **options):
        self.desc = desc
        self._h = C.c_void_p()
        check(lib().escoin_plan_create(C.byref(desc), C.byref(self._h)), "escoin_plan_create")
        if kernel != KERNEL_AUTO:
            self.set_option("kernel", kernel)
        if conv_mode != CONV_MODE_SCONV_PAR:
            self.set_option("conv_mode", conv_mode)
        for k, v in options.items():      # e.g. tiling_batch=256, dense_gate=1, dense_threshold_pct=30
            self.set_option(k, v)
        self.out_hw = out_shape(desc)

    def close(self):
        if self._h:
            lib().escoin_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        check(lib().escoin_plan_set_option(self._h, key.encode(), int(value)),
              "escoin_plan_set_option(%s)" % key)

    def weight_align(self, dense_w, stream=None):
        """WeightAlign().  dense_w: numpy (host) or torch CUDA tensor (device), M x C/g x KH x KW;
        float64 input makes a Dtype = double plan (escoin_weight_align_f64)."""
        if isinstance(dense_w, np.ndarray):
            f64 = dense_w.dtype == np.float64
            w = np.ascontiguousarray(dense_w, np.float64 if f64 else np.float32)
            fn = lib().escoin_weight_align_f64 if f64 else lib().escoin_weight_align
            check(fn(self._h, _np_ptr(w), 0, stream), "escoin_weight_align")
        else:
            w = dense_w.contiguous()
            assert w.is_cuda and w.dtype.is_floating_point and w.element_size() in (4, 8)
            fn = lib().escoin_weight_align_f64 if w.element_size() == 8 else lib().escoin_weight_align
            check(fn(self._h, C.c_void_p(w.data_ptr()), 1, stream), "escoin_weight_align")

    # ---- Caffe::CPU mode (no device needed) ----------------------------------------------------
    def weight_align_cpu(self, dense_w):
        """WeightAlign() in CPU mode: host CSR only.  float32 or float64 numpy."""
        f64 = dense_w.dtype == np.float64
        w = np.ascontiguousarray(dense_w, np.float64 if f64 else np.float32)
        fn = lib().escoin_weight_align_cpu_f64 if f64 else lib().escoin_weight_align_cpu
        check(fn(self._h, _np_ptr(w)), "escoin_weight_align_cpu")

    def forward_cpu(self, bottom, bias=None, n_threads=0, out=None):
        """Forward_cpu on numpy arrays (float32 or float64, matching the aligned weights).  `out`: a C-contiguous top
        blob to write into (a Caffe top blob is allocated once at Reshape, not per Forward)."""
        d = self.desc
        f64 = bottom.dtype == np.float64
        dt = np.float64 if f64 else np.float32
        x = np.ascontiguousarray(bottom, dt)
        assert tuple(x.shape[1:]) == (d.C, d.H, d.W), "bottom shape mismatch"
        b = None if bias is None else np.ascontiguousarray(bias, dt)
        shape = (x.shape[0], d.M) + tuple(self.out_hw)
        if out is None:
            top = np.empty(shape, dt)
        else:
            assert out.dtype == dt and tuple(out.shape) == shape and out.flags["C_CONTIGUOUS"], "out: wrong dtype / shape / layout"
            top = out
        fn = lib().escoin_forward_cpu_f64 if f64 else lib().escoin_forward_cpu
        check(fn(self._h, _np_ptr(x), _np_ptr(b) if b is not None else None, _np_ptr(top), x.shape[0],
                 int(n_threads)), "escoin_forward_cpu")
        return top

    def set_csr(self, rowptr, colidx, values, nnz_per_group, stream=None):
        rp = np.ascontiguousarray(rowptr, np.int32)
        ci = np.ascontiguousarray(colidx, np.int32)
        f64 = isinstance(values, np.ndarray) and values.dtype == np.float64
        va = np.ascontiguousarray(values, np.float64 if f64 else np.float32)
        ng = np.ascontiguousarray(nnz_per_group, np.int32)
        fn = lib().escoin_plan_set_csr_f64 if f64 else lib().escoin_plan_set_csr
        check(fn(self._h, _np_ptr(rp), _np_ptr(ci), _np_ptr(va), _np_ptr(ng), stream), "escoin_plan_set_csr")

    def nnz(self, group=-1):
        n = lib().escoin_plan_nnz(self._h, group)
        if n < 0:
            check(int(n), "escoin_plan_nnz")
        return int(n)

    def get_csr(self, stretched=False):
        d = self.desc
        mg = d.M // d.group
        nnz = self.nnz()
        f64 = self.stat("is_f64") == 1
        rp = np.zeros(d.group * (mg + 1), np.int32)
        ci = np.zeros(max(nnz, 1), np.int32)
        va = np.zeros(max(nnz, 1), np.float64 if f64 else np.float32)
        fn = lib().escoin_plan_get_csr_f64 if f64 else lib().escoin_plan_get_csr
        check(fn(self._h, _np_ptr(rp), _np_ptr(ci), _np_ptr(va), int(stretched)), "escoin_plan_get_csr")
        ng = np.array([self.nnz(g) for g in range(d.group)], np.int32)
        return rp, ci[:nnz], va[:nnz], ng

    def export_aligned(self):
        """The aligned form (CSR + channel deal + unit table + code object) as a numpy uint8 array."""
        n = C.c_size_t()
        check(lib().escoin_plan_export_aligned(self._h, None, 0, C.byref(n)), "escoin_plan_export_aligned")
        buf = np.zeros(n.value, np.uint8)
        check(lib().escoin_plan_export_aligned(self._h, _np_ptr(buf), buf.size, C.byref(n)),
              "escoin_plan_export_aligned")
        return buf[:n.value]

    def import_aligned(self, blob, stream=None):
        """Restore what export_aligned wrote; True when the persisted code object was loaded as it was.
        blob: numpy uint8 (host), or a torch CUDA uint8 tensor -- the buffer an RCCL broadcast filled -- which is
        handed over as it is (escoin_plan_import_aligned_dev: no .cpu().numpy() round trip)."""
        if not isinstance(blob, np.ndarray) and getattr(blob, "is_cuda", False):
            t = blob.contiguous()
            assert t.element_size() == 1
            check(lib().escoin_plan_import_aligned_dev(self._h, C.c_void_p(t.data_ptr()), t.numel(), stream),
                  "escoin_plan_import_aligned_dev")
            return self.stat("import_fast") == 1
        b = np.ascontiguousarray(blob, np.uint8)
        check(lib().escoin_plan_import_aligned(self._h, _np_ptr(b), b.size, stream), "escoin_plan_import_aligned")
        return self.stat("import_fast") == 1

    def stat(self, key):
        v = lib().escoin_plan_stat(self._h, key.encode())
        if v < 0:
            check(int(v), "escoin_plan_stat(%s)" % key)
        return int(v)

    @property
    def align_ms(self):
        return self.stat("align_us") * 1e-3

    @property
    def workspace_bytes(self):
        return int(lib().escoin_plan_workspace_bytes(self._h))

    @property
    def kernel_name(self):
        return lib().escoin_plan_kernel_name(self._h).decode()

    @property
    def tiling_info(self):
        """How the fast kernel tiles the layer (one line; "" for the generic / dense / lowered kernels)."""
        return lib().escoin_plan_tiling_info(self._h).decode()

    def forward_ptr(self, bottom_ptr, bias_ptr, top_ptr, n_images, stream=None):
        """Raw-pointer Forward_gpu (device pointers as ints)."""
        check(lib().escoin_forward(self._h, C.c_void_p(bottom_ptr),
                                   C.c_void_p(bias_ptr) if bias_ptr else None,
                                   C.c_void_p(top_ptr), int(n_images), stream), "escoin_forward")

    def forward(self, bottom, bias=None, top=None):
        """Forward_gpu on torch CUDA tensors (float32, or float64 for a double plan), on torch's current stream."""
        import torch
        d = self.desc
        dt = bottom.dtype
        assert bottom.is_cuda and dt in (torch.float32, torch.float64) and bottom.is_contiguous()
        n = bottom.shape[0]
        assert tuple(bottom.shape[1:]) == (d.C, d.H, d.W), "bottom shape mismatch"
        if top is None:
            top = torch.empty((n, d.M) + tuple(self.out_hw), device=bottom.device, dtype=dt)
        assert top.is_contiguous() and top.dtype == dt and tuple(top.shape) == (n, d.M) + tuple(self.out_hw)
        if bias is not None:
            assert bias.is_cuda and bias.dtype == dt and bias.numel() == d.M
        stream = C.c_void_p(torch.cuda.current_stream(bottom.device).cuda_stream)
        if dt == torch.float64:
            check(lib().escoin_forward_f64(self._h, C.c_void_p(bottom.data_ptr()),
                                           C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                           C.c_void_p(top.data_ptr()), int(n), stream), "escoin_forward_f64")
            return top
        self.forward_ptr(bottom.data_ptr(), bias.data_ptr() if bias is not None else 0,
                         top.data_ptr(), n, stream)
        return top
