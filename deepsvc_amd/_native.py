"""ctypes binding of libdsvc.so (the C ABI declared in include/dsvc.h).

There is NO fallback: if the library is missing this module raises, and every op in
`deepsvc_amd.ops` goes through it.  Build with `python -m deepsvc_amd.build`.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSVC_LIB") or os.path.join(HERE, "lib", "libdsvc.so")     # DSVC_LIB: an instrumented build (tools/phase_probe.py)

ABI_VERSION = 3            # include/dsvc.h: DSVC_VERSION this binding was written against
F32, F16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_SIGMOID, ACT_HALF_TANH = range(6)
EW_COPY, EW_ADD, EW_SUB, EW_BLEND2, EW_BLEND3, EW_CLAMP01 = range(6)
EB_PARAMS_PER_CH = 59


class Seg(Structure):
    _fields_ = [("ptr", c_void_p), ("C", c_int32), ("ld", c_int32), ("dtype", c_int32)]


class ConvDesc(Structure):
    _fields_ = [
        ("B", c_int32), ("H", c_int32), ("W", c_int32), ("nseg", c_int32),
        ("seg", Seg * 3),
        ("kh", c_int32), ("kw", c_int32), ("stride", c_int32), ("pad", c_int32),
        ("Cout", c_int32), ("compute", c_int32),
        ("weight", c_void_p), ("bias", c_void_p),
        ("pre_act", c_int32), ("act", c_int32), ("slope", c_float),
        ("res1", c_void_p), ("res1_ld", c_int32), ("res1_dtype", c_int32),
        ("res1_scale", c_void_p),
        ("res2", c_void_p), ("res2_ld", c_int32), ("res2_dtype", c_int32),
        ("shuffle", c_int32),
        ("out", c_void_p), ("out_ld", c_int32), ("out_dtype", c_int32),
        ("s2d", c_int32),
        ("post_weight", c_void_p), ("post_bias", c_void_p),
        ("batch_inner", c_int32), ("weight_bstride", c_int64), ("bias_bstride", c_int64), ("in_bstride", (c_int64 * 2) * 3),
        ("out_bstride", c_int64), ("res1_bstride", c_int64),
    ]


class ScanGc(Structure):
    _fields_ = [("y", c_void_p), ("mu", c_void_p), ("scale", c_void_p),
                ("y_ld", c_int32), ("mu_ld", c_int32), ("scale_ld", c_int32), ("dtype", c_int32),
                ("table", c_void_p), ("levels", c_int32),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32),
                ("symbols", c_void_p), ("indexes", c_void_p),
                ("y_hat", c_void_p), ("y_hat_ld", c_int32)]


class ScanEb(Structure):
    _fields_ = [("z", c_void_p), ("z_ld", c_int32), ("dtype", c_int32),
                ("params", c_void_p),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("mode", c_int32),
                ("symbols", c_void_p),
                ("z_hat", c_void_p), ("z_hat_ld", c_int32),
                ("lik", c_void_p), ("lik_ld", c_int32)]


class _Operand(Structure):
    _fields_ = [("ptr", c_void_p), ("ld", c_int32), ("dtype", c_int32)]


class ScanEw(Structure):
    _fields_ = [("op", c_int32), ("C", c_int32), ("n_pix", c_int64),
                ("a", _Operand), ("b", _Operand), ("c", _Operand), ("d", _Operand), ("e", _Operand),
                ("out", c_void_p), ("out_ld", c_int32), ("out_dtype", c_int32)]


class ScanCall(Structure):
    _fields_ = [("p", c_void_p * 6), ("i", c_int32 * 10), ("f", c_float * 2)]


class _ScanU(ctypes.Union):
    _fields_ = [("conv", ConvDesc), ("gc", ScanGc), ("eb", ScanEb), ("ew", ScanEw), ("call", ScanCall)]


class ScanOp(Structure):
    _fields_ = [("kind", c_int32), ("reserved", c_int32), ("u", _ScanU)]


SCAN_CONV, SCAN_GC_QUANTIZE, SCAN_GC_DEQUANTIZE, SCAN_EB_PROCESS, SCAN_EB_DEQUANTIZE, SCAN_ELTWISE = 0, 1, 2, 3, 4, 5
SCAN_WARP, SCAN_UPSAMPLE2X, SCAN_POOL2X2, SCAN_PIXEL_SHUFFLE, SCAN_NCHW_TO_NHWC, SCAN_NHWC_TO_NCHW, SCAN_SE_GATE = 6, 7, 8, 9, 10, 11, 12
SCAN_UNION_MEMBER = {SCAN_CONV: "conv", SCAN_GC_QUANTIZE: "gc", SCAN_GC_DEQUANTIZE: "gc", SCAN_EB_PROCESS: "eb", SCAN_EB_DEQUANTIZE: "eb",
                     SCAN_ELTWISE: "ew", SCAN_WARP: "call", SCAN_UPSAMPLE2X: "call", SCAN_POOL2X2: "call", SCAN_PIXEL_SHUFFLE: "call",
                     SCAN_NCHW_TO_NHWC: "call", SCAN_NHWC_TO_NCHW: "call", SCAN_SE_GATE: "call"}


def pointer_offsets(ctype, base=0):
    """Byte offsets of every pointer-typed field inside a ctypes Structure / array (recursively)."""
    out = []
    if isinstance(ctype, type) and issubclass(ctype, ctypes.Array):
        n, et = ctype._length_, ctype._type_
        for k in range(n):
            out += pointer_offsets(et, base + k * ctypes.sizeof(et))
    elif isinstance(ctype, type) and issubclass(ctype, Structure):
        for name, ft in ctype._fields_:
            out += pointer_offsets(ft, base + getattr(ctype, name).offset)
    elif ctype in (c_void_p, c_char_p):
        out.append(base)
    return out


class Operand(Structure):
    _fields_ = [("ptr", c_void_p), ("ld", c_int32), ("dtype", c_int32)]


class CdfTable(Structure):
    _fields_ = [("cdf", c_void_p), ("rows", c_int32), ("stride", c_int32), ("cdf_length", c_void_p), ("offset", c_void_p),
                ("lut", c_void_p)]


# name -> (restype, argtypes); every symbol of include/dsvc.h is listed (tests check this against the header)
SIGNATURES = {
    "dsvc_last_error": (c_char_p, []),
    "dsvc_version": (c_int, []),
    "dsvc_device_count": (c_int, []),
    "dsvc_set_option": (c_int, [c_char_p, c_int]),
    "dsvc_get_counter": (c_int64, [c_char_p]),
    "dsvc_profile_begin": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "dsvc_profile_end": (c_int64, [c_void_p, c_void_p, c_int64]),
    "dsvc_conv_packed_bytes": (c_int64, [c_int32, c_int32, POINTER(c_int32), c_int32, c_int32, c_int32]),
    "dsvc_conv_pack_weights": (c_int, [c_void_p, c_int32, c_int32, POINTER(c_int32), c_int32, c_int32, c_int32, c_void_p]),
    "dsvc_deconv5x5s2_to_conv3x3": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "dsvc_conv2d": (c_int, [c_void_p, POINTER(ConvDesc)]),
    "dsvc_entropy_scan_step": (c_int, [c_void_p, c_void_p, c_int32]),
    "dsvc_pframe_segment_run": (c_int, [c_void_p, c_void_p, c_int32]),
    "dsvc_pframe_program_load": (c_int, [c_void_p, c_int64, c_int32, POINTER(c_void_p)]),
    "dsvc_pframe_program_destroy": (None, [c_void_p]),
    "dsvc_pframe_program_info": (c_int, [c_void_p, c_void_p]),
    "dsvc_pframe_compress": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsvc_pframe_decompress": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsvc_warp_bilinear_border": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                          c_int32, c_int32, c_int32, c_int32]),
    "dsvc_upsample2x_bilinear": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float]),
    "dsvc_pool2x2": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "dsvc_pixel_shuffle_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "dsvc_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "dsvc_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32]),
    "dsvc_eltwise": (c_int, [c_void_p, c_int32, c_int64, c_int32, POINTER(Operand), POINTER(Operand), POINTER(Operand),
                             POINTER(Operand), POINTER(Operand), c_void_p, c_int32, c_int32]),
    "dsvc_se_workspace_floats": (c_int64, [c_int32, c_int32]),
    "dsvc_se_gate": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                             c_void_p, c_void_p]),
    "dsvc_refine_aggregate": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32,
                                      c_void_p, c_int32]),
    "dsvc_reduce_workspace_floats": (c_int64, []),
    "dsvc_reduce": (c_int, [c_void_p, c_int32, c_int64, c_int32, POINTER(Operand), POINTER(Operand), c_void_p, c_void_p]),
    "dsvc_ssim_workspace_floats": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "dsvc_ssim_level": (c_int, [c_void_p, POINTER(Operand), POINTER(Operand), c_int32, c_int32, c_int32, c_int32, c_float,
                                c_void_p, c_void_p, c_void_p]),
    "dsvc_avgpool2_pad": (c_int, [c_void_p, POINTER(Operand), c_void_p, c_int32, c_int32, c_int32, c_int32]),
    "dsvc_gc_quantize": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int32,
                                 c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32]),
    "dsvc_gc_dequantize": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32]),
    "dsvc_gc_likelihood": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int64, c_int32,
                                   c_void_p, c_int32, c_void_p, c_int32]),
    "dsvc_eb_process": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                c_void_p, c_void_p, c_int32, c_void_p, c_int32]),
    "dsvc_eb_dequantize": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32]),
    "dsvc_pmf_to_quantized_cdf": (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "dsvc_cdf_build_lut": (c_int, [POINTER(CdfTable), c_void_p]),
    "dsvc_cdf_forget_lut": (c_int, [c_void_p]),
    "dsvc_rans_ms_slot_words": (c_int64, [c_int32]),
    "dsvc_rans_ms_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, POINTER(CdfTable), c_void_p, c_int32, c_void_p, c_void_p]),
    "dsvc_rans_ms_decode": (c_int, [c_void_p, c_void_p, c_int64, c_int32, POINTER(CdfTable), c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsvc_rans_encoder_create": (c_int, [POINTER(c_void_p)]),
    "dsvc_rans_encoder_destroy": (None, [c_void_p]),
    "dsvc_rans_encode_with_indexes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(CdfTable)]),
    "dsvc_rans_flush": (c_int64, [c_void_p, c_void_p, c_int64]),
    "dsvc_rans_decoder_create": (c_int, [POINTER(c_void_p), c_char_p, c_int64]),
    "dsvc_rans_decoder_destroy": (None, [c_void_p]),
    "dsvc_rans_decode_stream": (c_int, [c_void_p, c_void_p, c_int64, POINTER(CdfTable), c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: build it with `python -m deepsvc_amd.build` "
                              "(the HIP extension is mandatory; there is no fallback path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.dsvc_version() != ABI_VERSION:      # a stale build would silently reinterpret dtype codes / struct layouts
            raise NativeError(f"{LIB_PATH} reports ABI version {L.dsvc_version()}, this binding needs {ABI_VERSION}: "
                              "rebuild it with `python -m deepsvc_amd.build --force`")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc is not None and rc < 0:
        msg = lib().dsvc_last_error()
        raise NativeError(f"{what} failed (rc={rc}): {msg.decode(errors='replace') if msg else ''}")
    return rc
