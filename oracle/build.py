"""Build + ctypes binding of the oracle's C restatement (oracle/rans_ref.c).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Output goes to oracle/_build/ (git-ignored,
travels to the GPU box with the snapshot).

There is no `oracle/_ref` in this repo: the reference is 100 % Python and its only native code
is third-party (compressai 1.2.1's pybind11 extensions), whose sources are not under
/root/reference -- so there is no reference C/C++ to compile ("Oracle").
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rans_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle_rans.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", LIB, SRC, "-lm"])
    return LIB


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class _Lib:
    def __init__(self, path):
        L = ctypes.CDLL(path)
        P, LONG = ctypes.c_void_p, ctypes.c_long
        L.orc_rans_encode.restype = LONG
        L.orc_rans_encode.argtypes = [P, P, LONG, P, LONG, P, P, P, LONG, ctypes.POINTER(LONG)]
        L.orc_rans_dec_open.restype = P
        L.orc_rans_dec_open.argtypes = [ctypes.c_char_p, LONG]
        L.orc_rans_dec_close.argtypes = [P]
        L.orc_rans_dec_run.restype = LONG
        L.orc_rans_dec_run.argtypes = [P, P, LONG, P, LONG, P, P, P]
        L.orc_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.orc_pmf_to_quantized_cdf.argtypes = [P, ctypes.c_int, ctypes.c_int, P]
        self.L = L

    def encode(self, symbols, indexes, cdf, cdf_length, offset):
        s, i = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        c, cl, off = _i32(cdf), _i32(cdf_length), _i32(offset)
        cap = 2 * len(s) + 64
        out = np.zeros(cap, dtype=np.uint32)
        first = ctypes.c_long(0)
        nb = self.L.orc_rans_encode(s.ctypes.data, i.ctypes.data, len(s), c.ctypes.data, c.shape[1],
                                    cl.ctypes.data, off.ctypes.data, out.ctypes.data, cap, ctypes.byref(first))
        if nb < 0:
            raise RuntimeError("oracle rans encode overflow")
        return out[first.value:].astype("<u4").tobytes()

    def dec_open(self, data):
        return _Handle(self.L, self.L.orc_rans_dec_open(data, len(data)))

    def dec_run(self, h, indexes, cdf, cdf_length, offset):
        i = _i32(indexes).reshape(-1)
        c, cl, off = _i32(cdf), _i32(cdf_length), _i32(offset)
        out = np.empty(len(i), dtype=np.int32)
        self.L.orc_rans_dec_run(h.h, i.ctypes.data, len(i), c.ctypes.data, c.shape[1],
                                cl.ctypes.data, off.ctypes.data, out.ctypes.data)
        return out

    def pmf_to_quantized_cdf(self, pmf, precision=16):
        p = np.ascontiguousarray(pmf, dtype=np.float32)
        out = np.zeros(len(p) + 1, dtype=np.int32)
        rc = self.L.orc_pmf_to_quantized_cdf(p.ctypes.data, len(p), precision, out.ctypes.data)
        if rc:
            raise ValueError(f"pmf_to_quantized_cdf rc={rc}")
        return out


class _Handle:
    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        try:
            self.L.orc_rans_dec_close(self.h)
        except Exception:
            pass


def load():
    """Return the bound library, building it if gcc is available; None if that fails."""
    try:
        return _Lib(build())
    except Exception:
        return None
