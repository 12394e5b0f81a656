"""ctypes binding of libmmhip.so.  Prototypes are generated from include/mm_hip.h so the Python side cannot
drift from the C ABI.  There is NO fallback: if the library is missing or a call fails, we raise."""
from __future__ import annotations

import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_PKG, "..", "include", "mm_hip.h")
LIB_PATH = os.environ.get("MM_HIP_LIBRARY") or os.path.join(_PKG, "libmmhip.so")   # override: kernel experiments only

MM_BF16, MM_F32 = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_GELU_ERF, EPI_QUICK_GELU, EPI_RESIDUAL, EPI_ACCUMULATE, EPI_GELU_TANH = 1, 2, 4, 8, 16, 32
GELU_KIND = {EPI_GELU_ERF: 0, EPI_QUICK_GELU: 1, EPI_GELU_TANH: 2}      # mm_gelu_fwd/bwd `kind` of each epilogue flag


class MMHipError(RuntimeError):
    pass


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|const\s+char\s*\*)\s+(mm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                elif re.match(r"^(const\s+)?int64_t\b", a):
                    argtypes.append(ctypes.c_int64)
                elif re.match(r"^(const\s+)?float\b", a):
                    argtypes.append(ctypes.c_float)
                elif re.match(r"^(const\s+)?int\b", a):
                    argtypes.append(ctypes.c_int)
                else:
                    raise ValueError(f"unhandled C type in header: {a!r} ({name})")
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes)
    return protos


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMHipError(
                f"{LIB_PATH} not found: build it with `python multimeditron_amd/csrc/build.py` "
                "(or __graft_entry__.build()).  multimeditron_amd has no CPU / eager fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in parse_header().items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def call(name: str, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        msg = L.mm_error_string(rc)
        raise MMHipError(f"{name} failed: {rc} ({msg.decode() if msg else '?'})")


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    call("mm_get_option", name.encode(), ctypes.byref(v))
    return v.value
