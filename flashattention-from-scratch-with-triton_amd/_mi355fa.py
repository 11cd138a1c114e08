"""ctypes binding of libmi355fa.so (C ABI declared in include/mi355fa.h).

This is the only place the Python host touches native code.  There is NO fallback: if the
shared library is missing or does not export the expected ABI, importing this module
raises, and every caller fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355fa.so")

ABI_VERSION = 1
FP16, BF16 = 0, 1

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> (restype, argtypes); mirrors include/mi355fa.h one to one
SIGNATURES = {
    "fa_abi_version": (_i, []),
    "fa_last_error": (ctypes.c_char_p, []),
    "fa_supported": (_i, [_i, _i]),
    "fa_fwd": (_i, [_vp] * 5 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dq": (_i, [_vp] * 8 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dkv": (_i, [_vp] * 8 + [_i] * 7 + [_f, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmi355fa.so not found at %s -- build it first: "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s`"
            % (LIB_PATH, os.path.join(_HERE, "csrc")))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.fa_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libmi355fa.so ABI version %d, host expects %d" % (got, ABI_VERSION))
    return lib


lib = _load()


def check(rc, what):
    """Non-zero return code -> RuntimeError carrying fa_last_error()."""
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, lib.fa_last_error().decode()))
