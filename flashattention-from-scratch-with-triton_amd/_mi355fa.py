"""ctypes binding of libmi355fa.so (C ABI declared in include/mi355fa.h).

This is the only place the Python host touches native code.  There is NO fallback: if the
shared library is missing or does not export the expected ABI, importing this module
raises, and every caller fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355fa.so")

ABI_VERSION = 2
FP16, BF16 = 0, 1

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_sp = ctypes.POINTER(ctypes.c_longlong)   # const long long* strides (3 element strides) or NULL

# name -> (restype, argtypes); mirrors include/mi355fa.h one to one
SIGNATURES = {
    "fa_abi_version": (_i, []),
    "fa_last_error": (ctypes.c_char_p, []),
    "fa_supported": (_i, [_i, _i]),
    "fa_fwd": (_i, [_vp] * 5 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dq": (_i, [_vp] * 8 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dkv": (_i, [_vp] * 8 + [_i] * 7 + [_f, _vp]),
    # strided inputs: (ptr, strides) pairs for q, k, v (and dout), then the contiguous outputs
    "fa_fwd_strided": (_i, [_vp, _sp] * 3 + [_vp, _vp] + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dq_strided": (_i, [_vp, _sp] * 3 + [_vp] + [_vp, _sp] + [_vp] * 3 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dkv_strided": (_i, [_vp, _sp] * 4 + [_vp] * 4 + [_i] * 7 + [_f, _vp]),
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


def strides3(t):
    """ctypes array {batch, head, seq} of element strides for a [B, H, S, D] tensor, or None (= NULL, contiguous)
    when `t` is contiguous.  Callers must have checked `strided_ok(t)` first."""
    if t.is_contiguous():
        return None
    return (ctypes.c_longlong * 3)(t.stride(0), t.stride(1), t.stride(2))


def strided_ok(t):
    """True if the kernels can read `t` in place: unit head-dim stride, the other strides positive multiples of
    8 elements (16-byte rows), rows not overlapping, 16-byte aligned base (include/mi355fa.h, strided inputs)."""
    if t.is_contiguous():
        return True
    if t.stride(3) != 1 or t.data_ptr() % 16:
        return False
    if t.stride(2) < t.shape[3]:
        return False
    return all(t.shape[i] == 1 or (t.stride(i) >= 8 and t.stride(i) % 8 == 0) for i in range(3))


def check(rc, what):
    """Non-zero return code -> RuntimeError carrying fa_last_error()."""
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, lib.fa_last_error().decode()))
