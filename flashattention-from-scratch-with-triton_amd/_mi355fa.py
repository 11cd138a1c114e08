"""ctypes binding of libmi355fa.so (C ABI declared in include/mi355fa.h).

This is the only place the Python host touches native code.  There is NO fallback: if the
shared library is missing or does not export the expected ABI, importing this module
raises, and every caller fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355fa.so")

ABI_VERSION = 7
FP16, BF16 = 0, 1

_vp, _i, _f, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_ulonglong
_sp = ctypes.POINTER(ctypes.c_longlong)   # const long long* strides (3 element strides) or NULL

class Opts(ctypes.Structure):
    """mi355fa_opts (include/mi355fa.h): strides, cu_seqlens and dropout in any combination for the fa_*_ex entry points.
    `Opts.make(...)` zero-initialises and sets `size`."""
    _fields_ = ([("size", ctypes.c_uint)] +
                [(n, _sp) for n in ("q_strides", "k_strides", "v_strides", "o_strides", "dout_strides", "dq_strides",
                                    "dk_strides", "dv_strides")] +
                [("cu_seqlens_q", _vp), ("cu_seqlens_k", _vp), ("total_q", _i), ("total_k", _i),
                 ("p_drop", _f), ("seed", _u64), ("offset", _u64), ("q_scaled", _vp)])

    @classmethod
    def make(cls, **kw):
        o = cls()
        o.size = ctypes.sizeof(cls)
        for k, v in kw.items():
            setattr(o, k, v)
        return o


_op = ctypes.POINTER(Opts)

# name -> (restype, argtypes); mirrors include/mi355fa.h one to one
SIGNATURES = {
    "fa_abi_version": (_i, []),
    "fa_last_error": (ctypes.c_char_p, []),
    "fa_supported": (_i, [_i, _i]),
    "fa_fwd": (_i, [_vp] * 5 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dq": (_i, [_vp] * 8 + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dkv": (_i, [_vp] * 8 + [_i] * 7 + [_f, _vp]),
    # strided tensors: (ptr, strides) pairs for q, k, v, (o), (dout) and for the outputs o / dq / dk / dv
    "fa_fwd_strided": (_i, [_vp, _sp] * 4 + [_vp] + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dq_strided": (_i, [_vp, _sp] * 5 + [_vp] + [_vp, _sp] + [_vp] + [_i] * 7 + [_f, _vp]),
    "fa_bwd_dkv_strided": (_i, [_vp, _sp] * 4 + [_vp] * 2 + [_vp, _sp] * 2 + [_i] * 7 + [_f, _vp]),
    # variable-length: packed [total, H, D] tensors, then cu_seqlens_q, cu_seqlens_k (device int32), then
    # batch, H, total_q, total_k, max_seqlen_q, max_seqlen_k, D, dtype, causal, scale, stream
    "fa_fwd_varlen": (_i, [_vp] * 5 + [_vp] * 2 + [_i] * 9 + [_f, _vp]),
    "fa_bwd_dq_varlen": (_i, [_vp] * 8 + [_vp] * 2 + [_i] * 9 + [_f, _vp]),
    "fa_bwd_dkv_varlen": (_i, [_vp] * 8 + [_vp] * 2 + [_i] * 9 + [_f, _vp]),
    # attention dropout: the fixed-length signatures + p_drop (float), seed, offset (unsigned long long), stream
    "fa_dropout_keep_scale": (_f, [_f]),
    "fa_fwd_dropout": (_i, [_vp] * 5 + [_i] * 7 + [_f, _f, _u64, _u64, _vp]),
    "fa_bwd_dq_dropout": (_i, [_vp] * 8 + [_i] * 7 + [_f, _f, _u64, _u64, _vp]),
    "fa_bwd_dkv_dropout": (_i, [_vp] * 8 + [_i] * 7 + [_f, _f, _u64, _u64, _vp]),
    # general form: the plain signatures + const mi355fa_opts* (NULL = plain launch), stream
    "fa_fwd_ex": (_i, [_vp] * 5 + [_i] * 7 + [_f, _op, _vp]),
    "fa_bwd_dq_ex": (_i, [_vp] * 8 + [_i] * 7 + [_f, _op, _vp]),
    "fa_bwd_dkv_ex": (_i, [_vp] * 8 + [_i] * 7 + [_f, _op, _vp]),
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
    # a size-1 dimension may carry any stride in PyTorch: it is never multiplied by a non-zero index
    return (ctypes.c_longlong * 3)(*(t.stride(i) if t.shape[i] > 1 else 0 for i in range(2)),
                                   t.stride(2) if t.shape[2] > 1 else t.shape[3])


def strided_ok(t):
    """True if the kernels can read `t` in place: 16-byte aligned base, and either contiguous or unit head-dim stride
    with the other strides multiples of 8 elements (16-byte rows; batch / head may be 0), rows not overlapping and one
    (batch, head) slice within 2^31 bytes (include/mi355fa.h).  Anything else is copied by the caller."""
    if t.data_ptr() % 16:
        return False
    if t.is_contiguous():
        return True
    if t.stride(3) != 1:
        return False
    if t.stride(2) < t.shape[3]:
        return False
    if t.shape[2] > 1 and t.stride(2) % 8:
        return False
    if (t.shape[2] - 1) * t.stride(2) * 2 + 2 * t.shape[3] > (1 << 31) - 1:   # 32-bit buffer offsets inside a slice
        return False
    # batch / head strides may be 0: an expanded K/V shared by several heads is read in place
    return all(t.shape[i] == 1 or (t.stride(i) >= 0 and t.stride(i) % 8 == 0) for i in range(2))


def check(rc, what):
    """Non-zero return code -> RuntimeError carrying fa_last_error()."""
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, lib.fa_last_error().decode()))
