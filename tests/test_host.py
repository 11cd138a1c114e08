"""CPU tests of the drop-in boundary: the C-ABI library loads and exports what include/mi355fa.h
declares, rejects bad arguments before launching, and the Python host mirrors the reference's
call surface (names, argument order, defaults).  No compute is launched here (no GPU)."""
import ast
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import PKG, ROOT


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "mi355fa.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z_]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    assert _header_functions() == ["fa_abi_version", "fa_bwd_dkv", "fa_bwd_dkv_dropout", "fa_bwd_dkv_ex", "fa_bwd_dkv_strided",
                                   "fa_bwd_dkv_varlen", "fa_bwd_dq", "fa_bwd_dq_dropout", "fa_bwd_dq_ex", "fa_bwd_dq_strided",
                                   "fa_bwd_dq_varlen", "fa_dropout_keep_scale", "fa_fwd", "fa_fwd_dropout", "fa_fwd_ex",
                                   "fa_fwd_strided", "fa_fwd_varlen", "fa_last_error", "fa_supported"]


def test_library_exports_every_declared_symbol():
    import _mi355fa as fa
    raw = ctypes.CDLL(fa.LIB_PATH)
    for name in _header_functions():
        assert hasattr(raw, name), name
        assert name in fa.SIGNATURES, "python binding misses " + name
    assert fa.lib.fa_abi_version() == 7
    assert fa.lib.fa_supported(64, fa.BF16) == 1 and fa.lib.fa_supported(128, fa.FP16) == 1
    assert fa.lib.fa_supported(96, fa.BF16) == 0 and fa.lib.fa_supported(64, 7) == 0


def test_torch_binding_loads_and_matches_the_abi():
    import _mi355fa as fa
    import _mi355fa_torch as ext
    assert ext.abi_version() == fa.ABI_VERSION == 7
    for name in ("flash_attention", "forward_launch", "backward_launch", "flash_attention_varlen", "varlen_forward_launch",
                 "varlen_backward_launch", "flash_attention_dropout", "dropout_forward_launch", "dropout_backward_launch"):
        assert callable(getattr(ext, name)), name
    q = torch.randn(1, 1, 8, 64, dtype=torch.float16)
    with pytest.raises(AssertionError, match="device tensors"):       # M:133, before any allocation or launch
        ext.flash_attention(q, q, q, True)
    with pytest.raises(AssertionError, match="same shape"):
        ext.forward_launch(q, q, q[:, :, :4], False)


def test_argument_errors_are_rejected_before_launch():
    import _mi355fa as fa
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 15) & ~15
    L = fa.lib
    assert L.fa_fwd(None, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, None) == -1
    assert b"NULL" in L.fa_last_error()
    assert L.fa_fwd(p, p, p, p, p, 0, 1, 8, 8, 64, 1, 0, 0.125, None) == -2
    assert L.fa_fwd(p, p, p, p, p, 1, 1, 8, 8, 96, 1, 0, 0.125, None) == -3
    assert b"head dim" in L.fa_last_error()
    assert L.fa_fwd(p, p, p, p, p, 1, 1, 8, 8, 64, 5, 0, 0.125, None) == -4
    assert L.fa_fwd(p + 2, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, None) == -5
    assert L.fa_bwd_dq(p, p, p, p, p, p, p, None, 1, 1, 8, 8, 64, 1, 0, 0.125, None) == -1
    assert L.fa_bwd_dkv(p, p, p, p, p, p, p, p, 1, 1, 8, 8, 65, 1, 0, 0.125, None) == -3
    with pytest.raises(RuntimeError, match="head dim"):
        fa.check(-3, "fa_fwd")
    # strided entry points: element strides {batch, head, seq}; NULL = contiguous
    S3 = ctypes.c_longlong * 3
    ok = S3(8 * 2 * 64, 64, 2 * 64)          # a [B=1, S=8, H=2, D=64] buffer seen as [B, H, S, D]
    args = (1, 2, 8, 8, 64, 1, 0, 0.125, None)
    assert L.fa_fwd_strided(None, ok, p, ok, p, ok, p, None, p, *args) == -1
    assert L.fa_fwd_strided(p, S3(1024, 64, 100), p, ok, p, ok, p, None, p, *args) == -6      # 100 is not a multiple of 8
    assert b"multiples of 8" in L.fa_last_error()
    assert L.fa_fwd_strided(p, ok, p, ok, p, S3(1024, 64, 256), p, None, p, *args) == -6       # K and V row strides differ
    assert b"K and V" in L.fa_last_error()
    assert L.fa_fwd_strided(p, S3(1024, 64, 32), p, ok, p, ok, p, None, p, *args) == -6        # rows would overlap
    assert L.fa_bwd_dq_strided(p, ok, p, ok, p, ok, p, None, p, S3(-8, 64, 128), p, p, None, p, *args) == -6   # negative stride
    assert L.fa_bwd_dkv_strided(p, ok, p, ok, p, ok, p, ok, p, p, p, None, None, None, *args) == -1
    # outputs take strides too (ABI 5), but never a broadcast one: two heads would write the same rows
    assert L.fa_fwd_strided(p, ok, p, ok, p, ok, p, S3(1024, 0, 128), p, *args) == -6
    assert b"output" in L.fa_last_error()
    assert L.fa_bwd_dq_strided(p, ok, p, ok, p, ok, p, S3(1024, 0, 128), p, ok, p, p, S3(1024, 0, 128), p, *args) == -6
    assert L.fa_bwd_dkv_strided(p, ok, p, ok, p, ok, p, ok, p, p, p, ok, p, S3(1024, 64, 100), *args) == -6


def test_general_entry_points_validate_their_options():
    """fa_*_ex (include/mi355fa.h): the options struct is size-checked, cu_seqlens come in pairs and exclude strides, the
    dropout probability is range-checked -- all before anything is launched (no GPU needed)."""
    import _mi355fa as fa
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 15) & ~15
    L = fa.lib
    args = (1, 2, 8, 8, 64, 1, 0, 0.125)
    bad = fa.Opts.make()
    bad.size = 4
    assert L.fa_fwd_ex(p, p, p, p, p, *args, ctypes.byref(bad), None) == -2
    assert b"size" in L.fa_last_error()
    one = fa.Opts.make(cu_seqlens_q=p)
    assert L.fa_fwd_ex(p, p, p, p, p, *args, ctypes.byref(one), None) == -1
    S3 = ctypes.c_longlong * 3
    both = fa.Opts.make(cu_seqlens_q=p, cu_seqlens_k=p, total_q=8, total_k=8, q_strides=S3(1024, 64, 128))
    assert L.fa_bwd_dq_ex(p, p, p, p, p, p, p, p, *args, ctypes.byref(both), None) == -6
    assert b"packed" in L.fa_last_error()
    assert L.fa_bwd_dkv_ex(p, p, p, p, p, p, p, p, *args, ctypes.byref(fa.Opts.make(p_drop=1.0)), None) == -2
    assert b"dropout" in L.fa_last_error()
    assert L.fa_fwd_ex(None, p, p, p, p, *args, None, None) == -1                 # opts == NULL is the plain launch
    assert ctypes.sizeof(fa.Opts) == 128                                          # layout of mi355fa_opts on LP64


def test_schedule_table_lookup_and_override():
    """The generated shape -> schedule-family table (csrc/fa_table.h, tools/tune.py; the counterpart of the reference's
    autotune key (S_q, S_k, D, is_causal), K:18-32) answers for every shape, and fa_debug_force_impl overrides it."""
    import _mi355fa as fa
    raw = ctypes.CDLL(fa.LIB_PATH)
    pick, force = raw.fa_debug_pick, raw.fa_debug_force_impl
    pick.argtypes = [ctypes.c_int] * 8
    force.argtypes = [ctypes.c_int] * 3
    allowed = {0: {64: {1, 2, 3, 4}, 128: {1, 4}}, 1: {64: {1, 2, 3, 4}, 128: {1}}, 2: {64: {1, 2, 3, 4}, 128: {1, 2}}}
    for kernel in range(3):
        for D in (64, 128):
            for dtype in (0, 1):
                for causal in (0, 1):
                    for (B, H, S) in ((1, 1, 1), (4, 8, 512), (4, 32, 4096), (64, 32, 8192), (1, 2, 100000)):
                        assert pick(kernel, D, dtype, causal, B, H, S, S) in allowed[kernel][D]
    try:
        force(2, 3, 1)
        assert pick(0, 64, 0, 0, 4, 32, 4096, 4096) == 2 and pick(0, 128, 0, 0, 4, 32, 4096, 4096) == 1  # forced family 2
        assert pick(1, 64, 1, 1, 4, 32, 4096, 4096) == 3 and pick(1, 128, 1, 1, 4, 32, 4096, 4096) == 1
        assert pick(2, 64, 1, 1, 4, 32, 4096, 4096) == 1
        force(4, 0, 3)   # round 3: the one-wave-per-SIMD families; a shape one of them cannot take falls back
        assert pick(0, 64, 1, 1, 4, 32, 4096, 4096) == 4 and pick(0, 128, 0, 0, 4, 32, 500, 333) == 4
        assert pick(0, 64, 1, 1, 4, 32, 500, 500) == 1        # causal with a ragged last query tile: family 1
        assert pick(2, 64, 1, 1, 4, 32, 4096, 4096) == 3 and pick(2, 128, 1, 1, 4, 32, 4096, 4096) == 2
        force(0, 4, 0)   # round 4: the one-wave-per-SIMD dQ; whole 128-key tiles, causal: whole 256-row tiles that S_k covers
        assert pick(1, 64, 1, 1, 4, 32, 4096, 4096) == 4 and pick(1, 64, 0, 0, 4, 32, 300, 1024) == 4
        assert pick(1, 64, 1, 0, 4, 32, 512, 500) == 3 and pick(1, 64, 1, 1, 4, 32, 512, 384) == 3   # ragged keys / uncovered tile
        assert pick(1, 64, 1, 1, 1, 2, 200, 512) == 4 and pick(1, 128, 1, 1, 4, 32, 4096, 4096) == 1
        force(0, 0, 4)   # round 4: the dK/dV kernel on the pinned accumulator file; whole tiles, causal: bf16 with S_q >= S_k
        assert pick(2, 64, 1, 1, 4, 32, 4096, 4096) == 4 and pick(2, 64, 0, 0, 4, 32, 1024, 512) == 4
        assert pick(2, 64, 0, 1, 4, 32, 4096, 4096) == 3 and pick(2, 64, 1, 1, 4, 32, 512, 1024) == 3   # fp16 causal / S_q < S_k
        assert pick(2, 64, 1, 0, 4, 32, 500, 512) == 3 and pick(2, 128, 1, 1, 4, 32, 4096, 4096) == 2
    finally:
        force(0, 0, 0)
    hdr = open(os.path.join(PKG, "csrc", "fa_table.h")).read()
    assert hdr.startswith("// GENERATED by tools/tune.py") and "kFamily[3][2][2][2]" in hdr


def test_dropout_arguments_and_keep_scale():
    import _mi355fa as fa
    import My_FlashAttention_optimized as M
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 15) & ~15
    L = fa.lib
    assert L.fa_fwd_dropout(p, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, 1.0, 1, 0, None) == -2 and b"[0, 1)" in L.fa_last_error()
    assert L.fa_fwd_dropout(p, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, -0.1, 1, 0, None) == -2
    assert L.fa_fwd_dropout(None, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, 0.1, 1, 0, None) == -1
    # p > 0 that quantises to 0 is an error, never a silent "no dropout"; offsets need one 32-bit counter word
    assert L.fa_fwd_dropout(p, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, 0.001, 1, 0, None) == -2 and b"1/512" in L.fa_last_error()
    assert L.fa_fwd_dropout(p, p, p, p, p, 1, 1, 8, 8, 64, 1, 0, 0.125, 0.1, 1, 1 << 32, None) == -2 and b"2^32" in L.fa_last_error()
    assert L.fa_bwd_dq_dropout(p, p, p, p, p, p, p, p, 1, 1, 8, 8, 96, 1, 0, 0.125, 0.1, 1, 0, None) == -3
    assert L.fa_bwd_dkv_dropout(p, p, p, p, p, p, p, p + 2, 1, 1, 8, 8, 64, 1, 0, 0.125, 0.1, 1, 0, None) == -5
    assert abs(L.fa_dropout_keep_scale(0.25) - 256.0 / 192.0) < 1e-6 and L.fa_dropout_keep_scale(0.0) == 1.0
    assert abs(L.fa_dropout_keep_scale(0.1) - 256.0 / (256 - 26)) < 1e-6        # p quantised to 26 / 256
    q = torch.randn(1, 1, 8, 64, dtype=torch.float16)
    with pytest.raises(AssertionError, match="device tensors"):
        M.flash_attention_dropout(q, q, q, True, 0.1, seed=3)
    sig = [(p_.name, p_.default) for p_ in inspect.signature(M.flash_attention_dropout).parameters.values()]
    assert sig == [("Q", inspect.Parameter.empty), ("K", inspect.Parameter.empty), ("V", inspect.Parameter.empty),
                   ("is_causal", False), ("dropout_p", 0.0), ("seed", 0), ("offset", 0)]


def test_varlen_argument_errors_are_rejected_before_launch():
    import _mi355fa as fa
    import My_FlashAttention_optimized as M
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 15) & ~15
    L = fa.lib
    tail = (2, 4, 100, 100, 64, 64, 64, 1, 0, 0.125, None)   # batch, H, total_q, total_k, max_q, max_k, D, dtype, causal, scale, stream
    assert L.fa_fwd_varlen(p, p, p, p, p, None, p, *tail) == -1 and b"cu_seqlens" in L.fa_last_error()
    assert L.fa_fwd_varlen(None, p, p, p, p, p, p, *tail) == -1
    assert L.fa_fwd_varlen(p, p, p, p, p, p, p, 2, 4, 0, 100, 64, 64, 64, 1, 0, 0.125, None) == -2      # no query tokens
    assert L.fa_fwd_varlen(p, p, p, p, p, p, p, 2, 4, 100, 100, 0, 64, 64, 1, 0, 0.125, None) == -2     # max_seqlen_q < 1
    # (max_seqlen above the token count is a legal static bound: ADVICE r2 -- only the 2^31-byte limit applies)
    assert L.fa_fwd_varlen(p, p, p, p, p, p, p, 2, 4, 100, 100, 1 << 23, 64, 64, 1, 0, 0.125, None) == -2
    assert L.fa_fwd_varlen(p, p, p, p, p, p, p, 2, 4, 100, 100, 64, 64, 96, 1, 0, 0.125, None) == -3     # head dim
    assert L.fa_bwd_dq_varlen(p, p, p, p, p, p, p, p, p, p + 4, 2, 4, 100, 100, 64, 64, 64, 7, 0, 0.125, None) == -4
    assert L.fa_bwd_dkv_varlen(p, p, p, p, p, p, p, p + 2, p, p, *tail) == -5
    q = torch.randn(10, 2, 64, dtype=torch.float16)
    cu = torch.tensor([0, 4, 10], dtype=torch.int32)
    with pytest.raises(AssertionError, match="device tensors"):
        M.flash_attention_varlen(q, q, q, cu, cu, 6, 6, True)
    sig = [(p_.name, p_.default) for p_ in inspect.signature(M.flash_attention_varlen).parameters.values()]
    assert [n for n, _ in sig] == ["Q", "K", "V", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "is_causal",
                                   "dropout_p", "seed", "offset"]
    assert sig[7][1] is False and sig[8][1] == 0.0


def test_strided_ok_accepts_bshd_views_and_rejects_the_rest():
    import _mi355fa as fa
    x = torch.zeros(2, 16, 4, 64, dtype=torch.float16)            # [B, S, H, D]
    v = x.transpose(1, 2)                                          # [B, H, S, D] view
    assert not v.is_contiguous() and fa.strided_ok(v)
    assert list(fa.strides3(v)) == [16 * 4 * 64, 64, 4 * 64]
    assert fa.strides3(x) is None and fa.strided_ok(x)
    assert not fa.strided_ok(torch.zeros(2, 4, 64, 16, dtype=torch.float16).transpose(2, 3))   # head dim not unit stride
    assert not fa.strided_ok(torch.zeros(2, 4, 16, 68, dtype=torch.float16)[..., :64][..., 2:])  # odd row pitch / offset
    qkv = torch.zeros(2, 16, 3, 4, 64, dtype=torch.float16)       # fused projection output [B, S, 3, H, D]
    q, k, vv = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    assert all(fa.strided_ok(t) for t in (q, k, vv)) and k.stride(2) == vv.stride(2)
    kv = torch.zeros(2, 1, 16, 64, dtype=torch.float16).expand(2, 4, 16, 64)     # one K/V head shared by four query heads
    assert fa.strided_ok(kv) and list(fa.strides3(kv)) == [16 * 64, 0, 64]


def test_python_surface_matches_reference():
    import My_FlashAttention_optimized as M
    import _verify_func as V
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    E = inspect.Parameter.empty
    assert sig(M.flash_attention) == [("Q", E), ("K", E), ("V", E), ("is_causal", False)]          # M:169
    assert sig(M.flash_attention_forward) == [("Q", E), ("K", E), ("V", E), ("is_causal", E)]      # M:14
    assert [n for n, _ in sig(M.flash_attention_backward)] == ["Q", "K", "V", "O", "dO", "LSE", "is_causal"]  # M:62
    assert [n for n, _ in sig(M.compare_with_sdpa)][:4] == ["Q", "K", "V", "is_causal"]           # M:172
    assert issubclass(M.FlashAttentionFunction, torch.autograd.Function)
    assert [n for n, _ in sig(M.FlashAttentionFunction.forward)] == ["ctx", "Q", "K", "V", "is_causal"]
    assert [n for n, _ in sig(M.FlashAttentionFunction.backward)] == ["ctx", "dO"]
    assert sig(V.verify_results)[:5] == [("bench", E), ("triton_output", E), ("name", "Attention"),
                                         ("rtol", 1e-2), ("atol", 1e-3)]                           # V:3


def test_binding_asserts_like_reference_on_bad_inputs():
    import My_FlashAttention_optimized as M
    q = torch.randn(1, 1, 8, 64, dtype=torch.float16)
    with pytest.raises(AssertionError):  # M:133 is_cuda
        M.flash_attention(q, q, q)


def test_kv_shape_mismatch_is_rejected_before_any_launch():
    # the kernels take B and H from Q; a K / V with fewer heads or batches would be read out of bounds
    import My_FlashAttention_optimized as M
    q = torch.randn(2, 4, 16, 64, dtype=torch.float16)
    kv = torch.randn(2, 1, 16, 64, dtype=torch.float16)
    for bad_k, bad_v in ((kv, kv), (q[:1], q[:1]), (q, q[:, :, :8]), (q, q.to(torch.bfloat16))):
        with pytest.raises(AssertionError):
            M._check_qkv(q, bad_k, bad_v)
        with pytest.raises(AssertionError):           # the direct launchers check too, before allocating or launching
            M.flash_attention_forward(q, bad_k, bad_v, False)
        with pytest.raises(AssertionError):
            M.flash_attention_backward(q, bad_k, bad_v, q, q, q[..., 0].float(), False)
    M._check_qkv(q, kv.expand(2, 4, 16, 64), kv.expand(2, 4, 16, 64))   # the supported MQA form: expanded heads


def test_misaligned_or_oversize_views_are_copied_not_passed_through():
    import _mi355fa as fa
    import My_FlashAttention_optimized as M
    base = torch.zeros(2 * 4 * 16 * 64 + 8, dtype=torch.float16)
    off = 1 if base.data_ptr() % 16 == 0 else 0
    t = base[off + 0: off + 2 * 4 * 16 * 64].view(2, 4, 16, 64)            # contiguous, base 2 bytes off a 16-B boundary
    if t.data_ptr() % 16:
        assert t.is_contiguous() and not fa.strided_ok(t)
        (c,) = M._in_place(t)
        assert c.data_ptr() % 16 == 0 and c.is_contiguous() and torch.equal(c, t)
    big = torch.empty(0, dtype=torch.float16, device="meta").as_strided((1, 1, 3, 64), (0, 0, 1 << 30, 1))  # 2 GiB rows
    assert not fa.strided_ok(big)


def test_verify_results_kat_and_return_value(capsys):
    from _util import load_kat
    import _verify_func as V
    b = torch.linspace(-1, 1, 64).view(8, 8)
    for case in load_kat()["verify"]:
        t = b + case["eps"] * torch.sin(torch.arange(64.0)).view(8, 8)
        m = V.verify_results(b, t)
        out = capsys.readouterr().out
        assert ("Test Passed" in out) == case["passed"] == m["passed"]
        assert "Max Normalized Error (allclose-style)" in out
        assert abs(m["max_norm"] - case["max_norm"]) <= 6e-3 * case["max_norm"]


def test_package_naive_attention_kat():
    # a12: the PACKAGE's naive_attention (not the oracle's) against the KAT taken from the reference's function (P:130-144)
    from _util import load_kat
    import Performance_Comparison as P
    n = 2 * 1 * 4 * 8
    q = ((torch.arange(n) % 7 - 3) / 4.0).view(2, 1, 4, 8)
    k = ((torch.arange(n) % 5 - 2) / 3.0).view(2, 1, 4, 8)
    v = ((torch.arange(n) % 3 - 1) / 2.0).view(2, 1, 4, 8)
    for case in load_kat()["naive"]:
        o = P.naive_attention(q, k, v, case["causal"])
        assert torch.allclose(o.flatten(), torch.tensor(case["o"]), atol=1e-6)
        assert abs(o.sum().item() - case["sum"]) < 1e-5
    sig = [(p.name, p.default) for p in inspect.signature(P.benchmark_attention).parameters.values()]
    E = inspect.Parameter.empty
    assert sig[:11] == [("provider", E), ("mode", E), ("B", E), ("H", E), ("S_q", E), ("S_k", E), ("D", E),
                        ("is_causal", E), ("device", E), ("warmup", 10), ("repeat", 30)]          # P:9-21
    assert [p for p in inspect.signature(P.timing).parameters] == ["run_fn", "warmup", "repeat"]   # P:111


def test_host_modules_parse_on_py310_grammar():
    # the reference's own binding does not parse here (nested same-quote f-strings, M:205-211)
    for f in ("My_FlashAttention_optimized.py", "_verify_func.py", "Performance_Comparison.py", "_mi355fa.py", "_scaling.py"):
        ast.parse(open(os.path.join(PKG, f)).read(), feature_version=(3, 10))


def test_product_path_does_not_touch_the_oracle():
    for f in os.listdir(PKG):
        if f.endswith(".py"):
            src = open(os.path.join(PKG, f)).read()
            assert "fa_oracle" not in src and "oracle" not in src.lower().replace("# oracle", ""), f


def test_work_list_division_is_exact():
    """fa_kernels.h FastDiv (round 4): the persistent kernels decode a work item with a multiply-shift division by launch
    constants (slices per (batch, head), heads).  Exact for every 0 <= n < 2^31 and 1 <= d < 2^31 by construction; swept here
    over the divisors a launch can produce and the numerators around every multiple, plus random pairs."""
    import ctypes
    import random
    import _mi355fa as fa
    f = fa.lib.fa_debug_fastdiv
    f.argtypes = [ctypes.c_int, ctypes.c_int]
    f.restype = ctypes.c_int
    rng = random.Random(0)
    ds = list(range(1, 130)) + [255, 256, 257, 1000, 4095, 4096, 4097, 65535, 65536, 65537, 10 ** 6 + 3, 2 ** 30, 2 ** 31 - 1]
    for d in ds:
        ns = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31 - d, 2 ** 30}
        ns |= {k * d + e for k in (3, 7, 1000, (2 ** 31 - 1) // d) for e in (-1, 0, 1)}
        ns |= {rng.randrange(2 ** 31) for _ in range(50)}
        for n in ns:
            if 0 <= n < 2 ** 31:
                assert f(n, d) == n // d, (n, d, f(n, d))
    for _ in range(20000):
        n, d = rng.randrange(2 ** 31), rng.randrange(1, 2 ** 31)
        assert f(n, d) == n // d, (n, d)
