"""GPU tests (-m gpu) of the variable-length extension (reference text Phase_6.md:119-178): packed
[total, H, D] tensors + cu_seqlens through fa_*_varlen, against the per-sequence fp64 oracle
(fa_oracle.attention_varlen_fp64) and -- for equal lengths -- bit for bit against the fixed-length kernels.
Tolerances as in test_gpu_parity.py (fp16 relFro < 1e-3; bf16 < max(2 x PyTorch's bf16 SDPA level, 4e-3))."""
import ctypes

import pytest
import torch

import fa_oracle as fo

pytestmark = pytest.mark.gpu
F16, BF16 = torch.float16, torch.bfloat16


def _M():
    import My_FlashAttention_optimized as M
    return M


@pytest.fixture(params=[0, 1, 2, 3, 4], ids=["auto", "family1", "family2", "family3", "family4"])
def impl(request):
    import _mi355fa as fa
    fn = fa.lib.fa_debug_force_impl
    fn.argtypes = [ctypes.c_int] * 3
    fn.restype = None
    fn(request.param, request.param, request.param)
    yield request.param
    fn(0, 0, 0)


def _cu(lens):
    out = [0]
    for n in lens:
        out.append(out[-1] + n)
    return out


def _run(Q, K, V, dO, cu_q, cu_k, causal):
    M = _M()
    q, k, v = (x.cuda().requires_grad_(True) for x in (Q, K, V))
    cq = torch.tensor(cu_q, dtype=torch.int32, device="cuda")
    ck = torch.tensor(cu_k, dtype=torch.int32, device="cuda")
    mq = max(b - a for a, b in zip(cu_q, cu_q[1:]))
    mk = max(b - a for a, b in zip(cu_k, cu_k[1:]))
    o = M.flash_attention_varlen(q, k, v, cq, ck, mq, mk, causal)
    o.backward(dO.cuda())
    torch.cuda.synchronize()
    return {"O": o.detach().cpu(), "dQ": q.grad.cpu(), "dK": k.grad.cpu(), "dV": v.grad.cpu()}


CASES = [
    # (q lengths, k lengths or None = same, H, D, causal)
    ([5, 128, 333, 64, 1], None, 3, 64, False),
    ([5, 128, 333, 64, 1], None, 3, 64, True),
    ([256, 256, 256], None, 2, 64, True),
    ([700, 3, 129], [77, 500, 129], 2, 64, False),        # cross attention per sequence
    ([300, 40, 129], [200, 90, 129], 2, 64, True),        # causal with S_q != S_k in both directions
    ([64, 0, 200], [64, 0, 200], 2, 64, True),            # an empty sequence in the middle
    ([64, 5, 130], [64, 0, 130], 2, 64, True),            # queries with NO keys: O = 0, dQ = 0, nothing read
    ([64, 5], [64, 0], 2, 64, False),                     # ... as the last sequence: its K/V rows would lie past the end
    ([0, 70, 300], [33, 70, 0], 2, 64, True),             # keys with no queries: dK = dV = 0 for them
    ([200, 9], [200, 0], 2, 128, False),
    ([1000], None, 4, 64, True),                          # batch of one
    ([130, 257, 31], None, 2, 128, True),
    ([130, 257, 31], [100, 257, 300], 2, 128, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "q%s_k%s_H%d_D%d_%s" % ("-".join(map(str, c[0])), "same" if c[1] is None else "-".join(map(str, c[1])), c[2], c[3], "c" if c[4] else "f"))
@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_varlen_against_fp64_oracle(case, dtype, impl):
    lq, lk, H, D, causal = case
    lk = lq if lk is None else lk
    cu_q, cu_k = _cu(lq), _cu(lk)
    g = torch.Generator().manual_seed(sum(lq) * 31 + sum(lk))
    mk = lambda n: torch.randn(n, H, D, generator=g).to(dtype)
    Q, K, V, dO = mk(cu_q[-1]), mk(cu_k[-1]), mk(cu_k[-1]), mk(cu_q[-1])
    gt = fo.attention_varlen_fp64(Q, K, V, dO, cu_q, cu_k, causal)
    r = _run(Q, K, V, dO, cu_q, cu_k, causal)
    r2 = _run(Q, K, V, dO, cu_q, cu_k, causal)
    for k in ("O", "dQ", "dK", "dV"):
        assert torch.equal(r[k], r2[k]), (k, "not deterministic")
        assert torch.isfinite(r[k].float()).all(), k
        err = fo.rel_fro(gt[k], r[k])
        assert err < (1e-3 if dtype == F16 else 6e-3), (k, err)
    for b in range(len(lq)):   # a side without a partner: exact zeros (include/mi355fa.h "a length of 0 is allowed")
        if lq[b] > 0 and lk[b] == 0:
            assert (r["O"][cu_q[b]:cu_q[b + 1]] == 0).all() and (r["dQ"][cu_q[b]:cu_q[b + 1]] == 0).all()
        if lk[b] > 0 and lq[b] == 0:
            assert (r["dK"][cu_k[b]:cu_k[b + 1]] == 0).all() and (r["dV"][cu_k[b]:cu_k[b + 1]] == 0).all()
    if causal:   # keys beyond a sequence's query count are invisible: their gradient rows are written, as zeros
        for b in range(len(lq)):
            if lk[b] > lq[b] > 0:
                a, e = cu_k[b] + lq[b], cu_k[b + 1]
                assert (r["dK"][a:e] == 0).all() and (r["dV"][a:e] == 0).all()


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
@pytest.mark.parametrize("D", [64, 128])
def test_equal_length_varlen_is_bit_identical_to_the_fixed_length_kernels(dtype, causal, D, impl):
    """B sequences of one length, packed: the varlen launch runs the same kernels on the same numbers as the [B,H,S,D]
    launch on the transposed view -- O, dQ, dK, dV must agree bit for bit (and LSE / delta through the raw launchers)."""
    M = _M()
    import _mi355fa as fa
    import _mi355fa_torch as ext
    if impl == 2:   # packed rows always take family 1 for forward / dQ (like strided views): pin the fixed-length twin too
        fa.lib.fa_debug_force_impl(1, 1, 2)
    if impl == 4:   # ... and family 4 (forward only) takes fixed-length launches only
        fa.lib.fa_debug_force_impl(1, 1, 1)
    B, H, S = 3, 2, 320
    torch.manual_seed(5)
    Qp, Kp, Vp, dOp = (torch.randn(B * S, H, D, device="cuda", dtype=dtype) for _ in range(4))
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
    to4 = lambda t: t.view(B, S, H, D).transpose(1, 2).contiguous()
    O4, LSE4 = M.flash_attention_forward(to4(Qp), to4(Kp), to4(Vp), causal)
    dQ4, dK4, dV4 = M.flash_attention_backward(to4(Qp), to4(Kp), to4(Vp), O4, to4(dOp), LSE4, causal)
    Ov, LSEv = ext.varlen_forward_launch(Qp, Kp, Vp, cu, cu, S, S, causal)
    dQv, dKv, dVv = ext.varlen_backward_launch(Qp, Kp, Vp, Ov, dOp, LSEv, cu, cu, S, S, causal)
    torch.cuda.synchronize()
    for a4, av in ((O4, Ov), (dQ4, dQv), (dK4, dKv), (dV4, dVv)):
        assert torch.equal(to4(av), a4)
    assert torch.equal(LSEv.view(H, B, S).transpose(0, 1), LSE4)


def test_varlen_headline_like_batch_matches_padded_sdpa():
    """A realistic ragged batch (lengths up to 4096, 16 heads): every sequence against torch SDPA on that sequence."""
    M = _M()
    lens = [4096, 1000, 2500, 37, 3333]
    cu = _cu(lens)
    H, D = 16, 64
    torch.manual_seed(2)
    Q, K, V, dO = (torch.randn(cu[-1], H, D, device="cuda", dtype=BF16) for _ in range(4))
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    cut = torch.tensor(cu, dtype=torch.int32, device="cuda")
    o = M.flash_attention_varlen(q, k, v, cut, cut, max(lens), max(lens), True)
    o.backward(dO)
    for b in range(len(lens)):
        a, e = cu[b], cu[b + 1]
        sl = lambda t: t[a:e].transpose(0, 1).unsqueeze(0).clone().requires_grad_(True)
        q2, k2, v2 = sl(Q), sl(K), sl(V)
        o2 = torch.nn.functional.scaled_dot_product_attention(q2, k2, v2, is_causal=True)
        o2.backward(dO[a:e].transpose(0, 1).unsqueeze(0))
        for name, ours, ref in (("O", o[a:e], o2), ("dQ", q.grad[a:e], q2.grad), ("dK", k.grad[a:e], k2.grad), ("dV", v.grad[a:e], v2.grad)):
            ref_p = ref[0].transpose(0, 1).float()
            err = ((ours.detach().float() - ref_p).norm() / ref_p.norm()).item()
            assert err < 8e-3, (b, name, err)
