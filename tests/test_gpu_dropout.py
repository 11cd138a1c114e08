"""GPU tests (-m gpu) of attention dropout (reference text Phase_6.md:54-113): the three kernels regenerate
the same Philox4x32-10 keep mask from (seed, offset); checked against the CPU restatement of that mask
(fa_oracle.dropout_keep_mask, pinned by the Random123 known-answer vectors in tests/test_oracle.py) and the fp64
attention with that mask (fa_oracle.attention_dropout_fp64).  Tolerances as in test_gpu_parity.py."""
import pytest
import torch

import fa_oracle as fo
from _util import rand_inputs

pytestmark = pytest.mark.gpu
F16, BF16 = torch.float16, torch.bfloat16


def _M():
    import My_FlashAttention_optimized as M
    return M


def _run(Q, K, V, dO, causal, p, seed, offset=0):
    M = _M()
    q, k, v = (x.cuda().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention_dropout(q, k, v, causal, p, seed, offset)
    o.backward(dO.cuda())
    torch.cuda.synchronize()
    return {"O": o.detach().cpu(), "dQ": q.grad.cpu(), "dK": k.grad.cpu(), "dV": v.grad.cpu()}


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_forward_and_dkv_kernels_regenerate_the_oracle_mask_exactly(D, dtype):
    """Q = K = 0 makes the softmax uniform (P = 1/S_k) and V = dO = identity rows expose single weights:
    O[q, :] lists the kept weights of query q as the forward sees them, dV[k, :] lists them as the dK/dV kernel sees
    them.  Both must be the oracle's mask, element for element."""
    import _mi355fa_torch as ext
    B, H, S = 2, 2, D                      # S_q = S_k = D so that an identity fits V and dO
    p, seed, offset = 0.3, 0xDEADBEEF12345, 3
    keep, rp = fo.dropout_keep_mask(B, H, S, S, p, seed, offset)
    Z = torch.zeros(B, H, S, D, device="cuda", dtype=dtype)
    eye = torch.eye(S, D, device="cuda", dtype=dtype).expand(B, H, S, D).contiguous()
    O, LSE = ext.dropout_forward_launch(Z, Z, eye, False, p, seed, offset)
    want = keep.float() * (rp / S)
    assert torch.allclose(O.float().cpu(), want, rtol=2e-2, atol=1e-6)
    assert ((O.float().cpu() > 0) == keep).all()
    dQ, dK, dV = ext.dropout_backward_launch(Z, Z, eye, O, eye, LSE, False, p, seed, offset)
    assert ((dV.float().cpu().transpose(-1, -2) > 0) == keep).all()      # dV[k, q] = P_drop[q, k]
    # the kept fraction and the scale
    assert abs(keep.float().mean().item() - (1 - round(p * 256) / 256)) < 0.02
    assert abs(ext.dropout_keep_scale(p) - rp) < 1e-6


CASES = [
    # B, H, Sq, Sk, D, causal, p
    (2, 3, 256, 256, 64, False, 0.1),
    (2, 3, 256, 256, 64, True, 0.1),
    (1, 2, 333, 500, 64, False, 0.5),
    (1, 2, 500, 333, 64, True, 0.25),
    (1, 1, 129, 65, 64, True, 0.9),
    (1, 2, 320, 320, 128, True, 0.2),
    (1, 1, 77, 333, 128, False, 0.1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%dH%d_%dx%d_D%d_%s_p%g" % (c[0], c[1], c[2], c[3], c[4], "c" if c[5] else "f", c[6]))
@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_dropout_against_fp64_oracle_with_the_same_mask(case, dtype):
    B, H, Sq, Sk, D, causal, p = case
    Q, K, V, dO = rand_inputs(B, H, Sq, Sk, D, dtype, seed=17)
    seed, offset = 0x5EED0000 + Sq, 11
    keep, rp = fo.dropout_keep_mask(B, H, Sq, Sk, p, seed, offset)
    gt = fo.attention_dropout_fp64(Q, K, V, dO, causal, keep, rp)
    r = _run(Q, K, V, dO, causal, p, seed, offset)
    r2 = _run(Q, K, V, dO, causal, p, seed, offset)
    for k in ("O", "dQ", "dK", "dV"):
        assert torch.equal(r[k], r2[k]), (k, "same seed / offset must give the same bits")
        assert torch.isfinite(r[k].float()).all(), k
        err = fo.rel_fro(gt[k], r[k])
        assert err < (1.5e-3 if dtype == F16 else 8e-3), (k, err)
    r3 = _run(Q, K, V, dO, causal, p, seed, offset + 1)
    assert not torch.equal(r3["O"], r["O"])          # another offset: another mask


def test_zero_probability_is_plain_attention_and_autograd_matches_launchers():
    M = _M()
    Q, K, V, dO = (x.cuda() for x in rand_inputs(1, 2, 256, 256, 64, BF16, seed=4))
    with torch.no_grad():
        assert torch.equal(M.flash_attention_dropout(Q, K, V, True, 0.0, seed=9), M.flash_attention(Q, K, V, True))
    import _mi355fa_torch as ext
    O, LSE = ext.dropout_forward_launch(Q, K, V, True, 0.2, 7, 0)
    g = ext.dropout_backward_launch(Q, K, V, O, dO, LSE, True, 0.2, 7, 0)
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention_dropout(q, k, v, True, 0.2, seed=7)
    o.backward(dO)
    assert torch.equal(o, O) and torch.equal(q.grad, g[0]) and torch.equal(k.grad, g[1]) and torch.equal(v.grad, g[2])
    # LSE is the undropped softmax's
    assert torch.allclose(LSE, M.flash_attention_forward(Q, K, V, True)[1], atol=2e-3)


# ---------------------------------------------------------------- dropout composed with the other extensions (mi355fa_opts)
def _cu(lens):
    out = [0]
    for n in lens:
        out.append(out[-1] + n)
    return out


VARLEN_CASES = [
    # (q lengths, k lengths or None = same, H, D, causal, p)
    ([5, 128, 333, 64, 1], None, 3, 64, True, 0.2),
    ([300, 40, 129], [200, 90, 129], 2, 64, False, 0.5),
    ([64, 0, 200], [64, 0, 200], 2, 64, True, 0.1),
    ([130, 257, 31], [100, 257, 300], 2, 128, True, 0.25),
]


@pytest.mark.parametrize("case", VARLEN_CASES, ids=lambda c: "q%s_H%d_D%d_%s_p%g" % ("-".join(map(str, c[0])), c[2], c[3], "c" if c[4] else "f", c[5]))
@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_varlen_with_dropout_against_fp64_oracle(case, dtype):
    """Packed ragged batch + dropout: sequence b gets the mask of batch index b, positions counted inside the sequence."""
    M = _M()
    lq, lk, H, D, causal, p = case
    lk = lk or lq
    cu_q, cu_k = _cu(lq), _cu(lk)
    torch.manual_seed(23)
    Q = torch.randn(cu_q[-1], H, D).to(dtype)
    K, V = (torch.randn(cu_k[-1], H, D).to(dtype) for _ in range(2))
    dO = torch.randn(cu_q[-1], H, D).to(dtype)
    seed, offset = 0xABCDEF0123, 5
    gt = fo.attention_varlen_dropout_fp64(Q, K, V, dO, cu_q, cu_k, causal, p, seed, offset)
    q, k, v = (x.cuda().requires_grad_(True) for x in (Q, K, V))
    cq, ck = (torch.tensor(c, dtype=torch.int32, device="cuda") for c in (cu_q, cu_k))
    o = M.flash_attention_varlen(q, k, v, cq, ck, max(lq), max(lk), causal, dropout_p=p, seed=seed, offset=offset)
    o.backward(dO.cuda())
    for name, got in (("O", o.detach()), ("dQ", q.grad), ("dK", k.grad), ("dV", v.grad)):
        assert torch.isfinite(got.float()).all(), name
        err = fo.rel_fro(gt[name], got.cpu())
        assert err < (1.5e-3 if dtype == F16 else 8e-3), (name, err)


@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
def test_packed_equal_lengths_with_dropout_equal_the_padded_launch_bit_for_bit(causal):
    """Same sequences, same (seed, offset): the packed launch must drop exactly the weights the [B, H, S, D] launch
    drops (the Philox slice index is sequence * H + head in both, whatever order the work list visits them in)."""
    M = _M()
    B, H, S, D, p = 3, 4, 320, 64, 0.3
    Q, K, V, dO = (x.cuda() for x in rand_inputs(B, H, S, S, D, BF16, seed=31))
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention_dropout(q, k, v, causal, p, seed=99, offset=2)
    o.backward(dO)
    pk = lambda x: x.transpose(1, 2).reshape(B * S, H, D).contiguous()
    qp, kp, vp = (pk(x).requires_grad_(True) for x in (Q, K, V))
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
    op = M.flash_attention_varlen(qp, kp, vp, cu, cu, S, S, causal, dropout_p=p, seed=99, offset=2)
    op.backward(pk(dO))
    assert torch.equal(op, pk(o.detach()))
    for a, b in ((qp.grad, q.grad), (kp.grad, k.grad), (vp.grad, v.grad)):
        assert torch.equal(a, pk(b))


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_dropout_reads_strided_views_in_place_and_is_bit_identical(dtype):
    """[B, S, H, D] storage seen as [B, H, S, D]: same mask, same bits as on contiguous copies, outputs in the views' order."""
    M = _M()
    B, H, S, D, p = 2, 3, 333, 64, 0.2
    torch.manual_seed(5)
    xs = [torch.randn(B, S, H, D, device="cuda", dtype=dtype) for _ in range(4)]
    views = [x.transpose(1, 2) for x in xs]
    q, k, v = (x.clone().requires_grad_(True) for x in views[:3])     # clone keeps the [B, S, H, D] memory order
    assert not q.is_contiguous()
    o = M.flash_attention_dropout(q, k, v, True, p, seed=3, offset=1)
    o.backward(views[3])
    qc, kc, vc = (x.contiguous().requires_grad_(True) for x in views[:3])
    oc = M.flash_attention_dropout(qc, kc, vc, True, p, seed=3, offset=1)
    oc.backward(views[3].contiguous())
    assert o.transpose(1, 2).is_contiguous() and torch.equal(o, oc)
    for a, b in ((q.grad, qc.grad), (k.grad, kc.grad), (v.grad, vc.grad)):
        assert torch.equal(a, b)
