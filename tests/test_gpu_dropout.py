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
