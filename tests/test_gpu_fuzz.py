"""Seeded shape fuzz on the GPU (run with -m gpu): random (B, H, Sq, Sk, D, causal, dtype) -- ragged lengths, Sq != Sk,
lengths around every tile boundary the kernels use (32 / 64 / 128 / 256) -- each checked against the fp64 oracle and run
twice for bit-identical results (the kernels use no atomics; a difference is a race, cf. DESIGN.md "raw tile barrier").

Tolerances as in test_gpu_parity.py: fp16 relFro < max(1e-3, 1.5 x the reference algorithm's own error on the same
inputs); bf16 relFro < max(2 x PyTorch's own bf16 SDPA, 4e-3).
One large-grid case per dtype exercises the occupancy-dependent schedule choices (three workgroups per CU).
"""
import os
import random

import pytest
import torch

import fa_oracle as fo
from _util import rand_inputs

pytestmark = pytest.mark.gpu

F16, BF16 = torch.float16, torch.bfloat16


def _cases(n=int(os.environ.get("FA_FUZZ_CASES", "28")), seed=int(os.environ.get("FA_FUZZ_SEED", "20260101"))):
    """28 cases by default; FA_FUZZ_CASES / FA_FUZZ_SEED widen the hunt (e.g. 400 cases in about a minute)."""
    rng = random.Random(seed)
    edges = [1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 191, 192, 255, 256, 257, 320, 383, 384, 500, 512, 640, 777]
    out = []
    for i in range(n):
        D = rng.choice([64, 64, 64, 128])
        causal = rng.random() < 0.5
        Sq = rng.choice(edges)
        Sk = Sq if rng.random() < 0.5 else rng.choice(edges)   # causal with Sk > Sq included: the reference's mask is
        # top-left aligned (K:102), so keys beyond Sq are invisible -- their dK / dV must still be written, as zeros
        B, H = rng.choice([(1, 1), (1, 2), (2, 3), (1, 5)])
        out.append((B, H, Sq, Sk, D, causal, F16 if i % 2 else BF16))
    return out


def _run(Q, K, V, dO, causal):
    import My_FlashAttention_optimized as M
    q, k, v = (x.cuda().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention(q, k, v, is_causal=causal)
    o.backward(dO.cuda())
    torch.cuda.synchronize()
    return {"O": o.detach(), "dQ": q.grad, "dK": k.grad, "dV": v.grad}


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "B%dH%d_%dx%d_D%d_%s_%s" % (c[0], c[1], c[2], c[3], c[4], "c" if c[5] else "f", "h" if c[6] == F16 else "b"))
def test_random_shape_against_fp64_and_itself(case):
    B, H, Sq, Sk, D, causal, dtype = case
    Q, K, V, dO = rand_inputs(B, H, Sq, Sk, D, dtype, seed=Sq * 1000 + Sk)
    gt = fo.attention_fp64(Q, K, V, dO, causal)
    r = _run(Q, K, V, dO, causal)
    r2 = _run(Q, K, V, dO, causal)
    # what the reference's own algorithm (its rounding points restated on the CPU) / PyTorch's 16-bit SDPA reach on
    # the same inputs: at degenerate sizes (two rows, gradients that are pure cancellation) even those sit at 2e-2
    peer = fo.fwd_bwd_tiled(Q, K, V, dO, causal) if dtype == F16 else dict(
        zip(("O", "dQ", "dK", "dV"), fo.cpu_sdpa(Q, K, V, causal, dO)))
    if causal and Sk > Sq:
        assert (r["dK"][:, :, Sq:] == 0).all() and (r["dV"][:, :, Sq:] == 0).all()
    for k in ("O", "dQ", "dK", "dV"):
        assert torch.equal(r[k], r2[k]), (k, "not deterministic")
        out = r[k].cpu()
        assert torch.isfinite(out.float()).all(), k
        if gt[k].abs().max() < 1e-9:     # a single visible key: the gradient is exactly zero
            assert out.float().abs().max() < 2e-3, k
            continue
        err = fo.rel_fro(gt[k], out)
        if dtype == F16:
            assert err < max(1.5 * fo.rel_fro(gt[k], peer[k]), 1e-3), (k, err)
        else:
            # bf16 scale fold: Q is rounded once more after the multiply by scale * log2e, i.e. the
            # kernels compute the attention of a Q half a bf16 ulp away.  Over many keys that averages out; with two or
            # three keys per row a gradient that is pure cancellation moves by up to ~3x the reference algorithm's own
            # error (B1 H2 2x2 non-causal, seed 4242: 1.8e-2 against 6.7e-3, reproduced on the CPU by rounding Q only)
            few = min(Sq, Sk) <= 4
            assert err < max((3 if few else 2) * fo.rel_fro(gt[k], peer[k]), 4e-3), (k, err)


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
def test_large_grid_is_deterministic_and_matches_device_sdpa(dtype, causal):
    """B2 H16 N2048 D64: > 768 workgroups per kernel, i.e. every occupancy-dependent schedule choice is taken."""
    import My_FlashAttention_optimized as M
    torch.manual_seed(7)
    Q, K, V, dO = (torch.randn(2, 16, 2048, 64, device="cuda", dtype=dtype) for _ in range(4))
    outs = []
    for _ in range(3):
        q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
        o = M.flash_attention(q, k, v, causal)
        o.backward(dO)
        outs.append((o.detach(), q.grad, k.grad, v.grad))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)
    q2, k2, v2 = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o2 = torch.nn.functional.scaled_dot_product_attention(q2, k2, v2, is_causal=causal)
    o2.backward(dO)
    tol = 6e-3 if dtype == BF16 else 1.2e-3   # two 16-bit implementations, each at its rounding floor
    for name, a, b in zip(("O", "dQ", "dK", "dV"), outs[0], (o2, q2.grad, k2.grad, v2.grad)):
        err = ((a.float() - b.float()).norm() / b.float().norm()).item()
        assert err < tol, (name, err)
