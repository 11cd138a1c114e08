"""CPU tests: the oracle against the reference's own outputs (tests/golden, produced by
oracle/make_golden.py from the reference's Triton kernel bodies) and against fp64 math."""
import pytest
import torch

import fa_oracle as fo
from _util import golden_names, load_golden, load_kat, rand_inputs


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_kernels(name):
    g = load_golden(name)
    m = g["meta"]
    r = fo.fwd_bwd_tiled(g["Q"], g["K"], g["V"], g["dO"], m["causal"], m["BM"], m["BN"])
    # same algorithm, same rounding points: only fp32 summation order differs (a few fp16 ulps flip)
    for k in ("O", "dQ", "dK", "dV"):
        assert fo.rel_fro(g["ref_" + k], r[k]) < 5e-5, k
        assert (g["ref_" + k].float() - r[k].float()).abs().max() < 4e-3, k
    assert (g["ref_LSE"] - r["LSE"]).abs().max() < 4e-6
    assert (g["ref_delta"] - r["delta"]).abs().max() < 1e-3


@pytest.mark.parametrize("name", golden_names())
def test_reference_kernels_match_fp64(name):
    """The pins themselves are sane: reference outputs vs fp64 attention."""
    g = load_golden(name)
    m = g["meta"]
    gt = fo.attention_fp64(g["Q"], g["K"], g["V"], g["dO"], m["causal"])
    for k in ("O", "dQ", "dK", "dV"):
        assert fo.rel_fro(gt[k], g["ref_" + k]) < 1e-3, k  # BASELINE "within 1e-3 rel"
        assert fo.verify_metrics(gt[k], g["ref_" + k])["passed"], k
    assert (g["ref_LSE"].double() - gt["LSE"]).abs().max() < 1e-3  # Phase_3.md:752-753
    assert (g["ref_delta"].double() - gt["delta"]).abs().max() < 5e-3


@pytest.mark.parametrize("shape", [(1, 2, 500, 500, 64, True), (1, 2, 500, 500, 64, False),
                                   (1, 1, 77, 333, 64, False), (1, 2, 384, 128, 64, True),
                                   (1, 1, 200, 200, 128, True)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_oracle_ragged_and_cross_vs_fp64(shape, dtype):
    """Masked tails (Phase-3/4 semantics): any S_q, S_k; where the reference's descriptor path is wrong."""
    B, H, Sq, Sk, D, causal = shape
    Q, K, V, dO = rand_inputs(B, H, Sq, Sk, D, dtype, seed=3)
    gt = fo.attention_fp64(Q, K, V, dO, causal)
    r = fo.fwd_bwd_tiled(Q, K, V, dO, causal, 64, 64)
    tol = 1e-3 if dtype == torch.float16 else 8e-3
    for k in ("O", "dQ", "dK", "dV"):
        assert fo.rel_fro(gt[k], r[k]) < tol, (k, fo.rel_fro(gt[k], r[k]))
    assert (r["LSE"].double() - gt["LSE"]).abs().max() < 1e-3


def test_oracle_tile_config_insensitive():
    Q, K, V, dO = rand_inputs(1, 2, 256, 256, 64, torch.float16, seed=5)
    a = fo.fwd_bwd_tiled(Q, K, V, dO, True, 64, 64)
    b = fo.fwd_bwd_tiled(Q, K, V, dO, True, 32, 64)
    for k in ("O", "dQ", "dK", "dV"):
        assert fo.rel_fro(a[k], b[k]) < 1e-4


def test_fp64_matches_torch_sdpa():
    Q, K, V, dO = rand_inputs(2, 2, 96, 160, 64, torch.float32, seed=7)
    for causal in (False, True):
        gt = fo.attention_fp64(Q, K, V, dO, causal)
        o = torch.nn.functional.scaled_dot_product_attention(Q.double(), K.double(), V.double(), is_causal=causal)
        assert torch.allclose(gt["O"], o, atol=1e-12)


def test_verify_metrics_kat():
    kat = load_kat()["verify"]
    b = torch.linspace(-1, 1, 64).view(8, 8)
    for case in kat:
        t = b + case["eps"] * torch.sin(torch.arange(64.0)).view(8, 8)
        m = fo.verify_metrics(b, t)
        assert m["passed"] == case["passed"]
        for k in ("max_abs", "mean_abs", "max_rel", "max_norm"):
            assert abs(m[k] - case[k]) <= 6e-3 * abs(case[k]), k  # the reference prints 3 digits
        assert abs(m["cos"] - case["cos"]) < 2e-6


def test_naive_attention_kat():
    n = 2 * 1 * 4 * 8
    q = ((torch.arange(n) % 7 - 3) / 4.0).view(2, 1, 4, 8)
    k = ((torch.arange(n) % 5 - 2) / 3.0).view(2, 1, 4, 8)
    v = ((torch.arange(n) % 3 - 1) / 2.0).view(2, 1, 4, 8)
    for case in load_kat()["naive"]:
        o = fo.naive_attention(q, k, v, case["causal"])
        assert torch.allclose(o.flatten(), torch.tensor(case["o"]), atol=1e-6)


def test_flops_kat():
    for c in load_kat()["flops"]:
        for mode in ("fwd", "bwd", "fwd_bwd"):
            assert fo.attention_flops(c["B"], c["H"], c["S"], c["S"], c["D"], c["causal"], mode) == c[mode]
    assert fo.attention_flops(4, 32, 4096, 4096, 64, True, "fwd") == 274_877_906_944
    assert fo.attention_flops(4, 32, 4096, 4096, 64, True, "fwd_bwd") == 962_072_674_304


def test_varlen_oracle_equals_the_batched_oracle_on_equal_lengths():
    """attention_varlen_fp64 (the ground truth of the varlen extension, Phase_6.md:119-178) on a batch whose sequences
    all have the same length is the batched fp64 oracle on the packed-to-[B,H,S,D] view, and sequences do not leak."""
    torch.manual_seed(3)
    B, H, S, D = 3, 2, 24, 16
    Q, K, V, dO = (torch.randn(B, H, S, D) for _ in range(4))
    pack = lambda t: t.transpose(1, 2).reshape(B * S, H, D)
    cu = [0, S, 2 * S, 3 * S]
    for causal in (False, True):
        ref = fo.attention_fp64(Q, K, V, dO, causal)
        got = fo.attention_varlen_fp64(pack(Q), pack(K), pack(V), pack(dO), cu, cu, causal)
        for k in ("O", "dQ", "dK", "dV"):
            assert torch.allclose(got[k], pack(ref[k]), atol=1e-12), k
        assert torch.allclose(got["LSE"], ref["LSE"].permute(1, 0, 2).reshape(H, B * S), atol=1e-12)
    # ragged, different q / k lengths, one empty sequence: each sequence is its own problem
    cu_q, cu_k = [0, 5, 5, 17], [0, 9, 12, 20]
    Q, dO = torch.randn(17, H, D), torch.randn(17, H, D)
    K, V = torch.randn(20, H, D), torch.randn(20, H, D)
    got = fo.attention_varlen_fp64(Q, K, V, dO, cu_q, cu_k, True)
    one = fo.attention_fp64(Q[5:17].transpose(0, 1)[None], K[12:20].transpose(0, 1)[None], V[12:20].transpose(0, 1)[None],
                            dO[5:17].transpose(0, 1)[None], True)
    assert torch.allclose(got["O"][5:17], one["O"][0].transpose(0, 1), atol=1e-12)
    assert got["dK"][9:12].abs().max() == 0          # keys of the empty-query sequence get no gradient


def test_philox4x32_10_known_answers_and_keep_mask_layout():
    """The oracle's Philox against the Random123 known-answer vectors (kat_vectors: philox4x32 10), and the byte /
    word layout of the 4 x 4 keep patches the kernels regenerate (include/mi355fa.h, fa_*_dropout)."""
    import numpy as np
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kats:
        got = fo.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(x) for x in got) == out
    keep, rp = fo.dropout_keep_mask(2, 3, 37, 50, 0.25, seed=0x123456789abcdef, offset=7)
    assert keep.shape == (2, 3, 37, 50) and rp == 256.0 / 192.0
    # element (b=1, h=2, q=13, k=22): word q & 3 = 1, byte k & 3 = 2 of the patch (q >> 2, k >> 2) = (3, 5)
    w = fo.philox4x32_10(np.uint32(3), np.uint32(5), np.uint32(1 * 3 + 2), np.uint32(7), 0x89abcdef, 0x01234567)
    assert bool(keep[1, 2, 13, 22]) == (((int(w[1]) >> 16) & 255) >= 64)
    frac = keep.float().mean().item()
    assert abs(frac - 0.75) < 0.02                      # P(keep) = 1 - p
    k0, _ = fo.dropout_keep_mask(1, 1, 8, 8, 0.0, seed=1)
    assert k0.all()                                     # p = 0 keeps everything
    k2, _ = fo.dropout_keep_mask(2, 3, 37, 50, 0.25, seed=0x123456789abcdef, offset=8)
    assert not torch.equal(keep, k2)                    # a different offset is a different mask


def test_varlen_dropout_oracle_equals_the_batched_one_on_equal_lengths():
    """attention_varlen_dropout_fp64 on a packed batch of equal-length sequences = attention_dropout_fp64 on the same
    sequences stacked as [B, H, S, D] with the same (seed, offset): sequence b uses the mask of batch index b."""
    B, H, S, D, p, seed, offset = 3, 2, 37, 16, 0.3, 0x1234ABCD, 7
    torch.manual_seed(2)
    Q, K, V, dO = (torch.randn(B, H, S, D, dtype=torch.float64) for _ in range(4))
    keep, rp = fo.dropout_keep_mask(B, H, S, S, p, seed, offset)
    want = fo.attention_dropout_fp64(Q, K, V, dO, True, keep, rp)
    pk = lambda x: x.transpose(1, 2).reshape(B * S, H, D)
    cu = [S * i for i in range(B + 1)]
    got = fo.attention_varlen_dropout_fp64(pk(Q), pk(K), pk(V), pk(dO), cu, cu, True, p, seed, offset)
    for k in ("O", "dQ", "dK", "dV"):
        assert torch.allclose(got[k], pk(want[k]), rtol=1e-12, atol=1e-12), k
