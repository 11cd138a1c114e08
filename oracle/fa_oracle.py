"""CPU oracle for the FlashAttention forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
path (``flashattention-from-scratch-with-triton_amd/``); only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it,
and there only as the checker / the reported CPU baseline.

Parity status: PINNED.  ``oracle/make_golden.py`` ran the reference's own
Triton kernel bodies (``/root/reference/code/_flash_attention_kernel_optimized.py``)
under ``TRITON_INTERPRET=1`` in the build container and committed their
outputs (O, LSE, delta, dQ, dK, dV) as ``tests/golden/*.npz``;
``tests/test_oracle.py`` checks this restatement against those vectors and
against fp64 math.

What is restated (reference file:line, K = code/_flash_attention_kernel_optimized.py,
M = code/My_FlashAttention_optimized.py, P = code/Performance_Comparison.py,
V = code/_verify_func.py):

* ``fwd_tiled``      -> flash_attention_forward_kernel  K:60-129, launcher M:14-60
* ``bwd_dq_tiled``   -> flash_attention_dQ_kernel       K:188-258
* ``bwd_dkv_tiled``  -> flash_attention_dKV_kernel      K:315-386, order M:111-126
* ``attention_fp64`` -> ground truth (softmax(QK^T/sqrt(D) [+causal]) V and its
  autograd gradients, LSE = logsumexp as in Phase_3.md:699-708,
  delta = rowsum(dO*O) as in Phase_4.md:1176-1194)
* ``naive_attention``-> P:130-144
* ``verify_metrics`` -> V:3-40 (returns the numbers instead of printing them)
* ``attention_flops``-> P:101-107

The tiled functions keep the reference's rounding points: S and the softmax
state in fp32, ``l`` sums the un-rounded fp32 p (K:111) while P@V uses p rounded
to the 16-bit input dtype (K:115), delta is computed from the *rounded* O
(K:210-211), dS and P^T are rounded to 16 bit before their matmuls
(K:253, K:370, K:382), ``* scale`` is applied after each dot (K:93, K:253,
K:382), outputs are cast on store (K:123, K:256, K:385-386).  Unlike the
reference's TensorDescriptor path (which spills a tail tile into the next
head, SURVEY.md finding 3) the tails are masked the way the reference's
Phase-3/4 kernels do (Phase_3.md:148-159, Phase_4.md:365-379,599-605), which
is also what torch SDPA computes.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

LOG2_E = 1.44269504  # K:79


def _dot32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """tl.dot of two 16-bit tiles: exact products, fp32 accumulate."""
    return a.to(torch.float32) @ b.to(torch.float32)


def attention_flops(B: int, H: int, S_q: int, S_k: int, D: int, is_causal: bool, mode: str = "fwd") -> int:
    """Counted FLOPs, the reference's convention (P:101-107)."""
    f = 4 * B * H * S_q * S_k * D // (2 if is_causal else 1)
    if mode == "fwd":
        return f
    if mode == "bwd":
        return int(2.5 * f)
    if mode == "fwd_bwd":
        return int(3.5 * f)
    raise ValueError(mode)


def naive_attention(Q, K, V, is_causal):
    """Materialised attention, P:130-144 (mask built square from S.shape[-1])."""
    scale = 1 / (Q.shape[-1] ** 0.5)
    S = Q @ K.transpose(-2, -1) * scale
    if is_causal:
        n = S.shape[-1]
        mask = torch.triu(torch.ones(n, n), diagonal=1).bool()
        S = S.masked_fill(mask, float("-inf"))
    return torch.softmax(S, dim=-1) @ V


def verify_metrics(bench, out, rtol=1e-2, atol=1e-3) -> Dict[str, float]:
    """The five numbers and the verdict of verify_results, V:3-40."""
    b = bench.to(torch.float32)
    t = out.to(torch.float32)
    diff = (b - t).abs()
    cos = torch.nn.functional.cosine_similarity(b.flatten(), t.flatten(), dim=0).item()
    ok = bool(torch.allclose(b, t, rtol=rtol, atol=atol)) and cos > 0.999
    return {
        "max_abs": diff.max().item(),
        "mean_abs": diff.mean().item(),
        "max_rel": (diff / (b.abs() + 1e-5)).max().item(),
        "max_norm": (diff / (atol + rtol * t.abs())).max().item(),
        "cos": cos,
        "passed": ok,
    }


def rel_fro(ref: torch.Tensor, out: torch.Tensor) -> float:
    r = ref.to(torch.float64)
    return ((out.to(torch.float64) - r).norm() / r.norm().clamp_min(1e-300)).item()


# --------------------------------------------------------------------------
# ground truth
# --------------------------------------------------------------------------
def attention_fp64(Q, K, V, dO=None, is_causal=False) -> Dict[str, torch.Tensor]:
    """fp64 attention with top-left aligned causal mask (K:102) + autograd grads."""
    q = Q.detach().to(torch.float64).requires_grad_(dO is not None)
    k = K.detach().to(torch.float64).requires_grad_(dO is not None)
    v = V.detach().to(torch.float64).requires_grad_(dO is not None)
    D = q.shape[-1]
    S = q @ k.transpose(-2, -1) * (1.0 / math.sqrt(D))
    if is_causal:
        Sq, Sk = S.shape[-2:]
        keep = torch.arange(Sq)[:, None] >= torch.arange(Sk)[None, :]
        S = S.masked_fill(~keep, float("-inf"))
    lse = torch.logsumexp(S, dim=-1)
    P = torch.exp(S - lse[..., None])
    O = P @ v
    out = {"O": O.detach(), "LSE": lse.detach()}
    if dO is not None:
        do = dO.detach().to(torch.float64)
        O.backward(do)
        out.update(dQ=q.grad, dK=k.grad, dV=v.grad, delta=(do * O.detach()).sum(-1))
    return out


# --------------------------------------------------------------------------
# tiled restatement of the three kernels
# --------------------------------------------------------------------------
def _pad_rows(x: torch.Tensor, n: int) -> torch.Tensor:
    """Zero padding of a tail tile (descriptor padding="zero", M:38)."""
    if x.shape[-2] == n:
        return x
    pad = torch.zeros(*x.shape[:-2], n - x.shape[-2], x.shape[-1], dtype=x.dtype)
    return torch.cat([x, pad], dim=-2)


def fwd_tiled(Q, K, V, is_causal=False, BLOCK_M=64, BLOCK_N=64) -> Tuple[torch.Tensor, torch.Tensor]:
    """K:60-129.  Q [B,H,Sq,D], K/V [B,H,Sk,D] 16-bit -> O (dtype of Q), LSE fp32."""
    B, H, Sq, D = Q.shape
    Sk = K.shape[2]
    dt = Q.dtype
    scale = 1 / (D ** 0.5)  # M:56
    O = torch.empty(B, H, Sq, D, dtype=dt)
    LSE = torch.empty(B, H, Sq, dtype=torch.float32)
    for q0 in range(0, Sq, BLOCK_M):
        rows = min(BLOCK_M, Sq - q0)
        Qb = _pad_rows(Q[:, :, q0:q0 + rows], BLOCK_M)
        sq = q0 + torch.arange(BLOCK_M)
        m = torch.full((B, H, BLOCK_M), float("-inf"))          # K:75
        l = torch.zeros(B, H, BLOCK_M)                          # K:76
        o = torch.zeros(B, H, BLOCK_M, D)                       # K:77
        loop_end = min((q0 + BLOCK_M) if is_causal else Sk, Sk)  # K:82 (clamped: keys >= Sk are masked anyway)
        for s0 in range(0, loop_end, BLOCK_N):
            cols = min(BLOCK_N, Sk - s0)
            Kb = _pad_rows(K[:, :, s0:s0 + cols], BLOCK_N)
            Vb = _pad_rows(V[:, :, s0:s0 + cols], BLOCK_N)
            sk = s0 + torch.arange(BLOCK_N)
            S = _dot32(Qb, Kb.transpose(-2, -1)) * scale                       # K:93
            S = torch.where((sk < Sk)[None, None, None, :], S, float("-inf"))  # K:94
            if is_causal and not (q0 >= s0 + BLOCK_N - 1):                     # K:98-101
                S = torch.where((sq[:, None] >= sk[None, :])[None, None], S, float("-inf"))  # K:102-103
            m_new = torch.maximum(m, S.max(dim=-1).values)                     # K:106
            corr = torch.exp2((m - m_new) * LOG2_E)                            # K:108
            p = torch.exp2((S - m_new[..., None]) * LOG2_E)                    # K:109
            l = l * corr + p.sum(-1)                                           # K:111
            o = o * corr[..., None] + _dot32(p.to(dt), Vb)                     # K:115
            m = m_new
        o = o / l[..., None]                                                   # K:120
        O[:, :, q0:q0 + rows] = o[:, :, :rows].to(dt)                          # K:123
        LSE[:, :, q0:q0 + rows] = (m + torch.log(l))[:, :, :rows]              # K:126-129
    return O, LSE


def bwd_dq_tiled(Q, K, V, O, dO, LSE, is_causal=False, BLOCK_M=64, BLOCK_N=64):
    """K:188-258 -> (dQ 16-bit, delta fp32)."""
    B, H, Sq, D = Q.shape
    Sk = K.shape[2]
    dt = Q.dtype
    scale = 1 / (D ** 0.5)
    dQ = torch.empty(B, H, Sq, D, dtype=dt)
    delta = torch.empty(B, H, Sq, dtype=torch.float32)
    for q0 in range(0, Sq, BLOCK_M):
        rows = min(BLOCK_M, Sq - q0)
        Qb = _pad_rows(Q[:, :, q0:q0 + rows], BLOCK_M)
        dOb = _pad_rows(dO[:, :, q0:q0 + rows], BLOCK_M)
        Ob = _pad_rows(O[:, :, q0:q0 + rows], BLOCK_M)
        lse = torch.zeros(B, H, BLOCK_M)
        lse[:, :, :rows] = LSE[:, :, q0:q0 + rows]
        sq = q0 + torch.arange(BLOCK_M)
        dlt = (dOb.to(torch.float32) * Ob.to(torch.float32)).sum(-1)           # K:210-211
        acc = torch.zeros(B, H, BLOCK_M, D)
        loop_end = min((q0 + BLOCK_M) if is_causal else Sk, Sk)                # K:219
        for s0 in range(0, loop_end, BLOCK_N):
            cols = min(BLOCK_N, Sk - s0)
            Kb = _pad_rows(K[:, :, s0:s0 + cols], BLOCK_N)
            Vb = _pad_rows(V[:, :, s0:s0 + cols], BLOCK_N)
            sk = s0 + torch.arange(BLOCK_N)
            S = _dot32(Qb, Kb.transpose(-2, -1)) * scale                       # K:230
            S = torch.where((sk < Sk)[None, None, None, :], S, float("-inf"))  # K:231
            S = torch.where((sq < Sq)[None, None, :, None], S, float("-inf"))  # K:233
            if is_causal and not (q0 >= s0 + BLOCK_N - 1):                     # K:236-239
                S = torch.where((sq[:, None] >= sk[None, :])[None, None], S, float("-inf"))
            P = torch.exp2((S - lse[..., None]) * LOG2_E)                      # K:244
            dP = _dot32(dOb, Vb.transpose(-2, -1))                             # K:247
            dS = P * (dP - dlt[..., None])                                     # K:250
            acc = acc + _dot32(dS.to(dt), Kb) * scale                          # K:253
        dQ[:, :, q0:q0 + rows] = acc[:, :, :rows].to(dt)                       # K:256
        delta[:, :, q0:q0 + rows] = dlt[:, :, :rows]                           # K:258
    return dQ, delta


def bwd_dkv_tiled(Q, K, V, dO, LSE, delta, is_causal=False, BLOCK_M=64, BLOCK_N=64):
    """K:315-386 -> (dK, dV) 16-bit."""
    B, H, Sq, D = Q.shape
    Sk = K.shape[2]
    dt = Q.dtype
    scale = 1 / (D ** 0.5)
    dK = torch.empty(B, H, Sk, D, dtype=dt)
    dV = torch.empty(B, H, Sk, D, dtype=dt)
    for k0 in range(0, Sk, BLOCK_N):
        cols = min(BLOCK_N, Sk - k0)
        Kb = _pad_rows(K[:, :, k0:k0 + cols], BLOCK_N)
        Vb = _pad_rows(V[:, :, k0:k0 + cols], BLOCK_N)
        sk = k0 + torch.arange(BLOCK_N)
        dK_acc = torch.zeros(B, H, BLOCK_N, D)
        dV_acc = torch.zeros(B, H, BLOCK_N, D)
        loop_start = k0 if is_causal else 0                                    # K:341
        loop_start = (loop_start // BLOCK_M) * BLOCK_M  # tiles stay BLOCK_M aligned (equal when BM == BN)
        for q0 in range(loop_start, Sq, BLOCK_M):
            rows = min(BLOCK_M, Sq - q0)
            Qb = _pad_rows(Q[:, :, q0:q0 + rows], BLOCK_M)
            dOb = _pad_rows(dO[:, :, q0:q0 + rows], BLOCK_M)
            lse = torch.zeros(B, H, BLOCK_M)
            lse[:, :, :rows] = LSE[:, :, q0:q0 + rows]
            dlt = torch.zeros(B, H, BLOCK_M)
            dlt[:, :, :rows] = delta[:, :, q0:q0 + rows]
            sq = q0 + torch.arange(BLOCK_M)
            S = _dot32(Qb, Kb.transpose(-2, -1)) * scale                       # K:353
            S = torch.where((sk < Sk)[None, None, None, :], S, float("-inf"))  # K:354
            S = torch.where((sq < Sq)[None, None, :, None], S, float("-inf"))  # K:356
            if is_causal and not (q0 >= k0 + BLOCK_N - 1):                     # K:359-362
                S = torch.where((sq[:, None] >= sk[None, :])[None, None], S, float("-inf"))
            P = torch.exp2((S - lse[..., None]) * LOG2_E)                      # K:367
            dV_acc = dV_acc + _dot32(P.transpose(-2, -1).to(dt), dOb)          # K:370
            dP = _dot32(dOb, Vb.transpose(-2, -1))                             # K:373
            dS = P * (dP - dlt[..., None])                                     # K:379
            dK_acc = dK_acc + _dot32(dS.transpose(-2, -1).to(dt), Qb) * scale  # K:382
        dK[:, :, k0:k0 + cols] = dK_acc[:, :, :cols].to(dt)                    # K:385
        dV[:, :, k0:k0 + cols] = dV_acc[:, :, :cols].to(dt)                    # K:386
    return dK, dV


def fwd_bwd_tiled(Q, K, V, dO, is_causal=False, BLOCK_M=64, BLOCK_N=64) -> Dict[str, torch.Tensor]:
    """Launch order of M:14-128: fwd, then dQ (+delta), then dKV."""
    O, LSE = fwd_tiled(Q, K, V, is_causal, BLOCK_M, BLOCK_N)
    dQ, delta = bwd_dq_tiled(Q, K, V, O, dO, LSE, is_causal, BLOCK_M, BLOCK_N)
    dK, dV = bwd_dkv_tiled(Q, K, V, dO, LSE, delta, is_causal, BLOCK_M, BLOCK_N)
    return {"O": O, "LSE": LSE, "delta": delta, "dQ": dQ, "dK": dK, "dV": dV}


# --------------------------------------------------------------------------
# CPU baseline (bench.py cpu_baseline leg): PyTorch CPU SDPA, BASELINE.md section 3
# --------------------------------------------------------------------------
def cpu_sdpa(Q, K, V, is_causal, dO=None):
    """PyTorch CPU scaled_dot_product_attention, fwd or fwd+bwd (same-precision peer, O3 in SURVEY 8c)."""
    import torch.nn.functional as F

    if dO is None:
        with torch.no_grad():
            return F.scaled_dot_product_attention(Q, K, V, is_causal=is_causal)
    q = Q.detach().requires_grad_(True)
    k = K.detach().requires_grad_(True)
    v = V.detach().requires_grad_(True)
    o = F.scaled_dot_product_attention(q, k, v, is_causal=is_causal)
    o.backward(dO)
    return o.detach(), q.grad, k.grad, v.grad


def attention_varlen_fp64(Q, K, V, dO, cu_q, cu_k, causal):
    """Ground truth for the variable-length extension (reference text Phase_6.md:119-178; not implemented in the
    reference): packed Q [total_q, H, D], K / V [total_k, H, D], dO like Q; cu_q / cu_k = prefix sums (lists or int
    tensors).  Every sequence is an independent attention_fp64 problem.  Returns packed O, dQ, dK, dV (fp64), LSE and
    delta as [H, total_q]."""
    cu_q = [int(x) for x in cu_q]
    cu_k = [int(x) for x in cu_k]
    Tq, H, D = Q.shape
    out = {"O": torch.zeros(Q.shape, dtype=torch.float64), "dQ": torch.zeros(Q.shape, dtype=torch.float64),
           "dK": torch.zeros(K.shape, dtype=torch.float64), "dV": torch.zeros(V.shape, dtype=torch.float64),
           "LSE": torch.zeros(H, Tq, dtype=torch.float64), "delta": torch.zeros(H, Tq, dtype=torch.float64)}
    for b in range(len(cu_q) - 1):
        q0, q1, k0, k1 = cu_q[b], cu_q[b + 1], cu_k[b], cu_k[b + 1]
        if q1 == q0 or k1 == k0:
            continue
        sl = lambda t, a, e: t[a:e].transpose(0, 1).unsqueeze(0)      # [1, H, S, D]
        g = attention_fp64(sl(Q, q0, q1), sl(K, k0, k1), sl(V, k0, k1), sl(dO, q0, q1), causal)
        out["O"][q0:q1] = g["O"][0].transpose(0, 1)
        out["dQ"][q0:q1] = g["dQ"][0].transpose(0, 1)
        out["dK"][k0:k1] = g["dK"][0].transpose(0, 1)
        out["dV"][k0:k1] = g["dV"][0].transpose(0, 1)
        out["LSE"][:, q0:q1] = g["LSE"][0]
        out["delta"][:, q0:q1] = g["delta"][0]
    return out


# --------------------------------------------------------------------------
# attention dropout (reference text Phase_6.md:54-113; not implemented in the reference)
# --------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) on numpy uint32 arrays (broadcast): returns the 4 output words.
    Pinned by the Random123 known-answer vectors in tests/test_oracle.py."""
    import numpy as np
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint32(k0), lo1, hi0 ^ c3 ^ np.uint32(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def dropout_keep_mask(B, H, Sq, Sk, p_drop, seed, offset=0):
    """The keep mask the kernels regenerate (include/mi355fa.h, fa_*_dropout): bool [B, H, Sq, Sk] and the scale
    1 / (1 - p) of the quantised p.  Weight (b, h, q, k): byte (k & 3) of word (q & 3) of
    Philox(counter = {q >> 2, k >> 2, b*H + h, offset}, key = {seed[31:0], seed[63:32]}) >= round(256 p); offset < 2^32."""
    import numpy as np
    thresh = min(255, int(p_drop * 256.0 + 0.5))
    k0 = seed & 0xFFFFFFFF
    assert 0 <= offset < (1 << 32), "the Philox counter has one 32-bit word for the offset (include/mi355fa.h)"
    k1 = (seed >> 32) & 0xFFFFFFFF
    qg = np.arange((Sq + 3) // 4, dtype=np.uint32)[None, :, None]
    kg = np.arange((Sk + 3) // 4, dtype=np.uint32)[None, None, :]
    bh = np.arange(B * H, dtype=np.uint32)[:, None, None]
    words = philox4x32_10(qg, kg, bh, np.uint32(offset & 0xFFFFFFFF), k0, k1)          # 4 x [BH, Sq/4, Sk/4]
    w = np.stack(words, axis=2)                                                       # [BH, Sq/4, 4 (q&3), Sk/4]
    byts = np.stack([(w >> np.uint32(8 * j)) & np.uint32(255) for j in range(4)], axis=-1)   # [..., Sk/4, 4 (k&3)]
    keep = (byts >= thresh).reshape(B * H, 4 * qg.shape[1], 4 * kg.shape[2])[:, :Sq, :Sk]
    return torch.from_numpy(keep.reshape(B, H, Sq, Sk).copy()), 256.0 / (256.0 - thresh)


def attention_dropout_fp64(Q, K, V, dO, is_causal, keep, rp):
    """fp64 attention with the given keep mask on the softmax weights: O = (keep * rp * softmax(S)) V; autograd grads.
    LSE is that of the undropped softmax; delta = rowsum(dO * O)."""
    q = Q.detach().to(torch.float64).requires_grad_(True)
    k = K.detach().to(torch.float64).requires_grad_(True)
    v = V.detach().to(torch.float64).requires_grad_(True)
    D = q.shape[-1]
    S = q @ k.transpose(-2, -1) * (1.0 / math.sqrt(D))
    if is_causal:
        Sq, Sk = S.shape[-2:]
        S = S.masked_fill(~(torch.arange(Sq)[:, None] >= torch.arange(Sk)[None, :]), float("-inf"))
    lse = torch.logsumexp(S, dim=-1)
    P = torch.exp(S - lse[..., None]) * keep.to(torch.float64) * rp
    O = P @ v
    do = dO.detach().to(torch.float64)
    O.backward(do)
    return {"O": O.detach(), "LSE": lse.detach(), "dQ": q.grad, "dK": k.grad, "dV": v.grad,
            "delta": (do * O.detach()).sum(-1)}


def attention_varlen_dropout_fp64(Q, K, V, dO, cu_q, cu_k, causal, p_drop, seed, offset=0):
    """Variable-length attention WITH dropout (both extensions of Phase_6.md composed; include/mi355fa.h, mi355fa_opts):
    sequence b of the packed batch uses the keep mask of batch index b of a padded launch -- Philox counter
    {q >> 2, k >> 2, b*H + h, offset} with q, k counted inside the sequence.  Returns packed O, dQ, dK, dV (fp64)."""
    cu_q = [int(x) for x in cu_q]
    cu_k = [int(x) for x in cu_k]
    nb = len(cu_q) - 1
    Tq, H, D = Q.shape
    mq = max(cu_q[b + 1] - cu_q[b] for b in range(nb))
    mk = max(cu_k[b + 1] - cu_k[b] for b in range(nb))
    keep, rp = dropout_keep_mask(nb, H, mq, mk, p_drop, seed, offset)
    out = {"O": torch.zeros(Q.shape, dtype=torch.float64), "dQ": torch.zeros(Q.shape, dtype=torch.float64),
           "dK": torch.zeros(K.shape, dtype=torch.float64), "dV": torch.zeros(V.shape, dtype=torch.float64)}
    for b in range(nb):
        q0, q1, k0, k1 = cu_q[b], cu_q[b + 1], cu_k[b], cu_k[b + 1]
        if q1 == q0 or k1 == k0:
            continue
        sl = lambda t, a, e: t[a:e].transpose(0, 1).unsqueeze(0)      # [1, H, S, D]
        g = attention_dropout_fp64(sl(Q, q0, q1), sl(K, k0, k1), sl(V, k0, k1), sl(dO, q0, q1), causal,
                                   keep[b:b + 1, :, :q1 - q0, :k1 - k0], rp)
        out["O"][q0:q1] = g["O"][0].transpose(0, 1)
        out["dQ"][q0:q1] = g["dQ"][0].transpose(0, 1)
        out["dK"][k0:k1] = g["dK"][0].transpose(0, 1)
        out["dV"][k0:k1] = g["dV"][0].transpose(0, 1)
    return out
