"""TFLOPS benchmark: counterpart of the reference's code/Performance_Comparison.py.

Same entry points and conventions -- benchmark_attention(provider, mode, B, H, S_q, S_k, D,
is_causal, device, warmup=10, repeat=30) -> (avg_time_ms, tflops) (P:9-109), timing (P:111-128),
naive_attention (P:130-144); providers 'naive' / 'triton' / 'pytorch' keep their names ('triton'
= this repository's HIP kernels behind flash_attention, so the reference's driver loop runs
unchanged); modes fwd / fwd_bwd / bwd := fwd_bwd - fwd (P:92-93); counted FLOPs P:101-107.
Added: a `dtype` argument (the reference is fp16-only; the MI355X headline is bf16).
The 'pytorch' provider pins the FLASH backend under 16-bit autocast exactly as P:53-57 does; where this PyTorch-ROCm
build refuses that backend for the shape / dtype, it falls back to the default backend selection and says so:
`last_sdpa_backend()` returns "flash" or "default (<why flash was refused>)" for the most recent 'pytorch' run.
"""
from typing import Literal, Tuple

import torch
import torch.nn.functional as F
from torch.amp import autocast
from torch.nn.attention import SDPBackend, sdpa_kernel

_sdpa_backend = {"last": None}


def last_sdpa_backend():
    """Which SDPA backend the most recent provider='pytorch' benchmark ran: "flash" (pinned, as in the reference) or
    "default (...)" when the pin was refused by this build."""
    return _sdpa_backend["last"]


def _sdpa_flash(Q, K, V, is_causal, dtype):
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        with autocast(device_type=Q.device.type, dtype=dtype):
            return F.scaled_dot_product_attention(Q, K, V, is_causal=is_causal)


def _pick_sdpa(Q, K, V, is_causal, dtype):
    """P:53-57 pins SDPBackend.FLASH_ATTENTION + autocast; try that once (forward AND backward), otherwise default."""
    try:
        O = _sdpa_flash(Q, K, V, is_causal, dtype)
        if O.requires_grad:
            O.sum().backward()
            Q.grad = K.grad = V.grad = None
        _sdpa_backend["last"] = "flash"
        return lambda: _sdpa_flash(Q, K, V, is_causal, dtype)
    except RuntimeError as e:
        _sdpa_backend["last"] = "default (%s)" % str(e).strip().split("\n")[0][:120]
        return lambda: F.scaled_dot_product_attention(Q, K, V, is_causal=is_causal)


def benchmark_attention(
    provider: Literal['naive', 'triton', 'pytorch'],
    mode: Literal['fwd', 'bwd', 'fwd_bwd'],
    B: int,
    H: int,
    S_q: int,
    S_k: int,
    D: int,
    is_causal: bool,
    device: torch.device,
    warmup: int = 10,
    repeat: int = 30,
    dtype: torch.dtype = torch.float16,
) -> Tuple[float, float]:
    Q = torch.randn(B, H, S_q, D, device=device, dtype=dtype, requires_grad=True)
    K = torch.randn(B, H, S_k, D, device=device, dtype=dtype, requires_grad=True)
    V = torch.randn(B, H, S_k, D, device=device, dtype=dtype, requires_grad=True)
    dO = torch.randn(B, H, S_q, D, device=device, dtype=dtype)

    if provider == "naive":
        def fn():
            return naive_attention(Q, K, V, is_causal)
    elif provider == "triton":
        from My_FlashAttention_optimized import flash_attention

        def fn():
            return flash_attention(Q, K, V, is_causal)
    else:  # pytorch
        fn = _pick_sdpa(Q, K, V, is_causal, dtype)

    Q.grad = None
    K.grad = None
    V.grad = None

    def run_fwd():
        return fn()

    def run_all():
        O = fn()
        O.backward(dO)
        Q.grad = None
        K.grad = None
        V.grad = None
        return O

    if mode == 'fwd':
        avg_time_ms = timing(run_fwd, warmup, repeat)
    elif mode == 'fwd_bwd':
        avg_time_ms = timing(run_all, warmup, repeat)
    else:  # bwd is derived, P:92-93
        avg_time_ms = timing(run_all, warmup, repeat) - timing(run_fwd, warmup, repeat)

    flops = 4 * B * H * S_q * S_k * D // (2 if is_causal else 1)
    mult = {'fwd': 1.0, 'bwd': 2.5, 'fwd_bwd': 3.5}[mode]
    tflops = mult * flops / (avg_time_ms * 1e-3) / 1e12
    return avg_time_ms, tflops


def timing(run_fn, warmup, repeat):
    for _ in range(warmup):
        run_fn()
    starter = torch.cuda.Event(enable_timing=True)
    ender = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    starter.record()
    for _ in range(repeat):
        run_fn()
    ender.record()
    torch.cuda.synchronize()
    return starter.elapsed_time(ender) / repeat


def naive_attention(Q, K, V, is_causal):
    scale = 1 / (Q.shape[-1] ** 0.5)
    S = Q @ K.transpose(-2, -1) * scale
    if is_causal:
        seq_len = S.shape[-1]
        mask = torch.triu(torch.ones(seq_len, seq_len, device=S.device), diagonal=1).bool()
        S = S.masked_fill(mask, float('-inf'))
    P = torch.softmax(S, dim=-1)
    return P @ V


if __name__ == '__main__':
    import sys
    DEVICE = torch.device(torch.cuda.current_device())
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float16
    for is_causal in (True, False):
        for mode in ('fwd', 'fwd_bwd'):
            for provider in ['pytorch', 'triton']:
                result = []
                for S in [512, 1024, 2048, 4096, 8192, 16384]:
                    _, tflops = benchmark_attention(provider=provider, mode=mode, B=4, H=8, S_q=S, S_k=S, D=D,
                                                    is_causal=is_causal, device=DEVICE, dtype=dt)
                    result.append(round(tflops, 1))
                print("D=%d %s %s %-8s %s" % (D, "causal" if is_causal else "full", mode, provider, result), flush=True)
    print("# torch SDPA backend of the 'pytorch' provider: %s" % last_sdpa_backend(), flush=True)
