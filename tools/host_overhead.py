#!/usr/bin/env python3
"""Host-side cost of one call through the Python binding (diagnostic): tiny shapes, so the GPU is never the limit."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
import torch.nn.functional as F
Q, K, V = (torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.float16) for _ in range(3))
dO = torch.randn_like(Q)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    print("fwd launcher only   : %.1f us" % t(lambda: M.flash_attention_forward(Q, K, V, True)))
    print("flash_attention fwd : %.1f us" % t(lambda: M.flash_attention(Q, K, V, True)))
    print("torch SDPA fwd      : %.1f us" % t(lambda: F.scaled_dot_product_attention(Q, K, V, is_causal=True)))
q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
def fb():
    o = M.flash_attention(q, k, v, True); o.backward(dO); q.grad = k.grad = v.grad = None
def fb2():
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True); o.backward(dO); q.grad = k.grad = v.grad = None
print("flash_attention fwd+bwd : %.1f us" % t(fb, 1000))
print("torch SDPA fwd+bwd      : %.1f us" % t(fb2, 1000))
