#!/usr/bin/env python3
"""cProfile of the Python binding's host path (fwd+bwd, tiny shape): where the microseconds per call go."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M  # noqa: E402

Q, K, V = (torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.float16) for _ in range(3))
dO = torch.randn_like(Q)
q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))


def fb():
    o = M.flash_attention(q, k, v, True)
    o.backward(dO)
    q.grad = k.grad = v.grad = None


def t(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("fwd+bwd wall per call: %.1f us" % t(fb))
o = M.flash_attention(q, k, v, True)
print("forward only (grad mode): %.1f us" % t(lambda: M.flash_attention(q, k, v, True)))
print("backward launcher only  : %.1f us" % t(lambda: M.flash_attention_backward(Q, K, V, o.detach(), dO, torch.empty(1, 2, 128, device='cuda'), True)))
for _ in range(100):
    fb()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    fb()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
