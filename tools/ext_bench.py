#!/usr/bin/env python3
"""Throughput of the two extensions beside the plain path (round 2): variable-length batches and attention dropout."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M  # noqa: E402


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def fb(f, *tensors):
    def run():
        o = f()
        o.backward(run.dO)
        for t in tensors:
            t.grad = None
    return run


H, D, dt = 32, 64, torch.bfloat16
B, S = 4, 4096
torch.manual_seed(0)
Q, K, V = (torch.randn(B, H, S, D, device="cuda", dtype=dt, requires_grad=True) for _ in range(3))
dO = torch.randn(B, H, S, D, device="cuda", dtype=dt)
F = 3.5 * 4 * B * H * S * S * D / 2
r = fb(lambda: M.flash_attention(Q, K, V, True), Q, K, V); r.dO = dO
ms = timeit(r)
print("fixed-length  B4 H32 N4096 D64 causal bf16 fwd+bwd: %.3f ms  %.0f TFLOPS" % (ms, F / ms / 1e9))
# the same data as a packed varlen batch of four equal sequences
Qp, Kp, Vp = (x.detach().transpose(1, 2).reshape(B * S, H, D).contiguous().requires_grad_(True) for x in (Q, K, V))
dOp = dO.transpose(1, 2).reshape(B * S, H, D).contiguous()
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
r = fb(lambda: M.flash_attention_varlen(Qp, Kp, Vp, cu, cu, S, S, True), Qp, Kp, Vp); r.dO = dOp
ms = timeit(r)
print("varlen, 4 x 4096 packed                          : %.3f ms  %.0f TFLOPS" % (ms, F / ms / 1e9))
# a ragged batch with the same number of tokens
lens = [8192, 4096, 2048, 1024, 512, 256, 128, 128]
assert sum(lens) == B * S
cu2 = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
F2 = 3.5 * sum(4 * H * n * n * D / 2 for n in lens)
r = fb(lambda: M.flash_attention_varlen(Qp, Kp, Vp, cu2, cu2, max(lens), max(lens), True), Qp, Kp, Vp); r.dO = dOp
ms = timeit(r)
print("varlen, ragged %s: %.3f ms  %.0f TFLOPS (counted on the real lengths)" % (lens, ms, F2 / ms / 1e9))
# padding the same batch to its longest sequence instead
Bp = len(lens)
Qz, Kz, Vz = (torch.randn(Bp, H, max(lens), D, device="cuda", dtype=dt, requires_grad=True) for _ in range(3))
dOz = torch.randn(Bp, H, max(lens), D, device="cuda", dtype=dt)
r = fb(lambda: M.flash_attention(Qz, Kz, Vz, True), Qz, Kz, Vz); r.dO = dOz
ms = timeit(r, 5)
print("the same batch padded to 8 x 8192                : %.3f ms  (%.0f useful TFLOPS)" % (ms, F2 / ms / 1e9))
del Qz, Kz, Vz, dOz
# dropout
r = fb(lambda: M.flash_attention_dropout(Q, K, V, True, 0.1, seed=1, offset=0), Q, K, V); r.dO = dO
ms = timeit(r, 5)
print("dropout p=0.1, B4 H32 N4096 D64 causal fwd+bwd   : %.3f ms  %.0f TFLOPS" % (ms, F / ms / 1e9))
