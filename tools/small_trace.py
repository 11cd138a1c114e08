#!/usr/bin/env python3
"""A short loop of fwd+bwd steps at one small shape, to be run under `rocprofv3 --kernel-trace`: the start / end stamps of
consecutive kernels show whether the GPU waits for the host (diagnostic).  usage: small_trace.py D S dtype"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
D, S = int(sys.argv[1]), int(sys.argv[2])
dt = torch.bfloat16 if sys.argv[3] == "bf16" else torch.float16
q, k, v = (torch.randn(4, 8, S, D, device="cuda", dtype=dt, requires_grad=True) for _ in range(3))
dO = torch.randn(4, 8, S, D, device="cuda", dtype=dt)
for _ in range(200):
    o = M.flash_attention(q, k, v, True); o.backward(dO); q.grad = k.grad = v.grad = None
torch.cuda.synchronize()
