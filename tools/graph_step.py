#!/usr/bin/env python3
"""Eager vs hipGraph replay of one fwd+bwd step through the autograd binding (diagnostic).
usage: graph_step.py [B H S D] [fp16]   (default 4 32 4096 64 bf16; the small end of the reference's grid: 4 8 512 64)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
import _scaling as sc
dev = torch.device("cuda")
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B, H, S, D = nums if len(nums) == 4 else (4, 32, 4096, 64)
dt = torch.float16 if "fp16" in sys.argv else torch.bfloat16
print("B%d H%d S%d D%d %s causal" % (B, H, S, D, dt))
Q, K, V, dO = sc.make_shard(0, B, H, S, S, D, dt, dev)
for t in (Q, K, V): t.requires_grad_(True)
def step():
    O = M.flash_attention(Q, K, V, True)
    O.backward(dO)
    Q.grad = None; K.grad = None; V.grad = None
def timeit(fn, n=300, w=30):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager   : %.4f ms per step" % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        step()
    print("graphed : %.4f ms per step" % timeit(g.replay))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
print("eager   : %.4f ms per step" % timeit(step))
