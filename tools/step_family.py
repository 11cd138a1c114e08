#!/usr/bin/env python3
"""Whole fwd+bwd steps (autograd, like bench.py) with the forward pinned to one schedule family after another, interleaved:
the family table is tuned on single kernels, this checks a choice in the mix the three kernels really run in (the chip is
power-managed: a kernel's clock depends on what ran just before it).  usage: step_family.py [fwd families, default 1,2]
[--non-causal] [--dtype fp16]   Not part of the product."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
import _mi355fa as fa
fams = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1][0].isdigit() else "1,2").split(",")]
causal = "--non-causal" not in sys.argv
dt = torch.float16 if "fp16" in sys.argv else torch.bfloat16
fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
torch.manual_seed(0)
Q, K, V, dO = (torch.randn(4, 32, 4096, 64, device="cuda", dtype=dt) for _ in range(4))
for t in (Q, K, V):
    t.requires_grad_(True)
def step():
    M.flash_attention(Q, K, V, causal).backward(dO)
    Q.grad = K.grad = V.grad = None
def fwd():
    with torch.no_grad():
        M.flash_attention(Q, K, V, causal)
def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10): fn()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {f: ([], []) for f in fams}
for rnd in range(6):
    for f in fams:
        fa.lib.fa_debug_force_impl(f, 0, 0)
        a, b = timed(step, 40), timed(fwd, 60)
        if rnd:
            res[f][0].append(a); res[f][1].append(b)
fa.lib.fa_debug_force_impl(0, 0, 0)
import statistics
for f in fams:
    print("forward family %d: fwd+bwd step %.4f ms   forward-only %.4f ms" % (f, statistics.median(res[f][0]), statistics.median(res[f][1])))
