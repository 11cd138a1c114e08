#!/usr/bin/env python3
"""fwd+bwd step time at the small end of the reference's grid (B4 H8, S 512 / 1024 / 2048), per dtype and forced family
triple, and the backward kernels alone (event-timed through the C ABI launcher used by the binding): separates host
overhead from kernel choice (diagnostic)."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
import _mi355fa as host
lib = host.lib
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3

def step_us(q, k, v, dO, causal, n=300):
    def fb():
        o = M.flash_attention(q, k, v, causal); o.backward(dO); q.grad = k.grad = v.grad = None
    for _ in range(30): fb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fb()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for D in (64, 128):
  for S in (512, 1024, 2048):
    for causal in (True, False):
        line = "D%d S%d %s:" % (D, S, "causal" if causal else "full  ")
        for dt in (torch.bfloat16, torch.float16):
            q, k, v = (torch.randn(4, 8, S, D, device="cuda", dtype=dt, requires_grad=True) for _ in range(3))
            dO = torch.randn(4, 8, S, D, device="cuda", dtype=dt)
            for fam in ((0, 0, 0), (1, 1, 1), (1, 1, 2)):
                lib.fa_debug_force_impl(*fam)
                line += "  %s%s %.1f" % ("bf" if dt == torch.bfloat16 else "fp", "".join(map(str, fam)), step_us(q, k, v, dO, causal))
            lib.fa_debug_force_impl(0, 0, 0)
        print(line, flush=True)
