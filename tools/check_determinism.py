#!/usr/bin/env python3
"""Bit-for-bit repeatability of the three kernels at the headline shapes (no atomics anywhere: any difference is a
race -- this is how the raw-barrier race of DESIGN.md section 3 was found).
usage: check_determinism.py [iterations] [fwd,dq,dkv schedule families to force, e.g. 3,3,0]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
import _scaling as sc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:
    import _mi355fa as _fa
    _fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
    _fa.lib.fa_debug_force_impl(*[int(x) for x in sys.argv[2].split(",")])
    print("forced schedule families (fwd, dq, dkv):", sys.argv[2])
bad = 0
for D, H, dtype, causal in ((64, 32, torch.bfloat16, True), (64, 32, torch.float16, True), (64, 32, torch.bfloat16, False),
                            (128, 16, torch.bfloat16, True), (64, 32, torch.float16, False)):
    Q, K, V, dO = sc.make_shard(0, 4, H, 4096, 4096, D, dtype, torch.device("cuda"))
    O, L = M.flash_attention_forward(Q, K, V, causal)
    ref = (O, L) + tuple(M.flash_attention_backward(Q, K, V, O, dO, L, causal))
    diff = [0] * 5
    for it in range(n):
        O2, L2 = M.flash_attention_forward(Q, K, V, causal)
        got = (O2, L2) + tuple(M.flash_attention_backward(Q, K, V, O2, dO, L2, causal))
        for i, (a, b) in enumerate(zip(ref, got)):
            diff[i] += int((a != b).sum())
    bad += sum(diff)
    print("D=%d %s %s: differing elements over %d repeats  O %d  LSE %d  dQ %d  dK %d  dV %d"
          % (D, str(dtype).split(".")[1], "causal" if causal else "full", n, *diff), flush=True)
print("DETERMINISTIC" if bad == 0 else "NON-DETERMINISTIC: %d elements" % bad)
sys.exit(1 if bad else 0)
