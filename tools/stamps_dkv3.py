#!/usr/bin/env python3
"""Per-phase cycle account of a -DFA_STAMPS build of the family-3 dK/dV kernel (diagnostic only; the stamps cost an
s_memtime + lgkmcnt(0) drain each, so read shares, not absolutes).   usage: stamps_dkv3.py [--non-causal] [lib.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
libp = [a for a in sys.argv[1:] if a.endswith(".so")]
lib = ctypes.CDLL(os.path.join(ROOT, libp[0] if libp else "ab/stamps3.so"))
for name, (res, args) in host.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
causal = "--non-causal" not in sys.argv
B, H, S, D = 4, 32, 4096, 64
torch.manual_seed(0)
Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
dQ, dK, dV, delta = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(LSE)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
c, sc = int(causal), D ** -0.5
lib.fa_debug_force_impl(0, 0, 3)
lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, 1, c, sc, st)
lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, 1, c, sc, st)
nkt = S // 256
nwg = (nkt // 2 if causal else nkt) * B * H
dbg = torch.zeros(nwg * 4 * 64, dtype=torch.int64, device="cuda")   # 16 per wave + 48 per wave of --slots builds
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(12):
    lib.fa_debug_set_buffer(dbg.data_ptr())
    if i == 11: e0.record()
    assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(delta), P(dK), P(dV), B, H, S, S, D, 1, c, sc, st) == 0
    if i == 11: e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu()[:nwg * 64].view(nwg, 4, 16).double()
slots = dbg.cpu()[nwg * 64:].view(nwg, 4, 3, 16).double()
names = ["iteration %d (qb %d, group %d)" % (i, i >> 1, i & 1) for i in range(8)] + \
        ["commit: vmcnt(0) + row constants + barrier (per tile)", "prologue + diagonal tiles (per pass)", "pipeline fill (per pass)",
         "pipeline drain (per pass)", "epilogue (per pass)"]
iters = d[:, :, 13].sum()
tot = d[:, :, :13].sum()
tiles = iters / 8
print("%s: kernel %.3f ms; pipelined block iterations per wave (mean) %.0f; stamped cycles per pipelined tile (all phases): %.0f"
      % ("causal" if causal else "non-causal", ms, d[:, :, 13].mean(), tot / tiles))
for i, n in enumerate(names):
    per = d[:, :, i].sum() / (tiles if i < 9 else d.shape[0] * 4)
    print("  %-56s %5.1f%%  %8.0f cycles per %s" % (n, 100 * d[:, :, i].sum() / tot, per, "tile" if i < 9 else "wave"))
print("per-wave share of the commit segment:", [round(float(d[:, w, 8].sum() / d[:, :, 8].sum()), 3) for w in range(4)])
print("whole-wave: s_memtime %.0f cycles, s_memrealtime %.0f (100 MHz) => shader clock %.3f GHz; stamped share of wave lifetime %.3f"
      % (d[:, :, 14].mean(), d[:, :, 15].mean(), 0.1 * d[:, :, 14].sum() / d[:, :, 15].sum(), d[:, :, :13].sum() / d[:, :, 14].sum()))
if slots.sum() > 0:   # -DFA_STAMPS_SLOTS build: cycles per slot (incl. ~30 for the stamp) in iterations 0, 7 and 3
    per = slots.sum(dim=(0, 1)) / (d[:, :, 13].sum() / 8)
    for k, nm in enumerate(("iteration 0", "iteration 7", "iteration 3")):
        print("  %s per slot:" % nm, " ".join("%4.0f" % x for x in per[k]))
