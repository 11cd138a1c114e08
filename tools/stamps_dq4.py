#!/usr/bin/env python3
"""Per-phase cycle account of a -DFA_STAMPS [-DFA_STAMPS_ITER] build of the family-4 dQ kernel (diagnostic only; a stamp
costs an s_memtime + lgkmcnt(0) drain, so read shares, not absolutes).   usage: stamps_dq4.py [--non-causal] lib.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
libp = [a for a in sys.argv[1:] if a.endswith(".so")]
lib = ctypes.CDLL(os.path.join(ROOT, libp[0] if libp else "ab/stamps_dq4.so"))
for name, (res, args) in host.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
causal = "--non-causal" not in sys.argv
DKV = "--dkv" in sys.argv   # the family-4 dK/dV kernel (same record layout, five phases)
B, H, S, D = 4, 32, 4096, 64
torch.manual_seed(0)
Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
dQ, delta = torch.empty_like(Q), torch.empty_like(LSE)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
c, sc = int(causal), D ** -0.5
lib.fa_debug_force_impl(1, 1 if DKV else 4, 4)
lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, 1, c, sc, st)
dK, dV = torch.empty_like(K), torch.empty_like(V)
if DKV:
    lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, 1, c, sc, st)
nqt = S // 256
nwg = (nqt // 2 if causal else nqt) * B * H
dbg = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(12):
    lib.fa_debug_set_buffer(dbg.data_ptr())
    if i == 11: e0.record()
    if DKV:
        assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(delta), P(dK), P(dV), B, H, S, S, D, 1, c, sc, st) == 0
    else:
        assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, 1, c, sc, st) == 0
    if i == 11: e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu().view(nwg, 4, 32).double()
life = d[:, :, 18].sum()
passes = d[:, :, 20].sum()
tiles = d[:, :, 17].sum()
print("%s: kernel %.3f ms; waves %d, passes per wave %.2f, unmasked tiles per wave %.1f" % (
    "causal" if causal else "non-causal", ms, nwg * 4, passes / (nwg * 4), tiles / (nwg * 4)))
names = ["pipeline fill", "unmasked tiles", "diagonal phase (9 block visits per wave)",
         "drain", "epilogue (dQ staged and stored, scaled-Q workspace)"]
for i, n in enumerate(names):
    print("  %-70s %5.1f%% of wave lifetime  %8.0f cycles per pass" % (n, 100 * d[:, :, i].sum() / life, d[:, :, i].sum() / passes))
for i, n in ((5, "prologue: ring primed (20 LDS-DMA pieces issued)"), (6, "prologue: Q / dO / O fetched, delta, scale, pinned"), (7, "prologue: first barrier (vmcnt(0))")):
    print("  %-70s %5.1f%% of wave lifetime  %8.0f cycles per pass" % (n, 100 * d[:, :, i].sum() / life, d[:, :, i].sum() / passes))
if d[:, :, 9].sum() > 0 and d[:, :, 16].sum() == 0 and not DKV:
    for i, n in ((9, "  of the prologue: loop bookkeeping (item decode, descriptors)"), (10, "  of the prologue: lane addresses, wait for the staged rows"),
                 (11, "  diagonal phase: landing wait, barrier, next-stage descriptors"), (12, "  diagonal phase: 2 w visits below both diagonals"),
                 (13, "  diagonal phase: the two visits of key block w"), (14, "  diagonal phase: 6 - 2 w solo visits of row block 1")):
        print("  %-70s %5.1f%% of wave lifetime  %8.0f cycles per pass" % (n, 100 * d[:, :, i].sum() / life, d[:, :, i].sum() / passes))
print("  unmasked tile: %.0f stamped cycles (96 MFMAs = 3072 matrix cycles)" % (d[:, :, 1].sum() / tiles))
if d[:, :, 16].sum() > 0:   # a -DFA_STAMPS_ITER build (seg[16] = the commit)
    for i in range(8):
        print("    iteration %d (key block %d, row block %d)%s  %6.0f cycles per tile" % (
            i, i >> 1, i & 1, " without its commit" if i == 7 else "", d[:, :, 8 + i].sum() / tiles))
    print("    commit (vmcnt(8), lgkmcnt(0), s_barrier)            %6.0f cycles per tile; per wave %s" % (
        d[:, :, 16].sum() / tiles, [round(float(d[:, w, 16].sum() / max(d[:, w, 17].sum(), 1))) for w in range(4)]))
print("whole wave: s_memtime %.0f cycles, s_memrealtime %.0f (100 MHz) => shader clock %.3f GHz; stamped share of wave lifetime %.3f"
      % (d[:, :, 18].mean(), d[:, :, 19].mean(), 0.1 * d[:, :, 18].sum() / d[:, :, 19].sum(), (d[:, :, :5].sum() + d[:, :, 5:8].sum() + (d[:, :, 9:14].sum() if d[:, :, 16].sum() == 0 else 0)) / life))
