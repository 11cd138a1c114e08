#!/usr/bin/env python3
"""Per-phase cycle account of a -DFA_STAMPS build of the family-4 forward (diagnostic only; a stamp costs an s_memtime + an
lgkmcnt(0) drain, so read shares, not absolutes).   usage: stamps_fwd4.py [--non-causal] [--dim 128] [lib.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
libp = [a for a in sys.argv[1:] if a.endswith(".so")]
lib = ctypes.CDLL(os.path.join(ROOT, libp[0] if libp else "ab/stamps.so"))
for name, (res, args) in host.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
causal = "--non-causal" not in sys.argv
D = int(sys.argv[sys.argv.index("--dim") + 1]) if "--dim" in sys.argv else 64
B, H, S = 4, 32, 4096
torch.manual_seed(0)
Q, K, V = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
c, sc = int(causal), D ** -0.5
lib.fa_debug_force_impl(4, 0, 0)
nqt = S // 256
nwg = (nqt // 2 if causal else nqt) * B * H
if causal and "--not-persistent" not in sys.argv:   # causal launches are persistent: one workgroup per CU (round 4)
    nwg = min(nwg, torch.cuda.get_device_properties(0).multi_processor_count // 8 * 8)
dbg = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(12):
    lib.fa_debug_set_buffer(dbg.data_ptr())
    if i == 11: e0.record()
    assert lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, 1, c, sc, st) == 0
    if i == 11: e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu().view(nwg, 4, 16).double()
names = ["pass prologue: Q fragments, ring primed, tile 0 landed", "scout block + row constants", "pipeline fill",
         "plain tiles", "masked tiles (the 256 keys level with the query tile)", "drain",
         "end-of-pass check (row sums, flag, barrier)", "epilogue (+ priming the next pass's ring)",
         "loop bookkeeping: next item decoded, descriptors", "an unprimed pass's requests (Q rows, first tiles)", "Q fragments", "-"]
tot = d[:, :, :12].sum()
passes, tiles = d[:, :, 12].sum(), d[:, :, 13].sum()
bn = 128 if D == 64 else 64
print("%s D=%d: kernel %.3f ms; passes per wave %.2f, plain tiles (%d keys) per pass %.2f; stamped cycles per pass %.0f; per plain tile %.0f (MFMA pipe: %d)"
      % ("causal" if causal else "non-causal", D, ms, d[:, :, 12].mean(), bn, tiles / passes, tot / passes, d[:, :, 3].sum() / max(tiles, 1),
         2 * (bn // 32) * (2 * D // 16) * 32))
for i, n in enumerate(names):
    print("  %-58s %5.1f%%  %8.0f cycles per pass" % (n, 100 * d[:, :, i].sum() / tot, d[:, :, i].sum() / passes))
print("whole-wave: s_memtime %.0f cycles, s_memrealtime %.0f (100 MHz) => shader clock %.3f GHz; stamped share of wave lifetime %.3f"
      % (d[:, :, 14].mean(), d[:, :, 15].mean(), 0.1 * d[:, :, 14].sum() / d[:, :, 15].sum(), tot / d[:, :, 14].sum()))
