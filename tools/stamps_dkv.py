#!/usr/bin/env python3
"""Per-segment cycle shares of a -DFA_STAMPS build of the dK/dV kernel, schedule family 2 (diagnostic only).
usage: stamps_dkv.py [--non-causal] [--dim 128]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
lib = ctypes.CDLL(os.path.join(ROOT, "ab/stamps.so"))
for name, (res, args) in host.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_set_buffer.argtypes = [ctypes.c_void_p]
causal = "--non-causal" not in sys.argv
B, H, S, D = 4, 32, 4096, (int(sys.argv[sys.argv.index("--dim") + 1]) if "--dim" in sys.argv else 64)
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
lib.fa_debug_force_impl(0, 0, 2)
torch.manual_seed(0)
Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
dQ, dK, dV, delta = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(LSE)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
c, sc = int(causal), D ** -0.5
lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, 1, c, sc, st)
lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, 1, c, sc, st)
nblk = (S // 128) * B * H
dbg = torch.zeros(nblk * 4 * 12, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(12):
    lib.fa_debug_set_buffer(dbg.data_ptr())
    if i == 11: e0.record()
    assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(delta), P(dK), P(dV), B, H, S, S, D, 1, c, sc, st) == 0
    if i == 11: e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu().view(nblk, 4, 12).double()
names = ["DMA issue (per tile)", "slots 0-3 (S)", "slots 4-7 (dP)", "slots 8-11 (dV)", "barrier (per tile)", "slots 12-15 (dK)", "vmcnt(0) wait (per tile)", "row constants -> LDS (per tile)"]
blocks = d[:, :, 8].sum()
tot = d[:, :, :8].sum()
print("query blocks per wave (mean): %.1f; total stamped cycles per block: %.0f" % (d[:, :, 8].mean(), tot / blocks))
for i, n in enumerate(names):
    print("  %-28s %5.1f%%  %7.0f cycles per 32x32 block" % (n, 100 * d[:, :, i].sum() / tot, d[:, :, i].sum() / blocks))
print("per-wave share of the barrier segment:", [round(float(d[:, w, 4].sum() / d[:, :, 4].sum()), 3) for w in range(4)])
print("per-wave share of the vmcnt segment:  ", [round(float(d[:, w, 6].sum() / max(1.0, d[:, :, 6].sum())), 3) for w in range(4)])
wg_per_cu = 1 if int(os.environ.get("FA_LDS_PAD", "0")) > 20000 else 2
rounds = nblk / (256 * wg_per_cu)
per_wave = float(d[:, :, :8].sum(dim=2).mean())
print("kernel %.3f ms; stamped cycles per wave %.0f x %.1f sequential workgroups per CU slot => shader clock >= %.2f GHz"
      % (ms, per_wave, rounds, per_wave * rounds / (ms * 1e6)))
print("whole-wave: s_memtime %.0f, s_memrealtime %.0f (100 MHz) => shader clock %.3f GHz; stamped share of wave lifetime %.3f"
      % (d[:, :, 9].mean(), d[:, :, 10].mean(), 0.1 * d[:, :, 9].sum() / d[:, :, 10].sum(), d[:, :, :8].sum() / d[:, :, 9].sum()))
