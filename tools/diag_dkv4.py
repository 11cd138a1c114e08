#!/usr/bin/env python3
"""Where a dK/dV schedule family differs from another: per 32-key block, the number of differing elements of dK and dV.
usage: diag_dkv4.py A B [S] [dtype] [lib.so] (families; default 3 4 256 bf16), causal."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
fa_, fb_ = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = torch.float16 if (len(sys.argv) > 4 and sys.argv[4] == "fp16") else torch.bfloat16
code = 1 if dt == torch.bfloat16 else 0
libp = [a for a in sys.argv[1:] if a.endswith(".so")]
lib = host.lib
if libp:
    lib = ctypes.CDLL(os.path.join(ROOT, libp[0]))
    for name, (res, args) in host.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
B, H, D = 1, 1, 64
torch.manual_seed(1)
Q, dO = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(2))
K, V = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(2))
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
sc = D ** -0.5
o, lse = torch.empty_like(Q), torch.empty(B, H, S, device="cuda", dtype=torch.float32)
dq, delta = torch.empty_like(Q), torch.empty_like(lse)
lib.fa_debug_force_impl(1, 1, 0)
assert lib.fa_fwd(P(Q), P(K), P(V), P(o), P(lse), B, H, S, S, D, code, 1, sc, st) == 0
assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(o), P(dO), P(lse), P(dq), P(delta), B, H, S, S, D, code, 1, sc, st) == 0
outs = []
for fam in (fa_, fb_):
    lib.fa_debug_force_impl(0, 0, fam)
    dk, dv = torch.full_like(K, float("nan")), torch.full_like(V, float("nan"))
    assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(lse), P(delta), P(dk), P(dv), B, H, S, S, D, code, 1, sc, st) == 0
    torch.cuda.synchronize()
    outs.append((dk.float().cpu(), dv.float().cpu()))
for blk in range(S // 32):
    line = "key block %2d (wave %d, group %d):" % (blk, min(blk % 8, 7 - blk % 8), int(blk % 8 >= 4))
    for name, a, b in (("dK", outs[0][0], outs[1][0]), ("dV", outs[0][1], outs[1][1])):
        x, y = a[0, 0, 32 * blk:32 * blk + 32], b[0, 0, 32 * blk:32 * blk + 32]
        neq = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
        line += "  %s differing %4d nan %3d max|d| %.3g" % (name, int(neq.sum()), int(torch.isnan(y).sum()), float((x - y).abs().nan_to_num().max()))
    print(line)
