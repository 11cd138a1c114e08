#!/usr/bin/env python3
"""Where a dQ schedule family differs from another: per 32-row block of every (batch, head), the number of differing / NaN
elements.  usage: diag_dq4.py A B [S] [dtype] (families; default 3 4 256 fp16), causal."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
fa_, fb_ = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float16
code = 1 if dt == torch.bfloat16 else 0
lib = host.lib
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
B, H, D = 1, 1, 64
torch.manual_seed(1)
Q, dO = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(2))
K, V = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(2))
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
sc = D ** -0.5
o, lse = torch.empty_like(Q), torch.empty(B, H, S, device="cuda", dtype=torch.float32)
assert lib.fa_fwd(P(Q), P(K), P(V), P(o), P(lse), B, H, S, S, D, code, 1, sc, st) == 0
outs = []
for fam in (fa_, fb_):
    lib.fa_debug_force_impl(0, fam, 0)
    dq, delta = torch.full_like(Q, float("nan")), torch.full_like(lse, float("nan"))
    assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(o), P(dO), P(lse), P(dq), P(delta), B, H, S, S, D, code, 1, sc, st) == 0
    torch.cuda.synchronize()
    outs.append(dq.float().cpu())
a, b = outs
for blk in range(S // 32):
    x, y = a[0, 0, 32 * blk:32 * blk + 32], b[0, 0, 32 * blk:32 * blk + 32]
    neq = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
    rows = sorted(set(neq.nonzero()[:, 0].tolist()))
    print("row block %2d: differing %4d  nan %4d  max|d| %.4g rows %s" % (blk, int(neq.sum()), int(torch.isnan(y).sum()),
          float((x - y).abs().nan_to_num().max()), rows[:40]))
